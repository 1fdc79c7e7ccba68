#!/bin/bash
# A/B on ONE box: config 4 (and 3) of the bench, old tree (ab_old, 8ca39a0) vs the current one, interleaved twice
export QS_BENCH_NO_EXCHANGE_LIVE=1
for i in 1 2; do
  (cd ab_old && python3 bench.py --configs-only config4,config3 > ../gpurun_out/ab_old_$i.json 2> ../gpurun_out/ab_old_$i.err)
  python3 bench.py --configs-only config4,config3 > gpurun_out/ab_new_$i.json 2> gpurun_out/ab_new_$i.err
done
python3 - <<'PY'
import json
for t in ("old_1","new_1","old_2","new_2"):
    try:
        r=json.load(open(f"gpurun_out/ab_{t}.json"))["configs"]
        print(t, {k.split("_")[0]:(v.get("plain_ms"), v.get("pq_ms"), v.get("pq_graph_ms"), v.get("pq_over_plain")) for k,v in r.items()})
    except Exception as e:
        print(t, "error", e)
PY
