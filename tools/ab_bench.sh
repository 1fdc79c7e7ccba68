#!/bin/bash
# A/B on ONE box: the whole bench line (no CPU baseline, no DDP children) of an older tree checked out under ab_old/ (git worktree add
# ab_old <commit>; build it there) against the current tree, interleaved; prints the figures side by side
export QS_BENCH_NO_EXCHANGE_LIVE=1
for i in 1 2; do
  (cd ab_old && python3 bench.py --no-cpu-baseline > ../gpurun_out/abb_old_$i.json 2> ../gpurun_out/abb_old_$i.err)
  python3 bench.py --no-cpu-baseline > gpurun_out/abb_new_$i.json 2> gpurun_out/abb_new_$i.err
done
python3 - <<'PY'
import json
def row(r):
    c, k = r["config"], r["configs"]
    out = {"value": r["value"], "dense": c["dense_gelem_s"], "all": c["elide_all_gelem_s"], "frozen": c["frozen_mask_gelem_s"],
           "bwd_frac": r["roofline"]["frac"], "fwd_ms": r["roofline"]["apply_fwd_ms"], "stats_ms": r["roofline"]["stats_ms"]}
    out["c2"] = (k["config2_quantize8_256x64x56x56"]["eager"]["ms_per_step"], k["config2_quantize8_256x64x56x56"]["graph_replay"]["ms_per_step"])
    out["tok_site"] = k["token_major_site_256x197x3072"]["ms_per_step"]
    w = k["weights_pruned_quantized"]["resnet18_b128"]; out["weights"] = (w["plain_ms"], w["multi_tensor_ms"])
    h = k["host_overhead_per_site"]; out["host"] = {a: h[a] for a in ("plain_relu", "relu_prune_quantize_pair", "relu_quantize", "quantize_alone", "relu_prune_quantize_pair_token_major")}
    for n in ("config3_resnet18_cifar_b128", "config4_resnet50_imagenet_b256"):
        v = k[n]; out[n[:7]] = (v["plain_ms"], v["pq_ms"], v.get("pq_graph_ms"), v["opt_in_extensions"].get("pq_ms"), v["autocast_image_off"].get("pq_ms"))
    return out
for t in ("old_1", "new_1", "old_2", "new_2"):
    try:
        print(t, json.dumps(row(json.loads(open(f"gpurun_out/abb_{t}.json").read().strip().splitlines()[-1]))))
    except Exception as e:
        print(t, "error", repr(e))
PY
