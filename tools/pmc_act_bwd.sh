#!/bin/bash
# usage (GPU box, repo root): tools/pmc_act_bwd.sh  -> gpurun_out/profiles/r06_act_backward_pmc.txt
# VALU counters of the fused GELU backward (ste_relu_bwd_kernel<..., DACT>), of the gated site backward and of ATen's gelu_backward
# on 25,088 x 3,072 (tools/bench_act_bwd.py): one rocprofv3 --pmc pass per counter group (no trace domain next to the counters
# but --kernel-trace), a --kernel-trace --stats pass for the durations.
root=$(pwd); out=$root/gpurun_out/profiles; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pab_*
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab_t -o t -- python3 $root/tools/bench_act_bwd.py > /tmp/pab_t.out 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pab_a -o a -- python3 $root/tools/bench_act_bwd.py > /tmp/pab_a.out 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pab_b -o b -- python3 $root/tools/bench_act_bwd.py > /tmp/pab_b.out 2>&1
python3 - <<'PY' > $out/r06_act_backward_pmc.txt
import csv, glob, collections
def load(d):
    f = glob.glob(f"/tmp/{d}/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
stats = {r["Name"]: r for r in csv.DictReader(open(glob.glob("/tmp/pab_t/**/*kernel_stats.csv", recursive=True)[0]))}
a, b = load("pab_a"), load("pab_b")
n = 25088 * 3072
print("25,088 x 3,072 elements per launch; counters: mean per launch (rocprofv3 --pmc, one pass per group); us: rocprofv3 --kernel-trace --stats average")
print("SQ_INSTS_VALU counts wave-level instructions: x 64 lanes / elements = VALU instructions per element")
for name in sorted(set(a) | set(b)):
    if not any(k in name for k in ("ste_relu_bwd_kernel", "GeluBackward", "ew_kernel<qs::SteBwdOp")):
        continue
    c = {k: sum(v) / len(v) for k, v in list(a.get(name, {}).items()) + list(b.get(name, {}).items())}
    us = float(stats[name]["AverageNs"]) / 1e3 if name in stats else float("nan")
    valu = c.get("SQ_INSTS_VALU", 0)
    print(f"\n{name[:150]}\n  {us:8.1f} us  launches {len(next(iter(a.get(name, b.get(name)).values())))}  VALU instr/elem {valu * 64 / n:6.1f}  "
          + "  ".join(f"{k}={v:.3g}" for k, v in sorted(c.items())))
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        print(f"  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.3f} (share of wave time issuing VALU work)")
PY
cat $out/r06_act_backward_pmc.txt | cut -c1-400
