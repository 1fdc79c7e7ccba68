#!/usr/bin/env python3
"""Regenerates the evidence under profiles/ on an MI355X box (writes to gpurun_out/profiles/, to be copied):

  rNN_bench_default.json                 the JSON record of a plain `python3 bench.py`
  rNN_bench_kernel_stats.csv             rocprofv3 --kernel-trace --stats summary of the same command
  rNN_bench_step_timeline.txt            the last three headline steps of that trace, kernel by kernel
  rNN_headline_kernel_stats.csv          the same summary for `python3 bench.py --no-configs --no-variants --no-cpu-baseline`:
                                         headline kernels only, so the averages are the roofline's per-launch durations
  rNN_configN_kernel_stats.csv           the same summary for `python3 bench.py --configs-only configN` (N = 2, 3, 4)
  rNN_pmc_<COUNTER>_counter_collection.csv   rocprofv3 --pmc <COUNTER> rows of the library's kernels (one pass per counter)
  rNN_pmc_traffic.json                   HBM bytes per launch of the three streaming kernels, corrected as
                                         /opt/skills/guides/MI355X_MICROARCH.md prescribes, next to the algorithmic bytes

usage (from the repository root, on the GPU box):  python3 tools/refresh_profiles.py [round-tag, default r02] [all|gate]
(a second argument "all": only the PMC passes, for the opt-in mode elide_pruned="all" -> rNN_pmc_traffic_all.json;
 "gate": the PMC passes of tools/gate_traffic.py -- a ReLU site's forward / backward with and without the gate bitmap
 -> rNN_pmc_traffic_gate.json)
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
MODE_ALL = len(sys.argv) > 2 and sys.argv[2] == "all"
MODE_GATE = len(sys.argv) > 2 and sys.argv[2] == "gate"
OUT = os.path.join(ROOT, "gpurun_out", "profiles")
BENCH = os.path.join(ROOT, "bench.py")
ENV = dict(os.environ, TMPDIR="/tmp")
NUMEL = 256 * 256 * 56 * 56


def sh(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, cwd="/tmp", env=ENV, text=True, **kw)


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    assert hits, f"nothing matches {pattern}"
    return hits[-1]


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    if MODE_GATE:
        return gate_passes()
    if MODE_ALL:
        r = sh(["python3", BENCH, "--elide", "all", "--no-configs", "--no-cpu-baseline", "--no-variants"], stdout=subprocess.PIPE)
        assert r.returncode == 0
        return pmc_passes(json.loads(r.stdout.strip().splitlines()[-1]), ["--elide", "all"], "_all")

    # 1. the plain command
    r = sh(["python3", BENCH], stdout=subprocess.PIPE)
    assert r.returncode == 0
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    with open(os.path.join(OUT, f"{TAG}_bench_default.json"), "w") as f:
        json.dump(rec, f, indent=1)

    # 2. kernel stats of the same command
    d = os.path.join(OUT, "stats")
    assert sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "bench", "--",
               "python3", BENCH]).returncode == 0
    shutil.copy(find(os.path.join(d, "**", "*kernel_stats.csv")), os.path.join(OUT, f"{TAG}_bench_kernel_stats.csv"))
    rows = list(csv.DictReader(open(find(os.path.join(d, "**", "*kernel_trace.csv")))))
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    qs_rows = [x for x in rows if "qs::" in x["Kernel_Name"]]
    # the "off" variant runs right after the timed region and is the only user of the dense (non-eliding) widening
    # forward: the 15 library launches before its first launch are the last three steps of the timed region
    def dense_fwd(name):
        base = name.split("(")[0].rstrip()
        return "ew_widen_kernel" in base and base.endswith("false>")

    end = next((i for i, x in enumerate(qs_rows) if dense_fwd(x["Kernel_Name"])), len(qs_rows))
    tail = qs_rows[max(end - 15, 0):end]
    with open(os.path.join(OUT, f"{TAG}_bench_step_timeline.txt"), "w") as f:
        f.write("rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py\n")
        f.write("3 consecutive headline steps of the timed region (durations under the profiler; a gap holds a HIP event pair on sampled steps)\n\n")
        prev_end = None
        for x in tail:
            s_, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
            gap = 0.0 if prev_end is None else (s_ - prev_end) / 1e3
            name = x["Kernel_Name"].split("(")[0]
            f.write(f"{name:84s} dur={(e - s_) / 1e3:8.1f}us gap={gap:6.1f}us grid={x.get('Grid_Size_X', x.get('Grid_Size', '?')):>10s} wg={x.get('Workgroup_Size_X', x.get('Workgroup_Size', '?')):>4s} "
                    f"vgpr={x.get('VGPR_Count', '?')}\n")
            prev_end = e
    shutil.rmtree(d)

    # 2a. the headline alone (no configs, no variants, no CPU baseline): every row of this summary is a headline kernel, so its
    #     averages ARE the per-launch durations the roofline quotes (the summary of the plain command mixes in configs 2-4)
    d = os.path.join(OUT, "stats_headline")
    if sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "bench", "--",
           "python3", BENCH, "--no-configs", "--no-variants", "--no-cpu-baseline"]).returncode == 0:
        shutil.copy(find(os.path.join(d, "**", "*kernel_stats.csv")), os.path.join(OUT, f"{TAG}_headline_kernel_stats.csv"))
    shutil.rmtree(d, ignore_errors=True)

    # 2b. per-config kernel stats (BASELINE configs 2-4), each from its own command
    for cfg in ("config2", "config3", "config4"):
        d = os.path.join(OUT, "stats_" + cfg)
        if sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "bench", "--",
               "python3", BENCH, "--configs-only", cfg]).returncode == 0:
            shutil.copy(find(os.path.join(d, "**", "*kernel_stats.csv")), os.path.join(OUT, f"{TAG}_{cfg}_kernel_stats.csv"))
        shutil.rmtree(d, ignore_errors=True)

    pmc_passes(rec, [], "")
    print("value", rec["value"], "ms/step", rec["ms_per_step"], "frac", rec["roofline"]["frac"])


def gate_passes():
    """HBM traffic of a channels_last fp32 ReLU site (256x256x56x56) with and without the gate bitmap"""
    driver = os.path.join(ROOT, "tools", "gate_traffic.py")
    numel = 256 * 256 * 56 * 56
    kinds = {"fwd_plain": ("ew_widen_kernel<qs::ScalerFwdOp<0>, 0, 3", 8.0), "fwd_recording": ("ew_widen_kernel<qs::GateOp<", 8.125),
             "bwd_from_x": ("ste_relu_bwd_kernel<0, 0, 3, true, false, false>", 12.0),
             "bwd_from_bitmap": ("ste_relu_bwd_kernel<0, 0, 3, true, false, true>", 8.125)}
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, "pmc_gate_" + counter)
        assert sh(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "gate", "--",
                   "python3", driver]).returncode == 0
        src = find(os.path.join(d, "**", "*counter_collection.csv"))
        rows = [x for x in csv.DictReader(open(src)) if "qs::" in x["Kernel_Name"]]
        with open(os.path.join(OUT, f"{TAG}_pmc_{counter}_gate_counter_collection.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
        for x in rows:
            per.setdefault(x["Kernel_Name"].split("(")[0], {}).setdefault(counter, []).append(float(x["Counter_Value"]))
        shutil.rmtree(d)
    out = {"command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 tools/gate_traffic.py  (one pass per counter)",
           "site": "ReLU -> prune -> quantize, channels_last fp32 256x256x56x56, 50 % channel mask (the site behind a residual add)",
           "unit": "bytes per launch", "correction": "fetch bytes = FETCH_SIZE*1024*2, write bytes = WRITE_SIZE*1024 (see rNN_pmc_traffic.json)",
           "kernels": {}}
    for key, (sub, bpe) in kinds.items():
        names = [k for k in per if sub in k]
        assert len(names) == 1, (sub, list(per))
        fk = sum(per[names[0]]["FETCH_SIZE"]) / len(per[names[0]]["FETCH_SIZE"])
        wk = sum(per[names[0]]["WRITE_SIZE"]) / len(per[names[0]]["WRITE_SIZE"])
        hbm, algo = int(fk * 1024 * 2 + wk * 1024), int(bpe * numel)
        out["kernels"][key] = {"kernel": names[0], "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1), "hbm_bytes": hbm,
                               "algorithmic_bytes_per_elem": bpe, "algorithmic_bytes": algo, "ratio": round(hbm / algo, 4)}
    with open(os.path.join(OUT, f"{TAG}_pmc_traffic_gate.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: (v["hbm_bytes"], v["ratio"]) for k, v in out["kernels"].items()}))


def pmc_passes(rec, extra, suffix):
    # 3. PMC passes, one counter each (never combined with other trace domains)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, "pmc_" + counter)
        assert sh(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "bench", "--",
                   "python3", BENCH, "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-configs", "--no-variants"] + extra).returncode == 0
        src = find(os.path.join(d, "**", "*counter_collection.csv"))
        rows = [x for x in csv.DictReader(open(src)) if "qs::" in x["Kernel_Name"]]
        rows = rows[-8 * 5:]                                     # the 8 timed steps, 5 launches each
        with open(os.path.join(OUT, f"{TAG}_pmc_{counter}{suffix}_counter_collection.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
        for x in rows:
            per.setdefault(x["Kernel_Name"].split("(")[0], {}).setdefault(counter, []).append(float(x["Counter_Value"]))
        shutil.rmtree(d)

    def pick(sub):
        names = [k for k in per if sub in k]
        assert len(names) == 1, (sub, list(per))
        return names[0]

    kept = rec["config"]["kept_channel_fraction"]
    bpe = rec["config"]["algorithmic_bytes_per_elem"]
    kernels = {"apply_fwd": (pick("ew_widen_kernel<qs::ScalerFwdOp"), int(round(bpe["apply_fwd"] * NUMEL))),
               "apply_bwd": (pick("SteBwdOp"), int(round(bpe["apply_bwd"] * NUMEL))),
               "stats": (pick("mean_outer_vec_kernel"), 2 * NUMEL)}
    traffic = {
        "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 8 --warmup 3 "
                   "--no-cpu-baseline --no-configs --no-variants" + "".join(" " + e for e in extra) + "  (one pass per counter; tools/refresh_profiles.py)",
        "mode": rec["config"]["elide_pruned"], "kept_channel_fraction": kept,
        "note": "algorithmic_bytes are mask-aware: the apply forward of the default mode reads only the kept channels "
                "((2*kept + 4) B/elem); FETCH_SIZE must show the same drop",
        "unit": "bytes per launch",
        "correction": "FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a wide coalesced streaming "
                      "read at exactly half its bytes (MI355X_MICROARCH.md, HBM section), so fetch bytes = FETCH_SIZE*1024*2.  "
                      "That section calibrates 16-byte/lane loads, which is what all three kernels issue; cross-check on the "
                      "statistics kernel's known input: FETCH_SIZE*1024*2 reproduces the 411,041,792-byte bf16 tensor.  "
                      "WRITE_SIZE is used as reported.",
        "kernels": {},
    }
    for key, (name, algo) in kernels.items():
        fk = sum(per[name]["FETCH_SIZE"]) / len(per[name]["FETCH_SIZE"])
        wk = sum(per[name]["WRITE_SIZE"]) / len(per[name]["WRITE_SIZE"])
        hbm = int(fk * 1024 * 2 + wk * 1024)
        traffic["kernels"][key] = {"kernel": name, "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                                   "hbm_bytes": hbm, "algorithmic_bytes": algo, "ratio": round(hbm / algo, 4)}
    with open(os.path.join(OUT, f"{TAG}_pmc_traffic{suffix}.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps({k: v["ratio"] for k, v in traffic["kernels"].items()}))


if __name__ == "__main__":
    main()
