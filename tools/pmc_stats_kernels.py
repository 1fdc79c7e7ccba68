#!/usr/bin/env python3
"""Occupancy / stall counters of the statistics kernels ("say why channels_last statistics stop at
0.44-0.60 with counters").  Two modes:

  python3 tools/pmc_stats_kernels.py run            the workload alone: channels_last (`qs_mean_dim_cl`) and NCHW (`qs_mean_dim`,
                                                    the control) statistics of three ResNet-50 batch-256 activation shapes,
                                                    rotating inputs past the Infinity Cache
  python3 tools/pmc_stats_kernels.py [tag]          (GPU box, repo root) one `rocprofv3 --pmc` pass per counter group over that
                                                    workload (the program directly after `--`, NO trace domain next to the
                                                    counters) + a separate kernel-trace pass for the durations; writes the
                                                    per-kernel table to gpurun_out/profiles/<tag>_stats_kernels_pmc.txt.  A group
                                                    that fails marks the table INCOMPLETE and the tool exits non-zero.
  QS_PMC_DTYPE=f32: the float32 inputs of the sites behind residual adds instead of bf16.
"""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = os.environ.get("QS_PMC_DTYPE", "bf16") == "f32"
SHAPES = [(256, 256, 56, 56), (256, 512, 28, 28)] if F32 else [(256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14)]
GROUPS = {
    "sq": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
           "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU"],
    "sq2": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
            "SQ_WAVES", "SQ_WAVE_CYCLES"],
    "grbm": ["GRBM_GUI_ACTIVE"],
    "tcp": ["TCP_PENDING_STALL_CYCLES", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
    "tcc": ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
}


def workload():
    sys.path.insert(0, ROOT)
    import torch
    from qsparse_amd import _hip

    lib = _hip.load()
    for shp in SHAPES:
        N, C, H, W = shp
        dt, code, size = (torch.float32, 0, 4) if F32 else (torch.bfloat16, 1, 2)
        nrot = max(2, min(6, int(9e8 // (N * C * H * W * size))))
        xs = [torch.randn(shp, device="cuda").to(dt) for _ in range(nrot)]
        stage = torch.empty(C * H * W, device="cuda", dtype=dt)
        part = torch.empty(C * H * W, device="cuda")
        amax = torch.zeros(C * 32, device="cuda")
        for it in range(12):
            x = xs[it % nrot]
            assert lib.qs_mean_dim_cl(x.data_ptr(), stage.data_ptr(), N, H * W, C, code, code, 1 | 4, None, part.data_ptr(), None) == 0
        for it in range(12):
            x = xs[it % nrot]
            assert lib.qs_mean_dim(x.data_ptr(), stage.data_ptr(), 1, N, C * H * W, code, code, 1 | 4, None, amax.data_ptr(), 32, H * W, C, None) == 0
        torch.cuda.synchronize()
        del xs


def short(name):
    base = name.split("(")[0]
    return base.replace("void qs::", "").replace("qs::", "")[:64]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    out_dir = os.path.join(ROOT, "gpurun_out", "profiles")
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    me = os.path.abspath(__file__)
    per = {}        # kernel -> launch index -> counter -> value
    dispatches = {}  # kernel -> counter group -> launches seen in that pass
    order = []
    notes = []
    for group, counters in GROUPS.items():
        d = f"/tmp/pmc_stats_{group}"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "p", "--", "python3", me, "run"]
        print("+", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd="/tmp", env=env, text=True, capture_output=True)
        if r.returncode != 0:
            notes.append(f"group {group} ({' '.join(counters)}): rocprofv3 exit {r.returncode}: {r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ''}")
            continue
        hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not hits:
            notes.append(f"group {group}: no counter_collection.csv")
            continue
        seen = {}
        for row in csv.DictReader(open(hits[0])):
            if "qs::" not in row["Kernel_Name"]:
                continue
            k = short(row["Kernel_Name"])
            did = row.get("Dispatch_Id", "")
            key = (k, row.get("Grid_Size", ""))
            if key not in order:
                order.append(key)
            idx = seen.setdefault(key, {}).setdefault(did, len(seen[key]))
            per.setdefault(key, {}).setdefault(idx, {})[row["Counter_Name"]] = float(row["Counter_Value"])
        for key, dids in seen.items():           # every pass must have seen the same launches: the averages pair them by position
            dispatches.setdefault(key, {})[group] = len(dids)
        shutil.rmtree(d, ignore_errors=True)
    # durations from a plain kernel trace
    d = "/tmp/pmc_stats_trace"
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", "python3", me, "run"], cwd="/tmp", env=env,
                   text=True, capture_output=True)
    dur = {}
    hits = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if hits:
        for row in csv.DictReader(open(hits[0])):
            if "qs::" in row["Kernel_Name"]:
                key = (short(row["Kernel_Name"]), row.get("Grid_Size", row.get("Grid_Size_X", "")))
                dur.setdefault(key, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    lines = ["rocprofv3 --pmc <group> -- python3 tools/pmc_stats_kernels.py run   (one pass per group, no trace domain next to the counters; "
             "durations from a separate --kernel-trace pass; averages over the last 8 of 12 launches)",
             f"shapes ({'float32' if F32 else 'bf16'}, batch 256): " + ", ".join("x".join(map(str, s)) for s in SHAPES) + "; channels_last = mean_cl_*, NCHW control = mean_outer_vec_kernel",
             "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); per-wave = / SQ_WAVES", ""]
    for key, by_group in dispatches.items():
        if len(set(by_group.values())) > 1:
            notes.append(f"{key[0]} grid={key[1]}: the passes saw different launch counts {by_group} -- its per-position averages mix launches")
    if notes:
        lines += ["INCOMPLETE: a counter group failed or the passes disagree -- see the notes"] + notes + [""]
    for key in order:
        launches = per[key]
        idxs = sorted(launches)[4:] or sorted(launches)
        avg = {}
        for i in idxs:
            for c, v in launches[i].items():
                avg.setdefault(c, []).append(v)
        avg = {c: sum(v) / len(v) for c, v in avg.items()}
        ds = dur.get(key, [])
        ds = ds[4:] or ds
        us = sum(ds) / len(ds) if ds else float("nan")
        lines.append(f"{key[0]}  grid={key[1]}  avg duration {us:.1f} us")
        w = avg.get("SQ_WAVES", 0)
        for c in sorted(avg):
            extra = ""
            if w and c.startswith("SQ_") and c != "SQ_WAVES" and ("CYCLES" in c or "WAIT" in c or "ACTIVE" in c or "INSTS" in c):
                extra = f"   per wave {avg[c] / w:12.1f}"
            lines.append(f"    {c:32s} {avg[c]:16.0f}{extra}")
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            parts = [(c, avg.get(c, 0) / wc) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")]
            lines.append("    share of wave cycles: " + ", ".join(f"{c} {p:.3f}" for c, p in parts))
        lines.append("")
    path = os.path.join(out_dir, f"{tag}_stats_kernels_pmc.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if notes:
        sys.exit(1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        workload()
    else:
        main()
