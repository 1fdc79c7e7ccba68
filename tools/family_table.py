#!/usr/bin/env python3
"""Per-family table of the library's kernels inside a network step, from a rocprofv3 kernel trace of tools/profile_config.py:

    python3 tools/family_table.py <kernel_trace.csv> '<json line of profile_config.py>' > profiles/rNN_configN_family_table.txt

Takes the LAST steps x launches_per_step launches whose kernel lives in namespace qs:: (the K steady-state steps the script
ran last), sums their durations per kernel family and per kernel, and sets the family sums against the algorithmic bytes of a
step (the JSON line) and the 8 TB/s roofline."""
import csv
import json
import sys

PEAK = 8000.0  # GB/s


def family(name):
    base = name.split("(")[0]
    if "ste_relu_bwd_kernel" in base or "SteBwdOp" in base or "multi_ste_kernel" in base:
        return "apply_bwd"
    if "ChanMaskOp" in base or "mask_full" in base or "mask_bcast" in base:
        return "mask_apply"
    if "FwdOp" in base or "multi_quant_kernel" in base:
        return "apply_fwd"
    if any(k in base for k in ("mean_", "reduce_", "multi_absmax", "keys_init", "keys_to_float", "select_hist", "l0_flag", "mask_ge",
                               "kth_small", "select_scan", "select_init")):
        return "statistics"
    return "c_sized"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    meta = json.loads(sys.argv[2])
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    qs_rows = [r for r in rows if "qs::" in r["Kernel_Name"]]
    steps = meta["steps"]
    # launches per step as the TRACE has them: the event log behind `launches_per_step` keeps to the fine-grained entry
    # points, a real step may go through composite entry points that launch differently.  The K steady-state steps are
    # identical, so the per-step count is the smallest period of the kernel-name sequence at the end of the trace.
    names = [r["Kernel_Name"].split("(")[0] for r in qs_rows]
    per_step = meta["launches_per_step"]
    for cand in range(8, len(names) // max(steps, 2) + 1):
        if all(names[-cand:] == names[-(k + 1) * cand:-k * cand] for k in range(1, steps)):
            per_step = cand
            break
    meta = dict(meta, launches_per_step=per_step)
    want = steps * per_step
    tail = qs_rows[-want:]
    assert len(tail) == want, (len(qs_rows), want)
    fam, ker = {}, {}
    for r in tail:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        f = family(r["Kernel_Name"])
        a = fam.setdefault(f, [0.0, 0])
        a[0] += us
        a[1] += 1
        k = r["Kernel_Name"].split("(")[0]
        b = ker.setdefault(k, [0.0, 0, f])
        b[0] += us
        b[1] += 1
    # wall time of the K steps: first launch of the window to the end of the last kernel of the trace
    t0 = int(tail[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    opts = "default options (autocast image on: its bytes are counted)" if meta.get("autocast_image") else "autocast_image=False"
    print(f"{meta['arch']} batch {meta['batch']}, channels_last, bf16 autocast, {opts}: last {steps} steps of the trace, "
          f"{meta['launches_per_step']} library launches per step; step wall time ~{(t1 - t0) / 1e6 / steps:.2f} ms")
    print("(durations: rocprofv3 --kernel-trace; bytes: every data operand of a launch once, dense, in the dtype/layout the site saw)\n")
    total_ms = sum(v[0] for v in fam.values()) / 1e3 / steps
    total_gb = meta["algorithmic_GB_per_step"]
    print(f"{'family':12s} {'ms/step':>8s} {'launches':>9s} {'GB/step':>8s} {'GB/s':>8s} {'of 8 TB/s':>10s}")
    for f, (us, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        ms = us / 1e3 / steps
        gb = meta["families"].get(f, {}).get("GB", 0.0)
        print(f"{f:12s} {ms:8.3f} {n // steps:9d} {gb:8.3f} {gb / ms * 1e3 if gb else 0:8.0f} {gb / ms * 1e3 / PEAK if gb else 0:10.3f}")
    print(f"{'all':12s} {total_ms:8.3f} {want // steps:9d} {total_gb:8.3f} {total_gb / total_ms * 1e3:8.0f} {total_gb / total_ms * 1e3 / PEAK:10.3f}\n")
    # ATen's cast passes around the sites (the fp32-promotion tax), same window of the trace
    t_first = int(tail[0]["Start_Timestamp"])
    casts = {}
    for r in rows:
        if int(r["Start_Timestamp"]) >= t_first and ("bfloat16_copy_kernel" in r["Kernel_Name"] or "bfloat16tofloat32" in r["Kernel_Name"]
                                                      or "float16" in r["Kernel_Name"] and "copy_kernel" in r["Kernel_Name"]):
            k = "fp32 -> bf16 casts" if "bfloat16_copy_kernel" in r["Kernel_Name"] else "bf16 -> fp32 casts"
            a = casts.setdefault(k, [0.0, 0])
            a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a[1] += 1
            # parameter-sized launches (weights forward, weight gradients backward: at most a few hundred thousand work-items;
            # the smallest activation of these networks has millions) apart from the activation-sized ones
            if int(r.get("Grid_Size_X", r.get("Grid_Size", 1 << 30))) <= 800_000:
                b = casts.setdefault(k + ", parameter-sized launches among them", [0.0, 0])
                b[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                b[1] += 1
    for r in rows:      # ATen's own ReLU passes (networks with in-place ReLU modules, or fold_relu=False)
        name = r["Kernel_Name"]
        if int(r["Start_Timestamp"]) >= t_first and "qs::" not in name and ("threshold" in name or "clamp" in name or "relu" in name.lower()):
            k = "ReLU backward (threshold_backward)" if "threshold" in name else "ReLU forward (clamp / relu)"
            a = casts.setdefault(k, [0.0, 0])
            a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a[1] += 1
    for k, (us, n) in sorted(casts.items()):
        print(f"(not library) ATen {k}: {us / 1e3 / steps:.3f} ms/step in {n / steps:.1f} launches")
    print()
    print(f"{'kernel':100s} {'family':10s} {'ms/step':>8s} {'calls/step':>10s} {'avg us':>8s}")
    for k, (us, n, f) in sorted(ker.items(), key=lambda kv: -kv[1][0]):
        print(f"{k[:100]:100s} {f:10s} {us / 1e3 / steps:8.3f} {n / steps:10.1f} {us / n:8.1f}")
    # the last step launch by launch: a streaming launch's work-items x 8 elements is (within a workgroup) its tensor's size
    print("\nlast step, launch by launch (grid = work-items):")
    print(f"{'kernel':100s} {'grid':>12s} {'us':>8s}")
    for r in tail[-per_step:]:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        print(f"{r['Kernel_Name'].split('(')[0][:100]:100s} {grid:>12s} {us:8.1f}")


if __name__ == "__main__":
    main()
