#!/usr/bin/env python3
"""HBM traffic of the library's kernels INSIDE a network step, from the PMC counters (the in-model
statistics kernels had no counters, their traffic ratio was asserted, not measured).

    python3 tools/pmc_model.py <tag> <arch> <batch> [steps]      ->  gpurun_out/profiles/<tag>_pmc_model.json / .txt

Two rocprofv3 passes over tools/profile_config.py, one counter each (FETCH_SIZE, WRITE_SIZE; `--kernel-trace` is the only
trace domain next to `--pmc`, the combination the pool's gpurun permits), the last `steps` steady-state steps of each.  Bytes per kernel = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024: on gfx950
FETCH_SIZE tallies a wide coalesced streaming read (16 bytes per lane, what every kernel here issues) at half its bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section; checked on the headline statistics kernel's known input in
profiles/rNN_pmc_traffic.json).  Set against the algorithmic bytes of the step's kernel families (every data operand of a launch
once, dense, in the dtype / layout the site saw: bench.py::library_kernel_accounting through tools/profile_config.py)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from family_table import family          # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "profiles")


def main():
    tag, arch = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "resnet50"
    batch = sys.argv[3] if len(sys.argv) > 3 else "256"
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    per, meta = {}, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = f"/tmp/pmc_{tag}_{counter}"
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "cfg", "--",
                            "python3", os.path.join(ROOT, "tools", "profile_config.py"), arch, batch, str(steps)],
                           cwd="/tmp", env=env, text=True, capture_output=True)
        assert r.returncode == 0, r.stderr[-2000:]
        meta = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        src = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
        rows = [x for x in csv.DictReader(open(src)) if "qs::" in x["Kernel_Name"]]
        names = [x["Kernel_Name"].split("(")[0] for x in rows]
        per_step = meta["launches_per_step"]          # as the trace has them (see tools/family_table.py)
        for cand in range(8, len(names) // max(steps, 2) + 1):
            if all(names[-cand:] == names[-(k + 1) * cand:-k * cand] for k in range(1, steps)):
                per_step = cand
                break
        for x in rows[-steps * per_step:]:
            k = x["Kernel_Name"].split("(")[0]
            a = per.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
            a[counter] += float(x["Counter_Value"])
            if counter == "FETCH_SIZE":
                a["launches"] += 1
        shutil.rmtree(d, ignore_errors=True)
    fams, kernels = {}, {}
    for k, a in per.items():
        hbm = (a["FETCH_SIZE"] * 1024 * 2 + a["WRITE_SIZE"] * 1024) / steps
        kernels[k] = {"family": family(k), "launches_per_step": a["launches"] / steps, "fetch_GB_per_step": round(a["FETCH_SIZE"] * 2048 / steps / 1e9, 4),
                      "write_GB_per_step": round(a["WRITE_SIZE"] * 1024 / steps / 1e9, 4), "hbm_GB_per_step": round(hbm / 1e9, 4)}
        f = fams.setdefault(family(k), 0.0)
        fams[family(k)] = f + hbm
    out = {"command": f"rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 tools/profile_config.py {arch} {batch} {steps}  "
                      "(one pass per counter; tools/pmc_model.py)",
           "config": f"{arch} batch {batch}, channels_last, bf16 autocast, default options, last {steps} steps",
           "correction": "bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 tallies a 16-byte-per-lane streaming read at half its bytes)",
           "families": {}, "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_GB_per_step"]))}
    for f, hbm in sorted(fams.items(), key=lambda kv: -kv[1]):
        algo = meta["families"].get(f, {}).get("GB", 0.0)
        out["families"][f] = {"hbm_GB_per_step": round(hbm / 1e9, 3), "algorithmic_GB_per_step": algo,
                              "ratio": round(hbm / 1e9 / algo, 4) if algo else None}
    with open(os.path.join(OUT, f"{tag}_pmc_model.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(OUT, f"{tag}_pmc_model.txt"), "w") as fh:
        fh.write(out["config"] + "\n" + out["command"] + "\n" + out["correction"] + "\n\n")
        fh.write(f"{'family':12s} {'HBM GB/step':>12s} {'algorithmic':>12s} {'ratio':>7s}\n")
        for f, v in out["families"].items():
            fh.write(f"{f:12s} {v['hbm_GB_per_step']:12.3f} {v['algorithmic_GB_per_step']:12.3f} {v['ratio'] if v['ratio'] is not None else 0:7.3f}\n")
        fh.write(f"\n{'kernel':100s} {'launches':>8s} {'fetch GB':>9s} {'write GB':>9s}\n")
        for k, v in out["kernels"].items():
            fh.write(f"{k[:100]:100s} {v['launches_per_step']:8.1f} {v['fetch_GB_per_step']:9.4f} {v['write_GB_per_step']:9.4f}\n")
    print(open(os.path.join(OUT, f"{tag}_pmc_model.txt")).read())


if __name__ == "__main__":
    main()
