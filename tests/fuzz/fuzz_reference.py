#!/usr/bin/env python3
"""Randomised differential test of the package's CPU path against THE REAL REFERENCE (mlzxy/qsparse 2.0.1 imported from
/root/reference -- so it runs only in the build container, never on the GPU box, and is not part of the pytest suites).

    PYTHONPATH=.:tests python tests/fuzz/fuzz_reference.py [cases] [seed]

The configurations are those of tests/fuzz/fuzz_cpu_gpu.py (operators, wrapped layers, convert-built sites, NaN / Inf / -Inf in
inputs and raw weights): every case is built twice from the same random draws, once with the reference's `qsparse` and once with
`qsparse_amd`, and run on the CPU; outputs, gradients and every state_dict tensor must agree bit for bit (NaNs: in the same
places).  The package's CPU path is what the HIP path is compared with on the GPU box (fuzz_cpu_gpu.py), so this closes the
chain reference -> CPU path -> HIP path on the same random cases.  Cases that use an extension the reference does not have
(`saturate`) are skipped."""
import importlib.util
import os
import random
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("QSPARSE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def _load_harness():
    spec = importlib.util.spec_from_file_location("fuzz_cpu_gpu", os.path.join(ROOT, "tests", "fuzz", "fuzz_cpu_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    if not os.path.isdir(REF):
        print(f"{REF} not present: nothing to compare with")
        return 0
    torch.set_num_threads(1)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fz = _load_harness()
    package = fz.qs
    sys.path.insert(0, REF)
    import qsparse as reference
    assert reference.__version__ == "2.0.1" and os.path.realpath(reference.__file__).startswith(os.path.realpath(REF))
    reference.set_qsparse_options(log_on_created=False, log_during_train=False)
    package.set_qsparse_options(log_on_created=False, log_during_train=False)
    # the harness only ever calls the shared API; the route switches of the HIP path mean nothing here
    shim = types.SimpleNamespace(**{k: getattr(reference, k) for k in dir(reference) if not k.startswith("__")})
    shim.set_qsparse_options = lambda **kw: None
    real_run, real_build, last = fz.run, fz.build, {}

    def build(r):
        out = real_build(r)
        last["desc"] = out[0]
        return out

    fz.build = build
    rng = random.Random(seed)
    ran = fails = skipped = no_view = 0
    for i in range(cases):
        state = rng.getstate()
        captured = {}
        for label, mod in (("reference", shim), ("package", package)):
            rng.setstate(state)
            fz.qs = mod
            rec = []
            fz.run = lambda *a, **k: rec.append(a) or []         # record what one_case would run (the first call: "cpu")
            fz.one_case(rng, i)
            captured[label] = rec[0]
        fz.run, fz.qs = real_run, package
        if last["desc"].get("saturate"):        # an extension: the reference has no such switch
            skipped += 1
            continue
        outs = {}
        for label, mod in (("reference", shim), ("package", package)):
            fz.qs = mod
            args = list(captured[label])
            args[3] = "cpu"
            try:
                outs[label] = real_run(*args)
            except Exception as e:      # noqa: BLE001 -- both must fail alike
                outs[label] = ("raised", type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "")
        fz.qs = package
        a, b = outs["reference"], outs["package"]
        ran += 1
        if isinstance(a, tuple) and "view size is not compatible" in a[2] and (last["desc"]["channels_last"] or last["desc"].get("permute")):
            ran -= 1            # the reference's quantizer statistics `.view` their input (quantize.py:333): it cannot take a
            no_view += 1        # channels_last / permuted activation at all; the package can -- nothing to compare
            continue
        if isinstance(a, tuple) or isinstance(b, tuple):
            if not (isinstance(a, tuple) and isinstance(b, tuple) and a[1] == b[1]):
                fails += 1
                print("FAIL", i, "reference:", a if isinstance(a, tuple) else "ran", "package:", b if isinstance(b, tuple) else "ran",
                      last["desc"], flush=True)
            continue
        bad = None
        if len(a) != len(b):
            bad = "number of outputs"
        else:
            for (ka, va), (kb, vb) in zip(a, b):
                if (ka == kb and ka.startswith(("gx", "grad:")) and va.shape == vb.shape and va.dtype == vb.dtype
                        and bool(((va == vb) | (va.isnan() & vb.isnan())).all())):
                    continue        # (the sign of a gradient clamped to [-0, +0]: see fuzz_cpu_gpu.py)
                if ka != kb or not fz.same(va, vb):
                    bad = (ka, kb)
                    break
        if bad:
            fails += 1
            print("FAIL", i, bad, last["desc"], flush=True)
    print(f"fuzz reference-vs-package (CPU): {ran} cases, {fails} failures; skipped: {skipped} with an extension the reference lacks, "
          f"{no_view} channels_last / permuted cases the reference cannot run (seed {seed})")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
