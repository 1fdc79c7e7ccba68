"""Drop-in check against the reference's OWN test-suite: when the read-only reference checkout is present (the build
container), its tests/ directory -- test_quantize.py, test_sparse.py, test_convert.py, test_fuse.py, test_util.py,
24 tests -- is run unmodified with `import qsparse` resolving to `qsparse_amd` (module aliasing in a child process;
nothing is copied, nothing is written into the checkout), and, seed by seed, next to the real reference.

The reference's tests draw unseeded random weights and inputs, and two of them are flaky by construction
(test_more_pruning_options: ties among L0 magnitudes; test_preload_state_dict: two scales an ulp apart around a rounding
boundary) -- under the reference itself they fail for about one seed in two.  So the check is the strongest one available:
for every seed the outcome of every test must be THE SAME as the reference's (it is: same passes, same failures), and
seeds on which the reference passes all 24 must pass all 24 here.  Skipped where /root/reference does not exist (GPU box).
"""
import os
import re
import subprocess
import sys
import textwrap

import pytest

REF = os.environ.get("QSPARSE_REFERENCE", "/root/reference")
REF_TESTS = os.path.join(REF, "tests")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUNNER = textwrap.dedent("""
    import importlib, sys
    sys.dont_write_bytecode = True
    if {alias}:
        sys.path.insert(0, {root!r})
        import qsparse_amd
        sys.modules["qsparse"] = qsparse_amd
        for sub in ("quantize", "sparse", "util", "convert", "fuse", "imitation", "common"):
            sys.modules["qsparse." + sub] = importlib.import_module("qsparse_amd." + sub)
    else:
        sys.path.insert(0, {ref!r})
    import numpy as np, torch
    torch.manual_seed({seed}); np.random.seed({seed})
    import pytest
    sys.exit(pytest.main([{tests!r}, "-p", "no:cacheprovider", "-q", "-rf", "--rootdir=/tmp", "-W", "ignore"]))
""")


def _run(alias, seed):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    code = RUNNER.format(alias=alias, root=ROOT, ref=REF, seed=seed, tests=REF_TESTS)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", env=env, timeout=900)
    failed = sorted(re.findall(r"^FAILED (\S+)", r.stdout, flags=re.M))
    m = re.search(r"(\d+) passed", r.stdout)
    assert m, (r.stdout[-2000:], r.stderr[-2000:])
    return failed, int(m.group(1))


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference checkout not present (GPU box)")
def test_reference_test_suite_has_the_same_outcome_as_under_the_reference_seed_by_seed():
    clean = 0
    for seed in (0, 2, 3):
        ours, ref = _run(True, seed), _run(False, seed)
        assert ours == ref, (seed, ours, ref)
        assert ours[1] + len(ours[0]) == 24
        clean += not ours[0]
    assert clean >= 1      # seeds 2 and 3 pass all 24 under torch 2.10; seed 0 fails test_more_pruning_options in both
