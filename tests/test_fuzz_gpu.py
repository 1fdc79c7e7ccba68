"""Fixed-seed runs of the randomised differential tests under tests/fuzz/ (site state machines against the oracle, CPU path
against HIP path of whole modules and of the functional API).  A few hundred random configurations per run; the tools
themselves take a case count and a seed for longer campaigns."""
import importlib.util
import os
import random

import pytest

import qsparse_amd as qs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "fuzz", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(autouse=True)
def _quiet_and_restore():
    import torch
    before = {k: qs.get_qsparse_option(k) for k in ("log_on_created", "log_during_train", "fold_relu", "preserve_dtype", "graph_safe", "elide_pruned", "relu_gate")}
    threads = torch.get_num_threads()
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    torch.set_num_threads(1)      # ATen's channels_last reductions depend on the thread split (tests/fuzz/fuzz_parity.py)
    yield
    torch.set_num_threads(threads)
    qs.set_qsparse_options(**before)


def test_sites_against_oracle_state_machines():
    fz = _load("fuzz_parity")
    rng = random.Random(2024)
    bad = [r for r in (fz.one_case(rng, i) for i in range(120)) if r not in ("ok", None)]
    assert not bad, bad[:3]


def test_modules_cpu_path_against_hip_path():
    fz = _load("fuzz_cpu_gpu")
    rng = random.Random(2025)
    bad = [r for r in (fz.one_case(rng, i) for i in range(80)) if r not in ("ok", None)]
    assert not bad, bad[:3]


def test_converted_weight_networks_cpu_path_against_hip_path():
    """`convert(model, prune(...), weight_layers=[...])` (+ `convert(model, quantize(...), ...)`) networks only: the multi-tensor
    pruned-weight route (qs_multi_stage_mean / magnitude / mask_refresh) against the CPU path -- which tests/fuzz/fuzz_reference.py
    holds against the real reference on the same cases"""
    fz = _load("fuzz_cpu_gpu")
    fz.FORCE_WHAT = "net"
    fz.ENGAGED[0] = 0
    rng = random.Random(2028)
    bad = [r for r in (fz.one_case(rng, i) for i in range(100)) if r not in ("ok", None)]
    assert not bad, bad[:3]
    assert fz.ENGAGED[0] >= 30, fz.ENGAGED[0]     # (Adaptive quantizers and timeout=0 layers keep the inline path)


def test_token_major_and_rank5_sites_cpu_path_against_hip_path():
    """sites whose channel dim is not dim 1 -- (B, T, C) with masks over {last} / {1, 2} / {1} / ..., 5-d activations, last-dim
    channel-wise quantizers -- as `convert` builds them (reference sparse.py:231-239, util.py:92-99, quantize.py:100-107); the
    (B, T, C) + {last} ones take the composite's layout 3"""
    fz = _load("fuzz_cpu_gpu")
    fz.FORCE_WHAT = "tok"
    rng = random.Random(2029)
    bad = [r for r in (fz.one_case(rng, i) for i in range(150)) if r not in ("ok", None)]
    assert not bad, bad[:3]


def test_functional_api_cpu_path_against_hip_path():
    fz = _load("fuzz_cpu_gpu")
    rng = random.Random(2026)
    bad = [r for r in (fz.one_functional(rng, i) for i in range(250)) if r not in ("ok", None)]
    assert not bad, bad[:3]


def test_modules_with_steady_state_steps_replayed_from_a_hipgraph():
    """the same module cases; the activation sites that reach a steady state capture a whole step (forward + backward) into a
    hipGraph and replay it on each further step's data (`QS_FUZZ_GRAPH` of the tool)"""
    fz = _load("fuzz_cpu_gpu")
    fz.GRAPH = True
    fz.GRAPHED[0] = 0
    rng = random.Random(2027)
    bad = [r for r in (fz.one_case(rng, i) for i in range(120)) if r not in ("ok", None)]
    assert not bad, bad[:3]
    assert fz.GRAPHED[0] >= 10, fz.GRAPHED[0]


def test_autocast_image_route_against_the_plain_route():
    """tests/fuzz/fuzz_image.py: random sites in front of real autocast consumers, with the image (the default) against
    `autocast_image=False`, everything the user can observe bit for bit -- consumer orders, late hooks, replacing hooks,
    retained gradients, autograd.grad with respect to the output, evaluation steps, in-place activations, fp16 autocast"""
    fz = _load("fuzz_image")
    fz.USED[0] = 0
    rng = random.Random(2029)
    bad = [r for r in (fz.one_case(rng, i) for i in range(150)) if r != "ok"]
    assert not bad, bad[:3]
    assert fz.USED[0] >= 60, fz.USED[0]
