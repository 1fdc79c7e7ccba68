"""Model-level plumbing for the BASELINE.json configurations: the --pq conversion recipe on the MNIST CNN
(config 0, CPU) and on residual networks (configs 2-4 as small GPU cases: fused path == unfused path,
bit for bit, through whole training steps including optimizer updates)."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import MnistNet, TokenNet, convert_pq, convert_pq_tokens, resnet18, resnet50
from qsparse_amd.quantize import QuantizeLayer
from qsparse_amd.sparse import PruneLayer

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def test_mnist_pq_recipe_trains_prunes_and_quantizes_on_cpu():
    from examples.mnist_pq import main

    model, losses, acc = main(["--device", "cpu", "--steps", "40", "--batch", "32"])
    assert losses[-1] < 0.5 * losses[0] and acc > 0.8
    prunes = [m for m in model.modules() if isinstance(m, PruneLayer)]
    quants = [m for m in model.modules() if isinstance(m, QuantizeLayer)]
    assert len(prunes) == 2 and len(quants) == 8      # 2 pruned ReLUs; input + 3 activations + 4 weights
    for p in prunes:
        kept = p.mask.sum().item()
        assert kept == p.mask.numel() // 4, (kept, p.mask.numel())      # 75 % of the channels pruned
    assert all(q._quantized for q in quants)
    sd = model.state_dict()
    assert any(k.endswith("_cur_sparsity") for k in sd) and any(k.endswith("quantize.weight") for k in sd)


def _train(model, steps, shape, classes, device, seed=0):
    torch.manual_seed(seed)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(seed)
    losses = []
    model.train()
    for _ in range(steps):
        x = torch.randn(shape, generator=g).to(device)
        y = torch.randint(0, classes, (shape[0],), generator=g).to(device)
        opt.zero_grad()
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_resnet_pq_fused_equals_unfused_on_gpu(arch):
    torch.manual_seed(0)
    if arch == "resnet18":
        base, shape = resnet18(num_classes=10, cifar_stem=True, width=16), (8, 3, 32, 32)
    else:
        base, shape = resnet50(num_classes=10, cifar_stem=False, width=8), (4, 3, 64, 64)
    torch.backends.cudnn.deterministic = True      # MIOpen: deterministic convolution algorithms
    runs = []
    for fuse in (False, False, True):
        model = convert_pq(copy.deepcopy(base), sparsity=0.5, bits=4, prune_start=2, prune_interval=2, repetition=2,
                           quant_timeout=3, fuse=fuse).cuda()
        losses = _train(model, 9, shape, 10, "cuda")
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        runs.append((losses, state))
    (lref, sref), (l0, s0), (l1, s1) = runs
    assert s0.keys() == s1.keys()
    if lref == l0:      # deterministic convolutions (the case on this stack): whole trajectories agree bit for bit
        assert l0 == l1, (l0, l1)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k
    else:               # run-to-run noise in MIOpen's backward: what does not depend on it must still agree exactly
        #                 (per-site bit-exactness against the oracle is tests/test_sites_gpu.py's job, on real inputs)
        for k in s0:
            if k.endswith(("_n_updates", "_cur_sparsity", ".t")):
                assert torch.equal(s0[k], s1[k]), k
            elif k.endswith(".mask"):
                assert s0[k].sum().item() == s1[k].sum().item(), k
        assert all(abs(a - b) < 0.25 * max(abs(a), 1.0) for a, b in zip(l0, l1)), (l0, l1)
    masks = [v for k, v in s0.items() if k.endswith(".mask")]
    assert masks and all(abs(m.float().mean().item() - 0.5) < 0.26 for m in masks)
    assert all(torch.isfinite(torch.tensor(l0)))


@pytest.mark.gpu
@pytest.mark.parametrize("act", [nn.GELU, nn.ReLU])
def test_token_major_network_pq_fused_equals_unfused_on_gpu(act, monkeypatch):
    """a transformer-style MLP encoder on (B, T, C) activations, hidden activations pruned along the last dim and quantized
    (reference sparse.py:231-239 builds that mask; convert.py:214-218 the sites): the fused sites -- qs_site_plan layout 3 on the
    composite calls -- against the same network module by module, whole training trajectories bit for bit (GEMMs are deterministic)"""
    from qsparse_amd import _hip
    calls = []
    real = _hip.site_fwd
    monkeypatch.setattr(_hip, "site_fwd", lambda plan_ref, *a, **k: (calls.append(plan_ref._obj.layout), real(plan_ref, *a, **k))[1])
    torch.manual_seed(0)
    base = TokenNet(num_classes=10, dim=32, hidden=64, depth=2, patch=4, act=act)
    runs = []
    for fuse in (False, True):
        model = convert_pq_tokens(copy.deepcopy(base), act=act, sparsity=0.5, bits=4, prune_start=2, prune_interval=2, repetition=2,
                                  quant_timeout=3, fuse=fuse).cuda()
        losses = _train(model, 9, (8, 3, 16, 16), 10, "cuda")
        runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (l0, s0), (l1, s1) = runs
    assert l0 == l1, (l0, l1)
    assert s0.keys() == s1.keys()
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    masks = [v for k, v in s1.items() if k.endswith(".mask") and v.numel() > 1]
    assert len(masks) == 2 and all(m.shape == (1, 1, 64) and 16 <= int(m.sum()) <= 48 for m in masks)
    assert calls and set(calls) == {3}, calls          # the fused run's sites went through the composite, token-major layout


@pytest.mark.gpu
def test_mnist_pq_on_gpu_matches_cpu_state_machine():
    """same recipe on the GPU: the schedule-driven state (sparsity, counters) matches the CPU run exactly;
    masks/scales depend on cuDNN-vs-CPU convolution rounding and are only checked for shape and sparsity."""
    from examples.mnist_pq import main

    mc, _, _ = main(["--device", "cpu", "--steps", "24", "--batch", "16"])
    mg, lg, accg = main(["--device", "cuda", "--steps", "24", "--batch", "16"])
    sc, sg = mc.state_dict(), mg.state_dict()
    assert sc.keys() == sg.keys()
    for k in sc:
        assert sc[k].shape == sg[k].shape and sc[k].dtype == sg[k].dtype, k
        if k.endswith(("_n_updates", "_cur_sparsity", ".t")):
            assert torch.equal(sc[k], sg[k].cpu()), k
    assert lg[-1] < lg[0]


@pytest.mark.gpu
def test_resnet_recipe_example_runs(capsys):
    """examples/resnet_pq_ddp.py (BASELINE configs 3-5 as a recipe): single process, channels_last, whole-step graph
    replay and batched weight quantizers together."""
    import qsparse_amd as qs
    from examples import resnet_pq_ddp
    try:
        resnet_pq_ddp.main(["--arch", "resnet18", "--batch", "16", "--steps", "3", "--warmup", "14", "--channels-last", "--graph"])
    finally:
        qs.set_qsparse_options(graph_safe=False, preserve_dtype=False, log_on_created=True, log_during_train=True, autocast_image=True)
    out = capsys.readouterr().out
    assert "images/s" in out and "graphed" in out
