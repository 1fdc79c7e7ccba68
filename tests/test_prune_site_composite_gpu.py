"""The composite call of a prune-ONLY activation site (`qs_site_fwd` / `qs_site_bwd` with QS_SITE_NO_QUANT: staged mean, select,
mask apply from one FFI call; reference qsparse/sparse.py:99-122, 215-273 behind an activation module, convert.py:214-218) against

  * the ORACLE (`oracle.PruneSim`) on the same inputs: output, input gradient, mask, running magnitude, counters, bit for bit;
  * the fine-grained route (`_importance` + `qs_pq_select` + `_MaskApply`, call by call), which it must equal."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd import _hip, sparse
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
START, INTERVAL, REPS, STEPS = 2, 2, 2, 10


def make(act, **cbkw):
    site = nn.Sequential(act, qs.prune(sparsity=0.5, start=START, interval=INTERVAL, repetition=REPS, dimensions={1},
                                       callback=qs.MagnitudePruningCallback(**cbkw)))
    return fuse_prune_quantize_pairs(nn.Sequential(site))[0].cuda().train()


CASES = [
    ((6, 16, 10, 12), torch.float32, False, "relu"),
    ((6, 16, 10, 12), torch.bfloat16, True, "relu"),
    ((4, 24, 7, 7), torch.bfloat16, False, "identity"),
    ((4, 24, 7, 7), torch.float16, True, "relu6"),
    ((32, 48), torch.float32, False, "relu"),
    ((17, 40), torch.bfloat16, False, "identity"),
    ((5, 300, 6, 6), torch.float32, False, "leaky"),
]


@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("shape,dtype,channels_last,act", CASES)
def test_prune_only_composite_equals_oracle_and_fine_grained(shape, dtype, channels_last, act, frozen, monkeypatch):
    mk = {"relu": nn.ReLU, "identity": nn.Identity, "relu6": nn.ReLU6, "leaky": lambda: nn.LeakyReLU(0.1)}[act]
    cbkw = dict(mask_refresh_interval=2, stop_mask_refresh=4) if frozen else {}
    site, fine = make(mk(), **cbkw), make(mk(), **cbkw)
    psim = O.PruneSim(0.5, [1], START, INTERVAL, REPS, False, **cbkw)
    calls = []
    real_fwd = _hip.site_fwd
    monkeypatch.setattr(_hip, "site_fwd", lambda *a, **k: (calls.append(a[4]), real_fwd(*a, **k))[1])
    real_plan = sparse._prune_plan
    for step in range(STEPS):
        training = step != STEPS - 3
        site.train(training), fine.train(training)
        g = torch.Generator().manual_seed(300 + step)
        x = (torch.randn(shape, generator=g) * torch.linspace(0.2, 3.0, shape[1]).view((1, -1) + (1,) * (len(shape) - 2))).to(dtype)
        gr = torch.randn(shape, generator=g).to(dtype)
        if channels_last:
            x, gr = x.contiguous(memory_format=torch.channels_last), gr.contiguous(memory_format=torch.channels_last)
        # oracle: ATen's activation on the CPU, then the prune state machine
        xa = x.clone().requires_grad_(True)
        h = {"relu": torch.relu, "identity": lambda t: t, "relu6": lambda t: F.hardtanh(t, 0.0, 6.0),
             "leaky": lambda t: F.leaky_relu(t, 0.1)}[act](xa)
        n_before = psim.n_updates
        y_ref = psim.step(h.detach(), training)
        gin = psim.grad(gr, n_before >= psim.start)
        (gx_ref,) = torch.autograd.grad(h, xa, gin) if act != "identity" else (gin,)
        outs = []
        for m, composite in ((site, True), (fine, False)):
            monkeypatch.setattr(sparse, "_prune_plan", real_plan if composite else (lambda *a, **k: None))
            xd = x.cuda().requires_grad_(True)
            y = m(xd)
            (gx,) = torch.autograd.grad(y, xd, gr.cuda())
            outs.append((y.detach().cpu(), gx.cpu()))
        tag = (shape, str(dtype), channels_last, act, frozen, step)
        (y, gx), (y2, gx2) = outs
        assert same(y, y2) and same(gx, gx2), ("composite vs fine-grained", tag)
        assert y.dtype == y_ref.dtype and torch.equal(y.float(), y_ref.float()), ("output vs oracle", tag)
        if step >= START:       # (the inactive steps are ATen's own device activation kernels: -0.0 details differ from the CPU's)
            assert same(y.contiguous(), y_ref.contiguous()), ("output vs oracle, bits", tag)
        assert torch.equal(gx.float(), gx_ref.float()), ("input gradient vs oracle", tag)
        for m in (site, fine):
            pl = m[1]
            assert same(pl.mask.detach().cpu(), psim.mask), ("mask", tag)
            assert pl._n_updates.item() == psim.n_updates and pl.callback.t.item() == psim.t, ("counters", tag)
            if psim.magnitude is not None:
                assert same(pl.callback.magnitude.detach().cpu(), psim.magnitude), ("magnitude", tag)
    no_quant = [f for f in calls if f & _hip.SITE_NO_QUANT]
    assert len(no_quant) >= (2 if frozen else 5), calls          # the composite really ran on the live / refreshing steps
