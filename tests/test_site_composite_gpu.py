"""The composite site calls (`qs_site_fwd` / `qs_site_bwd`: one FFI call per site and direction, include/qsparse_hip.h) over
everything they cover -- Scaler and Decimal quantizers (reference quantize.py:100-117, 189-272, 275-367), NCHW, channels_last
and 2-d `[N, C]` activations (the output of an nn.Linear; reference sparse.py:215-273 with `dimensions={1}`), with and without a
folded nn.ReLU -- against

  * the ORACLE (`oracle.PruneSim` / `oracle.QuantizeSim`) replaying the same inputs on the CPU: output, input gradient, mask,
    running magnitude, scale and counters at every step of the schedule, bit for bit;
  * the fine-grained route (the same entry points issued one by one from Python), which the composite must equal.

Each case asserts that the composite route really ran (call counts of `_hip.site_fwd` / `_hip.site_bwd`)."""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd import _hip, fused

pytestmark = pytest.mark.gpu

START, INTERVAL, REPS, TIMEOUT = 2, 2, 2, 3
STEPS = 9


FREEZE = dict(mask_refresh_interval=1, stop_mask_refresh=3)      # the mask freezes three steps after pruning started


def make_site(kind, relu, device, frozen=False):
    from qsparse_amd.fused import FusedPruneQuantize
    cb = qs.ScalerQuantizer() if kind == "scaler" else qs.DecimalQuantizer()
    p = qs.prune(sparsity=0.5, start=START, interval=INTERVAL, repetition=1 if frozen else REPS, dimensions={1},
                 callback=qs.MagnitudePruningCallback(**FREEZE) if frozen else None)
    q = qs.quantize(bits=4, timeout=TIMEOUT, channelwise=-1, callback=cb)
    return FusedPruneQuantize(nn.Sequential(nn.ReLU() if relu else nn.Identity(), p), q).to(device), p, q


def inputs(shape, dtype, channels_last, step):
    g = torch.Generator().manual_seed(100 + step)
    x = torch.randn(shape, generator=g) * torch.linspace(0.2, 3.0, shape[1]).view((1, -1) + (1,) * (len(shape) - 2))
    x = x.to(dtype)
    gr = torch.randn(shape, generator=g)
    if channels_last:
        x, gr = x.contiguous(memory_format=torch.channels_last), gr.contiguous(memory_format=torch.channels_last)
    return x, gr


class Counter:
    def __init__(self, monkeypatch, name):
        self.n = 0
        orig = getattr(_hip, name)

        def spy(*a, **k):
            self.n += 1
            return orig(*a, **k)

        monkeypatch.setattr(_hip, name, spy)


CASES = [
    ("scaler", (6, 16, 10, 12), torch.float32, False, True),
    ("decimal", (6, 16, 10, 12), torch.float32, False, True),
    ("decimal", (6, 16, 10, 12), torch.bfloat16, True, True),
    ("decimal", (4, 24, 7, 7), torch.bfloat16, False, False),
    ("decimal", (4, 24, 7, 7), torch.float16, True, False),
    ("scaler", (32, 48), torch.float32, False, True),
    ("scaler", (32, 48), torch.bfloat16, False, False),
    ("decimal", (17, 40), torch.bfloat16, False, True),
    ("scaler", (64, 1000), torch.float32, False, False),
    ("decimal", (5, 36), torch.float16, False, True),
    ("scaler", (16, 5000), torch.bfloat16, False, True),            # C in (4096, 8192]: the LDS-resident select, 8 channels per thread
    ("scaler", (4, 3000, 2, 2), torch.float32, False, False),       # C in (2048, 4096]
]
# the recipe's steady state: mask frozen (stop_mask_refresh passed), the scale still follows the data (QS_SITE_SCALE_ONLY)
FROZEN_CASES = [
    ("scaler", (6, 16, 10, 12), torch.float32, False, True),
    ("scaler", (6, 16, 10, 12), torch.bfloat16, True, True),
    ("decimal", (4, 24, 7, 7), torch.bfloat16, False, False),
    ("scaler", (300, 64, 4, 4), torch.bfloat16, True, True),        # channels_last, many rows: the two-stage reduction's workspace
    ("scaler", (300, 40), torch.float32, False, True),              # 2-d, many rows
    ("decimal", (17, 40), torch.float16, False, False),
]


@pytest.mark.parametrize("kind,shape,dtype,channels_last,relu,frozen",
                         [c + (False,) for c in CASES] + [c + (True,) for c in FROZEN_CASES])
def test_composite_site_equals_oracle_and_fine_grained(kind, shape, dtype, channels_last, relu, frozen, monkeypatch):
    dev = torch.device("cuda:0")
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    fwd_calls, bwd_calls = Counter(monkeypatch, "site_fwd"), Counter(monkeypatch, "site_bwd")
    site, p, q = make_site(kind, relu, dev, frozen)
    fine, pf, qf = make_site(kind, relu, dev, frozen)
    psim = O.PruneSim(0.5, [1], START, INTERVAL, 1 if frozen else REPS, False, **(FREEZE if frozen else {}))
    flags_seen = []
    real_site_fwd = _hip.site_fwd
    monkeypatch.setattr(_hip, "site_fwd", lambda *a, **k: (flags_seen.append(a[4]), real_site_fwd(*a, **k))[1])
    qsim = O.QuantizeSim(kind, 4, -1, TIMEOUT)
    real_plan = fused._site_plan
    for step in range(STEPS):
        training = step != STEPS - 2                    # one evaluation step inside the schedule, training again after it
        for m in (site, fine):
            m.train(training)
        x, gr = inputs(shape, dtype, channels_last, step)
        # oracle
        h = torch.relu(x) if relu else x
        n_before = psim.n_updates
        h = psim.step(h, training)
        y_ref = qsim.step(h.contiguous(), training)
        gin = qsim.grad(gr.contiguous(), x.dtype)         # (inactive steps: y is x's dtype and so is the gradient handed in)
        gin = psim.grad(gin, n_before >= psim.start)
        if relu:
            gin = torch.where(x.contiguous() > 0, gin, torch.zeros_like(gin))
        outs = []
        for m, composite in ((site, True), (fine, False)):
            monkeypatch.setattr(fused, "_site_plan", real_plan if composite else (lambda *a, **k: None))
            xd = x.to(dev).requires_grad_(True)
            y = m(xd)
            (gx,) = torch.autograd.grad(y, xd, gr.to(dev).to(y.dtype) if y.dtype != gr.dtype else gr.to(dev))
            outs.append((y.detach().cpu(), gx.cpu()))
        tag = (kind, shape, str(dtype), channels_last, relu, step)
        (y, gx), (y2, gx2) = outs
        assert y.dtype == y_ref.dtype and same(y.contiguous(), y_ref.contiguous()), ("output vs oracle", tag)
        assert gx.dtype == gin.dtype and same(gx.contiguous(), gin), ("input gradient vs oracle", tag)
        assert same(y, y2) and same(gx, gx2), ("composite vs fine-grained", tag)
        for pl, ql in ((p, q), (pf, qf)):
            assert same(pl.mask.detach().cpu(), psim.mask), ("mask", tag)
            assert pl._n_updates.item() == psim.n_updates and pl.callback.t.item() == psim.t, ("counters", tag)
            if psim.magnitude is not None:
                assert same(pl.callback.magnitude.detach().cpu(), psim.magnitude), ("magnitude", tag)
            assert same(ql.weight.detach().cpu(), qsim.weight), ("scale", tag)
            assert ql._n_updates.item() == qsim.n_updates and ql.callback.t == qsim.shared["t"], ("quantizer counters", tag)
    # live steps (quantizer and pruning both active, training) and the evaluation step went through the composite calls
    assert fwd_calls.n >= STEPS - TIMEOUT - 1 and bwd_calls.n >= STEPS - TIMEOUT - 1, (fwd_calls.n, bwd_calls.n)
    scale_only = sum(1 for f in flags_seen if f & _hip.SITE_SCALE_ONLY)
    assert (scale_only >= 2) if frozen else (scale_only == 0), flags_seen


def test_two_forwards_of_a_decimal_site_keep_their_own_step():
    """two live forwards before the backwards: each backward clamps with the power-of-two step of ITS forward"""
    dev = torch.device("cuda:0")
    site, p, q = make_site("decimal", True, dev)
    site.train()
    for step in range(5):                             # into the live regime
        x, _ = inputs((8, 16, 6, 6), torch.float32, False, step)
        site(x.to(dev))
    xs = []
    for scale in (0.05, 40.0):                        # very different magnitudes: the running scale, hence the step, moves
        x, gr = inputs((8, 16, 6, 6), torch.float32, False, 50)
        xd = (x * scale).to(dev).requires_grad_(True)
        w_before = q.weight.detach().clone()
        y = site(xd)
        dec = _hip.decimal_from_scale(q.weight.detach().view(-1)).item()
        xs.append((xd, y, gr.to(dev), dec, w_before))
    assert xs[0][3] != xs[1][3]
    for xd, y, gr, dec, _ in xs:
        (gx,) = torch.autograd.grad(y, xd, gr)
        clamped = O.ste_bwd(gr.cpu(), 4, 2.0 ** -dec)           # the gradient VALUES are clamped (quantize.py:120-131)
        masked = clamped * p.mask.detach().cpu().view(1, -1, 1, 1).to(clamped.dtype)      # g * mask: a signed zero on pruned channels
        assert same(gx.cpu(), torch.where(xd.detach().cpu() > 0, masked, torch.zeros_like(masked)))
