"""Activation sites whose channel dim is NOT dim 1: token-major `(B, T, C)` activations (transformer blocks) with
`prune(dimensions={2})` / `{1, 2}` / `{1}`, tensor-wise and last-dim channel-wise quantizers, and a 5-d activation -- the reference
builds its mask for any dim set (qsparse/sparse.py:231-239), averages one dim at a time (util.py:92-99: over B, rounded, then over
T, rounded -- NOT one mean over B*T) and quantizes any rank (quantize.py:100-107).

Every case is held, step by step through the schedule, against the ORACLE (`oracle.PruneSim` / `oracle.QuantizeSim` on the CPU):
output, input gradient, mask, running magnitude, scale, counters, bit for bit.  The `(B, T, C)` + `dimensions={2}` sites also run
through the composite calls (`qs_site_fwd` / `qs_site_bwd`, qs_site_plan layout 3) and must equal the fine-grained route; the test
asserts that the composite really ran.  A non-foldable activation (nn.GELU, what `convert(..., activation_layers=[nn.GELU])` wraps)
is ATen's own device kernel on both sides of the comparison."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd import _hip, fused, sparse
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
START, INTERVAL, REPS, TIMEOUT, STEPS = 2, 2, 2, 3, 9
FREEZE = dict(mask_refresh_interval=1, stop_mask_refresh=3)
ACTS = {"relu": (nn.ReLU, torch.relu), "identity": (nn.Identity, lambda t: t), "gelu": (nn.GELU, F.gelu),
        "relu6": (nn.ReLU6, lambda t: F.hardtanh(t, 0.0, 6.0)), "leaky": (lambda: nn.LeakyReLU(0.1), lambda t: F.leaky_relu(t, 0.1))}


def inputs(shape, dtype, step, cd):
    g = torch.Generator().manual_seed(700 + step)
    scale = torch.linspace(0.2, 3.0, shape[cd]).view([-1 if i == cd else 1 for i in range(len(shape))])
    x = (torch.randn(shape, generator=g) * scale).to(dtype)
    return x, torch.randn(shape, generator=g)


def act_on_device(act, x, dev):
    """(h on the CPU for the oracle, a function mapping the oracle's gradient w.r.t. h to the gradient w.r.t. x): the activation and
    its backward are ATen's DEVICE kernels here as in the module under test, so a transcendental (GELU) compares bit for bit"""
    xa = x.to(dev).requires_grad_(True)
    h = ACTS[act][1](xa)
    if act == "identity":
        return x, lambda gin: gin
    return h.detach().cpu(), lambda gin: torch.autograd.grad(h, xa, gin.to(dev))[0].cpu()


class Spy:
    def __init__(self, monkeypatch, name):
        self.flags = []
        orig = getattr(_hip, name)
        monkeypatch.setattr(_hip, name, lambda *a, **k: (self.flags.append(a[4]), orig(*a, **k))[1])


PAIRS = [
    # kind, shape, dtype, act, frozen
    ("scaler", (6, 10, 16), torch.float32, "relu", False),
    ("decimal", (5, 12, 24), torch.bfloat16, "identity", False),
    ("scaler", (4, 9, 40), torch.float16, "relu6", False),
    ("scaler", (7, 5, 33), torch.bfloat16, "leaky", False),          # C % 8 != 0
    ("scaler", (16, 50, 64), torch.bfloat16, "gelu", False),
    ("decimal", (6, 10, 16), torch.float32, "gelu", True),
    ("scaler", (300, 3, 48), torch.bfloat16, "relu", True),         # many rows: the frozen step's two-stage abs-max
    ("scaler", (4, 6, 3072), torch.bfloat16, "relu", False),         # C in (2048, 4096]: the select keeps 4 channels per thread in LDS
    ("decimal", (3, 4, 6000), torch.float32, "identity", False),     # C in (4096, 8192]: 8 per thread; T * C % 32 != 0: the atomics rider
    ("scaler", (2, 3, 9000), torch.float16, "relu", True),           # C > 8192: the select's global-memory passes
    ("scaler", (64, 197, 768), torch.bfloat16, "gelu", False),       # ViT-B/16 token grid
    ("scaler", (8, 128, 1024), torch.float16, "relu", False),
]


@pytest.mark.parametrize("kind,shape,dtype,act,frozen", PAIRS)
def test_token_major_pair_equals_oracle_and_fine_grained(kind, shape, dtype, act, frozen, monkeypatch):
    dev = torch.device("cuda:0")
    fwd, bwd = Spy(monkeypatch, "site_fwd"), Spy(monkeypatch, "site_bwd")

    def make():
        cb = qs.ScalerQuantizer() if kind == "scaler" else qs.DecimalQuantizer()
        p = qs.prune(sparsity=0.5, start=START, interval=INTERVAL, repetition=1 if frozen else REPS, dimensions={2},
                     callback=qs.MagnitudePruningCallback(**FREEZE) if frozen else None)
        q = qs.quantize(bits=4, timeout=TIMEOUT, channelwise=-1, callback=cb)
        return fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.Sequential(ACTS[act][0](), p), q)))[0].to(dev), p, q

    site, p, q = make()
    fine, pf, qf = make()
    assert type(site) is fused.FusedPruneQuantize
    psim = O.PruneSim(0.5, [2], START, INTERVAL, 1 if frozen else REPS, False, **(FREEZE if frozen else {}))
    qsim = O.QuantizeSim(kind, 4, -1, TIMEOUT)
    real_plan = fused._site_plan
    big = shape[0] * shape[1] * shape[2] > 1 << 20
    for step in range(5 if big else STEPS):
        training = step != STEPS - 2
        site.train(training), fine.train(training)
        x, gr = inputs(shape, dtype, step, 2)
        h, back = act_on_device(act, x, dev)
        n_before = psim.n_updates
        y_ref = qsim.step(psim.step(h, training), training)
        gin = psim.grad(qsim.grad(gr, x.dtype), n_before >= psim.start)
        gx_ref = back(gin)
        outs = []
        for m, composite in ((site, True), (fine, False)):
            monkeypatch.setattr(fused, "_site_plan", real_plan if composite else (lambda *a, **k: None))
            xd = x.to(dev).requires_grad_(True)
            y = m(xd)
            (gx,) = torch.autograd.grad(y, xd, gr.to(dev).to(y.dtype))
            outs.append((y.detach().cpu(), gx.cpu()))
        tag = (kind, shape, str(dtype), act, frozen, step)
        (y, gx), (y2, gx2) = outs
        assert same(y, y2) and same(gx, gx2), ("composite vs fine-grained", tag)
        assert y.dtype == y_ref.dtype and same(y, y_ref), ("output vs oracle", tag)
        assert gx.dtype == gx_ref.dtype and same(gx, gx_ref), ("input gradient vs oracle", tag)
        for pl, ql in ((p, q), (pf, qf)):
            assert pl.mask.shape == (1, 1, shape[2]) and same(pl.mask.detach().cpu(), psim.mask), ("mask", tag)
            assert pl._n_updates.item() == psim.n_updates and pl.callback.t.item() == psim.t, ("counters", tag)
            if psim.magnitude is not None:
                assert same(pl.callback.magnitude.detach().cpu(), psim.magnitude), ("magnitude", tag)
            assert same(ql.weight.detach().cpu(), qsim.weight), ("scale", tag)
            assert ql._n_updates.item() == qsim.n_updates and ql.callback.t == qsim.shared["t"], ("quantizer counters", tag)
    live_steps = (5 if big else STEPS - 1) - TIMEOUT
    assert len(fwd.flags) >= live_steps and len(bwd.flags) >= live_steps, (fwd.flags, bwd.flags)
    scale_only = sum(1 for f in fwd.flags if f & _hip.SITE_SCALE_ONLY)
    assert (scale_only >= 2) if frozen else (scale_only == 0), fwd.flags


PRUNES = [
    # shape, dtype, dims, act, composite expected
    ((6, 10, 16), torch.float32, {2}, "relu", True),
    ((5, 12, 24), torch.bfloat16, {2}, "identity", True),
    ((4, 9, 40), torch.float16, {2}, "gelu", True),
    ((7, 5, 33), torch.bfloat16, {2}, "leaky", True),
    ((32, 197, 768), torch.bfloat16, {2}, "gelu", True),
    ((6, 10, 16), torch.float32, {1, 2}, "relu", False),
    ((5, 12, 24), torch.bfloat16, {1, 2}, "identity", False),
    ((6, 10, 16), torch.bfloat16, {1}, "relu", False),              # (B, C, L): a Conv1d activation
    ((4, 9, 40), torch.float16, {0, 2}, "identity", False),
    ((1, 12, 24), torch.bfloat16, {2}, "relu", False),              # batch of one: one stage
    ((2, 8, 3, 5, 6), torch.bfloat16, {1}, "relu", False),          # 5-d
    ((2, 8, 3, 5, 6), torch.float32, {4}, "identity", False),
]


@pytest.mark.parametrize("shape,dtype,dims,act,composite", PRUNES)
def test_token_major_prune_only_equals_oracle(shape, dtype, dims, act, composite, monkeypatch):
    dev = torch.device("cuda:0")
    fwd = Spy(monkeypatch, "site_fwd")

    def make():
        site = nn.Sequential(ACTS[act][0](), qs.prune(sparsity=0.5, start=START, interval=INTERVAL, repetition=REPS, dimensions=dims))
        return fuse_prune_quantize_pairs(nn.Sequential(site))[0].to(dev)

    site, fine = make(), make()
    psim = O.PruneSim(0.5, sorted(dims), START, INTERVAL, REPS, False)
    real_plan = sparse._prune_plan
    cd = max(dims)
    for step in range(STEPS):
        training = step != STEPS - 3
        site.train(training), fine.train(training)
        x, gr = inputs(shape, dtype, step, cd)
        gr = gr.to(dtype)
        h, back = act_on_device(act, x, dev)
        n_before = psim.n_updates
        y_ref = psim.step(h, training)
        gx_ref = back(psim.grad(gr, n_before >= psim.start or not training))
        outs = []
        for m, comp in ((site, True), (fine, False)):
            monkeypatch.setattr(sparse, "_prune_plan", real_plan if comp else (lambda *a, **k: None))
            xd = x.to(dev).requires_grad_(True)
            y = m(xd)
            (gx,) = torch.autograd.grad(y, xd, gr.to(dev))
            outs.append((y.detach().cpu(), gx.cpu()))
        tag = (shape, str(dtype), sorted(dims), act, step)
        (y, gx), (y2, gx2) = outs
        assert same(y, y2) and same(gx, gx2), ("composite vs fine-grained", tag)
        assert y.dtype == y_ref.dtype and torch.equal(y.float(), y_ref.float()), ("output vs oracle", tag)
        assert torch.equal(gx.float(), gx_ref.float()), ("input gradient vs oracle", tag)
        if step >= START and act != "identity":
            assert same(y, y_ref) and same(gx, gx_ref), ("bits vs oracle", tag)
        for m in (site, fine):
            pl = m[1]
            assert same(pl.mask.detach().cpu(), psim.mask), ("mask", tag)
            assert pl._n_updates.item() == psim.n_updates and pl.callback.t.item() == psim.t, ("counters", tag)
            if psim.magnitude is not None:
                assert same(pl.callback.magnitude.detach().cpu(), psim.magnitude), ("magnitude", tag)
    no_quant = [f for f in fwd.flags if f & _hip.SITE_NO_QUANT]
    assert (len(no_quant) >= 5) if composite else (len(no_quant) == 0), fwd.flags


QUANTS = [
    # kind, shape, dtype, channelwise, act
    ("scaler", (6, 10, 16), torch.float32, -1, "relu"),
    ("decimal", (5, 12, 24), torch.bfloat16, -1, "gelu"),
    ("adaptive", (4, 9, 40), torch.float16, -1, "identity"),
    ("scaler", (1, 12, 24), torch.bfloat16, 2, "identity"),          # last-dim channel-wise: batch of one (B > 1 raises, below)
    ("decimal", (1, 12, 24), torch.float32, 2, "relu"),
    ("adaptive", (1, 9, 40), torch.bfloat16, 2, "identity"),
    ("adaptive", (4, 9, 40), torch.bfloat16, 1, "relu"),             # (B, C, L): the adaptive quantizer takes a batch on channel dim 1
    ("scaler", (1, 9, 40), torch.float16, 1, "identity"),
    ("scaler", (2, 8, 3, 5, 6), torch.bfloat16, -1, "relu"),
    ("scaler", (64, 197, 768), torch.bfloat16, -1, "gelu"),
]


@pytest.mark.parametrize("kind,shape,dtype,cw,act", QUANTS)
def test_token_major_quantize_only_equals_oracle(kind, shape, dtype, cw, act):
    dev = torch.device("cuda:0")
    cb = {"scaler": qs.ScalerQuantizer, "decimal": qs.DecimalQuantizer, "adaptive": qs.AdaptiveQuantizer}[kind]()
    site = fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(
        ACTS[act][0](), qs.quantize(bits=4, timeout=TIMEOUT, channelwise=cw, callback=cb))))[0].to(dev)
    qsim = O.QuantizeSim(kind, 4, cw, TIMEOUT)
    big = shape[0] * shape[1] * shape[2] > 1 << 20
    for step in range(5 if big else STEPS):
        training = step != STEPS - 2
        site.train(training)
        x, gr = inputs(shape, dtype, step, cw if cw >= 0 else len(shape) - 1)
        h, back = act_on_device(act, x, dev)
        y_ref = qsim.step(h, training)
        gx_ref = back(qsim.grad(gr.to(y_ref.dtype), x.dtype))
        xd = x.to(dev).requires_grad_(True)
        y = site(xd)
        (gx,) = torch.autograd.grad(y, xd, gr.to(dev).to(y.dtype))
        tag = (kind, shape, str(dtype), cw, act, step)
        assert y.dtype == y_ref.dtype and same(y.detach().cpu(), y_ref), ("output vs oracle", tag)
        assert same(gx.cpu(), gx_ref), ("input gradient vs oracle", tag)
        q = site[1]
        assert same(q.weight.detach().cpu(), qsim.weight), ("scale", tag)
        assert q._n_updates.item() == qsim.n_updates and q.callback.t == qsim.shared["t"], ("counters", tag)


def test_batched_last_dim_channelwise_scaler_raises_like_the_reference():
    """quantize.py:341-343: channel-wise Scaler / Decimal statistics of a batched activation raise -- on any channel dim"""
    q = qs.quantize(bits=8, timeout=1, channelwise=2).cuda().train()
    x = torch.randn(4, 6, 8, device="cuda")
    q(x)
    with pytest.raises(RuntimeError):
        q(x)


def test_batched_last_dim_channelwise_adaptive_raises_like_the_reference():
    """quantize.py:398-401: the batched channel-wise Adaptive statistics transpose the channel dim next to the batch and `.view` the
    result -- a RuntimeError for every channel dim other than 1 once the batch has more than one sample (an argument error: it
    fails for the contiguous tensor), reproduced on both devices; a batch of one is served"""
    for dev in ("cpu", "cuda"):
        q = qs.quantize(bits=4, timeout=1, channelwise=2, callback=qs.AdaptiveQuantizer()).to(dev).train()
        x = torch.randn(4, 6, 8, device=dev)
        q(x)
        with pytest.raises(RuntimeError, match="view size is not compatible"):
            q(x)
        q1 = qs.quantize(bits=4, timeout=1, channelwise=2, callback=qs.AdaptiveQuantizer()).to(dev).train()
        x1 = torch.randn(1, 6, 8, device=dev)
        q1(x1)
        assert q1(x1).shape == x1.shape


def test_token_major_site_through_convert_cpu_path_equals_hip_path():
    """`convert(model, prune(dimensions={2}), activation_layers=[nn.ReLU])` then `convert(model, quantize(), ...)` over a token-major
    MLP block: the CPU path of the package (held against the real reference by tests/fuzz/fuzz_reference.py) and the HIP path agree
    on every site's output, mask, magnitude and scale"""
    def build():
        torch.manual_seed(5)
        net = nn.Sequential(nn.ReLU(), nn.Identity(), nn.ReLU())
        net = qs.convert(net, qs.prune(sparsity=0.5, dimensions={2}, start=1, interval=1, repetition=2), activation_layers=[nn.ReLU],
                         log=False)
        return qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=2), activation_layers=[nn.ReLU], log=False)

    cpu, gpu = build(), build().cuda()
    for step in range(7):
        x, gr = inputs((6, 10, 32), torch.bfloat16, step, 2)
        outs = []
        for net, dev in ((cpu, "cpu"), (gpu, "cuda")):
            xd = x.to(dev).requires_grad_(True)
            y = net(xd)
            (gx,) = torch.autograd.grad(y, xd, gr.to(dev).to(y.dtype))
            outs.append((y.detach().cpu(), gx.cpu()))
        assert same(*[o[0] for o in outs]) and same(*[o[1] for o in outs]), step
        sc, sg = cpu.state_dict(), gpu.state_dict()
        assert list(sc) == list(sg)
        for k in sc:
            assert same(sc[k].cpu(), sg[k].cpu()), (k, step)
