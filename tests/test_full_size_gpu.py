"""Bit-exact parity at the BASELINE sizes, FULL tensors.

The headline tensor (256x256x56x56 bf16) and the config-2 tensor (256x64x56x56 bf16) run three LIVE training steps
through the HIP path -- a different input and gradient every step, mask refresh and running scale every step -- and
every element of every result is compared with the oracle's (reference sparse.py:99-122, 215-273; quantize.py:473-518,
327-349): output, input gradient, mask, running magnitude, scale, counters.  Grid arithmetic that only big tensors
reach (`WaveRows`, 32-bit group indices, `mean_lanes()` lane counts, the store-only waves of the eliding kernels) is
exactly what a slice cannot see.

  * elide_pruned = "off" / "forward": bit-exact, floats included.
  * elide_pruned = "all": numerically equal everywhere (+0.0 for the reference's -0.0 on pruned channels), bit-exact on
    the kept channels.
  * NCHW and channels_last (the layout the networks run in; its staged mean follows ATen's order for THAT layout).

The oracle runs with one intra-op thread (the contract of the staged mean's bits); its results are kept
on the GPU so that each is computed once and compared against all three elision modes.
"""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from oracle import qs_oracle as O
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda"
STEPS = 4          # step 0 brings both operators up (mask refresh needs t > 0, the quantizer its timeout), then three live steps
HEADLINE = (256, 256, 56, 56)
CONFIG2 = (256, 64, 56, 56)

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def _inputs(shape, step, channels_last, signed):
    """SURVEY 8d's synthetic activation (per-channel scaled, bf16), a fresh one per step; `signed`: not yet rectified
    (the site's own ReLU does that)"""
    g = torch.Generator(device=DEV).manual_seed(1000 + step)
    x = torch.randn(shape, generator=g, device=DEV)
    if not signed:
        x.relu_()
    x *= torch.linspace(0.25, 4.0, shape[1], device=DEV).view(1, -1, 1, 1)
    x = x.to(torch.bfloat16)
    gout = torch.randn(shape, generator=g, device=DEV)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
        gout = gout.contiguous(memory_format=torch.channels_last)
    return x, gout


def _bits_equal(a, b):
    a, b = a.contiguous(), b.contiguous()
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype == torch.bool:
        return bool((a == b).all())
    it = {4: torch.int32, 2: torch.int16}[a.element_size()]
    return bool(torch.equal(a.view(it), b.view(it)))


class _Ref:
    """what the oracle produced at one step, parked on the GPU"""
    __slots__ = ("y", "gx", "mask", "magnitude", "scale", "p_n", "p_t", "q_n", "q_t")


def _oracle_pair(shape, channels_last, relu):
    refs = []
    ps, qsim = O.PruneSim(0.75, [1], 0, 1, 1, False), O.QuantizeSim("scaler", 4, -1, 1)
    assert torch.get_num_threads() == 1
    for s in range(STEPS):
        x, gout = _inputs(shape, s, channels_last, relu)
        xc, gc = x.cpu(), gout.cpu()                       # (a channels_last tensor stays channels_last)
        h = torch.relu(xc) if relu else xc
        n_before = ps.n_updates
        hp = ps.step(h, True)
        # the reference's tensor-wise abs-max calls .view(1, -1), which a channels_last tensor refuses (quantize.py:333);
        # the maximum is order-independent, so the quantizer twin is fed the same values in NCHW order
        y = qsim.step(hp.contiguous(), True)
        gx = ps.grad(qsim.grad(gc.contiguous().to(y.dtype), torch.bfloat16), n_before >= ps.start)
        if relu:
            gx = torch.where(xc.contiguous() > 0, gx, torch.zeros_like(gx))       # ATen threshold_backward
        r = _Ref()
        r.y, r.gx = y.to(DEV), gx.to(DEV)
        r.mask, r.magnitude, r.scale = ps.mask.clone(), ps.magnitude.clone(), qsim.weight.clone()
        r.p_n, r.p_t, r.q_n, r.q_t = ps.n_updates, ps.t, qsim.n_updates, qsim.shared["t"]
        refs.append(r)
        del xc, gc, h, hp, y, gx
    return refs


@pytest.mark.parametrize("channels_last,relu", [(False, False), (True, False), (True, True)],
                         ids=["nchw", "channels_last", "channels_last-relu_folded"])
def test_headline_pair_three_live_steps_equal_the_oracle_on_the_full_tensor(channels_last, relu):
    refs = _oracle_pair(HEADLINE, channels_last, relu)
    assert 0.2 <= refs[-1].mask.float().mean().item() <= 0.3                # 75 % of the channels pruned
    try:
        for mode in ("off", "forward", "all"):
            qs.set_qsparse_options(elide_pruned=mode)
            act = nn.ReLU() if relu else nn.Identity()
            pair = nn.Sequential(nn.Sequential(act, qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1)),
                                 qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
            fuse_prune_quantize_pairs(pair)
            p, q = pair[0][1], pair[1]
            for s, r in enumerate(refs):
                x, gout = _inputs(HEADLINE, s, channels_last, relu)
                x.requires_grad_(True)
                y = pair(x)
                (gx,) = torch.autograd.grad(y, x, gout.to(y.dtype))
                tag = (mode, s)
                assert y.dtype == (torch.float32 if s > 0 else torch.bfloat16) and gx.dtype == torch.bfloat16, tag
                if channels_last:
                    assert y.is_contiguous(memory_format=torch.channels_last) and gx.is_contiguous(memory_format=torch.channels_last)
                if mode == "all":
                    # +0.0 where the reference's g * 0 / x * 0 carries a sign: equal as numbers, and bit-exact on kept channels
                    assert torch.equal(y, r.y) and torch.equal(gx, r.gx), tag
                    kept = r.mask.view(-1).to(DEV)
                    assert _bits_equal(y[:, kept], r.y[:, kept]) and _bits_equal(gx[:, kept], r.gx[:, kept]), tag
                else:
                    assert _bits_equal(y, r.y), ("output", tag)
                    assert _bits_equal(gx, r.gx), ("input gradient", tag)
                assert _bits_equal(p.mask.detach().cpu(), r.mask), ("mask", tag)
                assert _bits_equal(p.callback.magnitude.detach().cpu(), r.magnitude), ("magnitude", tag)
                assert _bits_equal(q.weight.detach().cpu(), r.scale), ("scale", tag)
                assert (p._n_updates.item(), p.callback.t.item(), q._n_updates.item(), q.callback.t) == \
                    (r.p_n, r.p_t, r.q_n, r.q_t), ("counters", tag)
                del y, gx, x, gout
            del pair
    finally:
        qs.set_qsparse_options(elide_pruned="forward")


@pytest.mark.parametrize("channels_last", [False, True], ids=["nchw", "channels_last"])
def test_config2_quantize8_three_live_steps_equal_the_oracle_on_the_full_tensor(channels_last):
    """BASELINE config 2: QuantizeLayer(bits=8, tensor-wise) alone on 256x64x56x56 bf16 (reference quantize.py:473-518)"""
    q = qs.quantize(bits=8, channelwise=-1, timeout=1).to(DEV).train()
    qsim = O.QuantizeSim("scaler", 8, -1, 1)
    for s in range(STEPS):                   # step 0 is the identity phase (timeout = 1)
        x, gout = _inputs(CONFIG2, s, channels_last, True)
        y_ref = qsim.step(x.cpu().contiguous(), True)
        if s == 0:
            assert q(x) is x and _bits_equal(y_ref.to(DEV), x)
            continue
        gx_ref = qsim.grad(gout.cpu().contiguous(), torch.bfloat16)
        x.requires_grad_(True)
        y = q(x)
        (gx,) = torch.autograd.grad(y, x, gout)
        assert _bits_equal(y.detach(), y_ref.to(DEV)), ("output", s)
        assert _bits_equal(gx, gx_ref.to(DEV)), ("input gradient", s)
        assert _bits_equal(q.weight.detach().cpu(), qsim.weight), ("scale", s)
        assert (q._n_updates.item(), q.callback.t) == (qsim.n_updates, qsim.shared["t"]), ("counters", s)
        del y, gx, y_ref, gx_ref


@pytest.mark.parametrize("kind", ["decimal", "adaptive"])
def test_config2_size_other_quantizers_equal_the_oracle_on_the_full_tensor(kind):
    """the other two quantizers of the reference on the config-2 tensor (256x64x56x56 bf16), live statistics:
    DecimalQuantizer (power-of-two scale from the running abs-max, truncation; quantize.py:31-63, 312-349) tensor-wise and
    AdaptiveQuantizer (running min/max lines per channel; :141-181, 393-430)"""
    if kind == "decimal":
        q = qs.quantize(bits=8, channelwise=-1, timeout=1, callback=qs.DecimalQuantizer()).to(DEV).train()
        qsim = O.QuantizeSim("decimal", 8, -1, 1)
    else:
        q = qs.quantize(bits=8, channelwise=1, timeout=1, callback=qs.AdaptiveQuantizer()).to(DEV).train()
        qsim = O.QuantizeSim("adaptive", 8, 1, 1)
    for s in range(3):
        x, gout = _inputs(CONFIG2, s, False, True)
        y_ref = qsim.step(x.cpu(), True)
        x.requires_grad_(True)
        y = q(x)
        if s == 0:
            assert y is x
            continue
        (gx,) = torch.autograd.grad(y, x, gout)
        gx_ref = qsim.grad(gout.cpu(), torch.bfloat16)
        assert _bits_equal(y.detach(), y_ref.to(DEV)), ("output", kind, s)
        assert _bits_equal(gx, gx_ref.to(DEV)), ("input gradient", kind, s)
        assert _bits_equal(q.weight.detach().cpu(), qsim.weight), ("state", kind, s)
        del y, gx, y_ref, gx_ref


def test_headline_pair_decimal_relu6_frozen_mask_on_the_full_tensor():
    """The composite routes added in round 4, at full size in one go: a DecimalQuantizer site (the power-of-two step of each
    forward, reference quantize.py:275-367), a folded nn.ReLU6 (hardtanh(0, 6), gate of `hardtanh_backward`), and the
    frozen-mask steps of a layerwise recipe (`stop_mask_refresh` passed: `QS_SITE_SCALE_ONLY`, sparse.py:107-116), channels_last,
    against the oracle with ATen's own activation on the CPU."""
    import torch.nn.functional as F
    steps, stop = 5, 2
    cbkw = dict(mask_refresh_interval=1, stop_mask_refresh=stop)
    ps = O.PruneSim(0.75, [1], 0, 1, 1, False, **cbkw)
    qsim = O.QuantizeSim("decimal", 4, -1, 1)
    pair = nn.Sequential(nn.Sequential(nn.ReLU6(), qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1,
                                                             callback=qs.MagnitudePruningCallback(**cbkw))),
                         qs.quantize(bits=4, channelwise=-1, timeout=1, callback=qs.DecimalQuantizer())).to(DEV).train()
    fuse_prune_quantize_pairs(pair)
    p, q = pair[0][1], pair[1]
    assert torch.get_num_threads() == 1
    flags = []
    from qsparse_amd import _hip
    real = _hip.site_fwd
    _hip.site_fwd = lambda *a, **k: (flags.append(a[4]), real(*a, **k))[1]
    try:
        for s in range(steps):
            x, gout = _inputs(HEADLINE, s, True, True)
            x = (x.float() * 2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)      # (so that the upper bound 6 is reached)
            xc, gc = x.cpu(), gout.cpu()
            h = F.hardtanh(xc, 0.0, 6.0)
            n_before = ps.n_updates
            y_ref = qsim.step(ps.step(h, True).contiguous(), True)
            gx_ref = ps.grad(qsim.grad(gc.contiguous().to(y_ref.dtype), torch.bfloat16), n_before >= ps.start)
            xcc = xc.contiguous()
            gx_ref = torch.where((xcc > 0) & (xcc < 6), gx_ref, torch.zeros_like(gx_ref))              # hardtanh_backward
            x.requires_grad_(True)
            y = pair(x)
            (gx,) = torch.autograd.grad(y, x, gout.to(y.dtype))
            assert _bits_equal(y.detach(), y_ref.to(DEV)), ("output", s)
            assert _bits_equal(gx, gx_ref.to(DEV)), ("input gradient", s)
            assert _bits_equal(p.mask.detach().cpu(), ps.mask) and _bits_equal(p.callback.magnitude.detach().cpu(), ps.magnitude), s
            assert _bits_equal(q.weight.detach().cpu(), qsim.weight), ("scale", s)
            assert (p._n_updates.item(), p.callback.t.item(), q._n_updates.item(), q.callback.t) == \
                (ps.n_updates, ps.t, qsim.n_updates, qsim.shared["t"]), ("counters", s)
            del y, gx, x, gout, xc, gc, h, y_ref, gx_ref, xcc
    finally:
        _hip.site_fwd = real
    # the last steps ran the frozen-mask composite, the ones before the live one
    assert sum(1 for f in flags if f & _hip.SITE_SCALE_ONLY) >= 2 and sum(1 for f in flags if not f & _hip.SITE_SCALE_ONLY) >= 1, flags
