"""Mask-aware traffic elision (``set_qsparse_options(elide_pruned=...)``, qs_elementwise.h): the kernels that carry a
channel mask skip the loads of pruned channels.

  * quantizer forward ("forward", the default): bit-identical to the loading path and to the oracle for finite inputs,
    in every layout / dtype / ragged-row geometry the kernels distinguish;
  * backward and mask apply ("all", opt-in): exact on kept channels, numerically equal (+0.0 for the reference's -0.0)
    on pruned ones -- the 1e-6 contract of north_star holds trivially, the sign-of-zero difference is asserted to be
    the ONLY difference;
  * "off": NaN / Inf on a pruned channel reproduce the reference's INT_MIN * s (quirk B15).
"""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd import _hip
from qsparse_amd.fused import fuse_prune_quantize_pairs
from qsparse_amd.sparse import apply_mask

pytestmark = pytest.mark.gpu
DEV = "cuda"
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


@pytest.fixture(autouse=True)
def _restore_mode():
    yield
    qs.set_qsparse_options(elide_pruned="forward", preserve_dtype=False)


def gen(seed):
    return torch.Generator().manual_seed(seed)


SHAPES = [(4, 16, 8, 8),        # rows of 64: CM_ROW, whole waves
          (3, 24, 14, 14),      # rows of 196 = 4k: widening kernel keeps one channel per lane, 8-per-lane kernels CM_ELEM
          (5, 12, 7, 7),        # rows of 49: CM_ELEM, lanes straddle two channels
          (2, 8, 3, 1),         # rows shorter than a lane
          (9, 40, 56, 56),      # many full waves + reverse walk
          (6, 33),              # 2-d activation: channel dim innermost, C % 8 != 0
          (64, 48),             # 2-d, C % 8 == 0: CM_LAST
          (3, 5, 8, 8),         # whole waves + a partial last wave (960 elements), rows shorter than a wave
          (2, 3, 24, 24),       # rows of 576 >= 512 elements (wave-uniform mask look-up) with a partial last wave
          (3, 4, 32, 40)]       # rows of 1280 elements: waves inside one row, waves straddling two rows


def _mask(C, seed, keep=0.3):
    m = torch.rand(C, generator=gen(seed)) < keep
    m[0] = True
    if C > 9:
        m[1:9] = False          # a run of 8 pruned channels: CM_LAST lanes that are skipped entirely
    return m


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("kind", ["scaler", "decimal"])
def test_forward_elision_is_bit_identical(dtype, kind):
    for si, shape in enumerate(SHAPES):
        C = shape[1]
        x = (torch.randn(shape, generator=gen(10 + si)) * 3).to(dtype)
        mask = _mask(C, 20 + si)
        layouts = [x]
        if x.dim() == 4:
            layouts.append(x.to(memory_format=torch.channels_last))
        for xl in layouts:
            for pre_relu in (False, True):
                for preserve in (False, True):
                    if preserve and dtype == torch.float32:
                        continue
                    param = torch.tensor([[0.37]]) if kind == "scaler" else torch.tensor([[2.0]])
                    out = {}
                    for mode in ("off", "forward"):
                        qs.set_qsparse_options(elide_pruned=mode)
                        y, codes = _hip.quant_fwd(kind, xl.to(DEV), param.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV),
                                                  mask_channel_index=1, out_dtype=dtype if preserve else torch.float32,
                                                  pre_relu=pre_relu, want_codes=not preserve)
                        out[mode] = (y.cpu(), None if codes is None else codes.cpu())
                    assert same(out["off"][0], out["forward"][0]), (shape, dtype, kind, pre_relu, preserve, xl.stride())
                    if out["off"][1] is not None:
                        assert torch.equal(out["off"][1], out["forward"][1])
                    if not preserve:        # and both are the oracle's
                        h = (x.relu() if pre_relu else x) * mask.view([1, -1] + [1] * (x.dim() - 2))
                        ref = O.scaler_fwd(h, 4, param, -1) if kind == "scaler" else O.decimal_fwd(h, 4, param, -1)
                        assert same(out["forward"][0].contiguous(), ref), (shape, dtype, kind, pre_relu)


def test_forward_elision_per_channel_scale_and_masked_rows():
    """per-channel scales together with a channel mask (PARAM_PER_CHANNEL kernels)"""
    for si, shape in enumerate(SHAPES[:5]):
        C = shape[1]
        x = (torch.randn(shape, generator=gen(40 + si)) * 3).bfloat16()
        s = torch.rand(C, 1, generator=gen(50 + si)) + 0.1
        mask = _mask(C, 60 + si)
        out = {}
        for mode in ("off", "forward"):
            qs.set_qsparse_options(elide_pruned=mode)
            out[mode] = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), 1, torch.float32, chan_mask=mask.to(DEV),
                                       mask_channel_index=1)[0].cpu()
        assert same(out["off"], out["forward"]), shape
        assert same(out["forward"], O.scaler_fwd(x * mask.view(1, -1, 1, 1), 8, s, 1)), shape


def _signless_equal(a, b):
    """equal as numbers (so -0.0 == +0.0), NaNs in the same places"""
    a, b = a.float(), b.float()
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


@pytest.mark.parametrize("gdtype,xdtype", [(torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16),
                                            (torch.float32, torch.float32)])
def test_backward_and_mask_apply_elision_differ_only_in_the_sign_of_zero(gdtype, xdtype):
    for si, shape in enumerate(SHAPES):
        C = shape[1]
        g = torch.randn(shape, generator=gen(70 + si)).to(gdtype)
        x = torch.randn(shape, generator=gen(80 + si)).to(xdtype)
        mask = _mask(C, 90 + si)
        mview = mask.view([1, -1] + [1] * (len(shape) - 2))
        res = {}
        for mode in ("forward", "all"):
            qs.set_qsparse_options(elide_pruned=mode)
            gx = _hip.ste_bwd(g.to(DEV), torch.tensor([[0.37]], device=DEV), False, -1, -8.0, 7.0, False, xdtype,
                              chan_mask=mask.to(DEV), mask_channel_index=1).cpu()
            grelu = _hip.ste_relu_bwd(g.to(DEV), x.to(DEV), torch.tensor([[0.37]], device=DEV), False, -8.0, 7.0,
                                      mask.to(DEV)).cpu()
            ym = apply_mask(x.to(DEV), mview.to(DEV)).cpu()
            yr = apply_mask(x.to(DEV), mview.to(DEV), pre_relu=True).cpu()
            res[mode] = (gx, grelu, ym, yr)
        for a, b in zip(res["forward"], res["all"]):
            assert _signless_equal(a, b), (shape, gdtype, xdtype)
            keep = mview.expand(shape)
            assert same(a[keep], b[keep])                              # kept channels: bit-identical
            assert bool((b[~keep] == 0).all())                         # pruned channels: zero (of either sign: a pruned
            #                                                            element that shares a lane with kept ones is loaded)
        # the loading path is the reference's x * mask, sign of zero included
        assert same(res["forward"][2], x * mview)


def test_off_mode_reproduces_nan_on_pruned_channels_like_the_reference():
    x = torch.randn(2, 8, 8, 8, generator=gen(5)).bfloat16()
    x[0, 1, 0, 0] = float("nan")
    x[1, 1, 3, 3] = float("inf")
    mask = torch.ones(8, dtype=torch.bool)
    mask[1] = False
    s = torch.tensor([[0.25]])
    ref = O.scaler_fwd(x * mask.view(1, -1, 1, 1), 4, s, -1)
    qs.set_qsparse_options(elide_pruned="off")
    y_off = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1)[0].cpu()
    assert same(y_off, ref)
    assert y_off[0, 1, 0, 0].item() == float(-2 ** 31) * 0.25
    qs.set_qsparse_options(elide_pruned="forward")
    y_el = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1)[0].cpu()
    finite = torch.isfinite(x.float())
    assert same(y_el[finite], ref[finite])
    assert y_el[0, 1, 0, 0].item() == 0.0 and y_el[1, 1, 3, 3].item() == 0.0     # documented divergence (B15 inputs only)
    # a NaN scale still propagates through the pruned channels: f32(0) * NaN
    y_nan = _hip.quant_fwd("scaler", x.to(DEV), torch.tensor([[float("nan")]], device=DEV), -1, torch.float32,
                           chan_mask=mask.to(DEV), mask_channel_index=1)[0].cpu()
    assert bool(y_nan[:, 1].isnan().all())


def test_default_mode_elides_only_where_it_saves_traffic():
    """the default (`"forward"`) changed NaN / Inf on PRUNED channels everywhere.  Since round 4 it elides
    only where a pruned channel is a row that can be skipped -- an NCHW forward without gate recording.  A channels_last
    forward (a pruned channel is a 2-byte column, nothing is saved) and a gate-recording forward (every element is loaded
    anyway) follow the reference on non-finite inputs too: f32(INT_MIN) * s (quirk B15), through the functional entry point
    and through a whole convert-built ReLU -> prune -> quantize site in training and evaluation."""
    assert qs.get_qsparse_option("elide_pruned") == "forward"
    x = torch.randn(4, 16, 8, 8, generator=gen(6)).bfloat16()
    x[0, 1, 0, 0], x[1, 1, 3, 3], x[2, 5, 1, 1] = float("nan"), float("inf"), float("-inf")
    mask = torch.ones(16, dtype=torch.bool)
    mask[1] = mask[5] = False
    s = torch.tensor([[0.25]])
    ref = O.scaler_fwd(x * mask.view(1, -1, 1, 1), 4, s, -1)
    assert ref[0, 1, 0, 0].item() == float(-2 ** 31) * 0.25
    xcl = x.to(DEV).contiguous(memory_format=torch.channels_last)
    y = _hip.quant_fwd("scaler", xcl, s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1)[0]
    assert same(y.cpu().contiguous(), ref)                                       # channels_last: the reference's bits
    ref_relu = O.scaler_fwd(x.relu() * mask.view(1, -1, 1, 1), 4, s, -1)
    for xin in (x.to(DEV), xcl):                                                 # gate recording, either layout
        y, _, gate = _hip.quant_fwd("scaler", xin, s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1,
                                    pre_relu=True, want_gate=True)
        assert same(y.cpu().contiguous(), ref_relu)
    # the one place the default still deviates: NCHW, no gate (a row that is really skipped)
    y = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1)[0].cpu()
    assert y[0, 1, 0, 0].item() == 0.0 and same(y[torch.isfinite(x.float())], ref[torch.isfinite(x.float())])

    # whole site (the composite route) in evaluation: a trained ReLU -> prune -> quantize pair meets non-finite values on its
    # pruned channels; fused == module by module (x * mask, then the quantizer on the product) in channels_last
    def site():
        return nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1))

    fused, plain = fuse_prune_quantize_pairs(site().to(DEV)).train(), site().to(DEV).train()
    for step in range(5):
        xs = (torch.randn(4, 16, 8, 8, generator=gen(300 + step)) * torch.linspace(0.25, 4, 16).view(1, -1, 1, 1)).bfloat16()
        assert same(fused(xs.to(DEV)).cpu(), plain(xs.to(DEV)).cpu())
    fused.eval(), plain.eval()
    pruned = (~fused[0][1].mask.view(-1)).nonzero().view(-1).tolist()
    assert len(pruned) >= 3
    for j, c in enumerate(pruned[:3]):
        xs[j, c, j, j] = (float("nan"), float("inf"), float("-inf"))[j]
    xe = xs.to(DEV).contiguous(memory_format=torch.channels_last)
    ye = fused(xe).cpu().contiguous()
    assert same(ye, plain(xe).cpu().contiguous())
    assert ye[0, pruned[0], 0, 0].item() == float(-2 ** 31) * float(fused[1].weight)


@pytest.mark.parametrize("mode", ["forward", "all"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_fused_pair_trajectory_with_elision_vs_oracle(mode, channels_last):
    """the headline pair, live mask refresh, against the oracle's state machines: forward, masks, scales, magnitudes
    bit for bit in both modes; the backward bit for bit in "forward" mode and up to the sign of zero in "all"."""
    qs.set_qsparse_options(elide_pruned=mode)
    shape, C = (8, 32, 16, 16), 32
    sims = (O.PruneSim(0.75, [1], 1, 1, 2, False), O.QuantizeSim("scaler", 4, -1, 1))
    pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1, repetition=2)),
                         qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
    fuse_prune_quantize_pairs(pair)
    for s in range(8):
        x = (torch.randn(shape, generator=gen(100 + s)) * torch.linspace(0.25, 4, C).view(1, -1, 1, 1)).bfloat16()
        gout = torch.randn(shape, generator=gen(200 + s))
        xin = x.to(DEV)
        if channels_last:
            xin = xin.to(memory_format=torch.channels_last)
        xg = xin.requires_grad_(True)
        y = pair(xg)
        y.backward(gout.to(DEV))
        n_before = sims[0].n_updates
        y_ref = sims[1].step(sims[0].step(x.relu()), True)
        gin = sims[0].grad(sims[1].grad(gout, torch.bfloat16), n_before >= 1)
        gx_ref = torch.where(x > 0, gin, torch.zeros_like(gin))     # ATen's threshold_backward: +0 where x <= 0
        assert same(y.detach().cpu().contiguous(), y_ref), s
        assert same(pair[0][1].mask.cpu(), sims[0].mask), s
        assert same(pair[1].weight.cpu(), sims[1].weight), s
        if mode == "forward":
            assert same(xg.grad.cpu().contiguous(), gx_ref), s
        else:
            assert _signless_equal(xg.grad.cpu().contiguous(), gx_ref), s
            assert bool(((xg.grad.cpu().float() - gx_ref.float()).abs() <= 1e-6 * gx_ref.float().abs()).all())


def test_option_validation():
    with pytest.raises(ValueError):
        qs.set_qsparse_options(elide_pruned="backward")
    assert qs.get_qsparse_option("elide_pruned") == "forward"
