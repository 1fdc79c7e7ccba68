"""Mask-aware traffic elision (``set_qsparse_options(elide_pruned=...)``, qs_elementwise.h): the kernels that carry a
channel mask skip the loads of pruned channels.

  * "forward" (the default) is EXACT: the quantizer forward of a prune -> quantize site elides through the elision mask its
    select wrote from this step's statistics (pruned channels holding a NaN / Inf are loaded), and nothing else elides --
    bit-identical to the loading path, the oracle and the CPU path for every input, NaN / Inf on pruned channels included
    (the reference's INT_MIN * s, quirk B15);
  * "all" (opt-in): every kernel that carries a channel mask -- forwards: bit-identical for finite inputs in every layout /
    dtype / ragged-row geometry the kernels distinguish, f32(0) * s for a NaN / Inf on a pruned channel; backward and mask
    apply: exact on kept channels, numerically equal (+0.0 for the reference's -0.0) on pruned ones -- the 1e-6 contract of
    north_star holds trivially, the sign-of-zero difference is asserted to be the ONLY difference;
  * "off": every element is loaded.
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
                    # "all": elision through the mask; "exact": the default mode handed an elision mask (all finite here: 0 / 1)
                    for mode in ("off", "all", "exact"):
                        qs.set_qsparse_options(elide_pruned="forward" if mode == "exact" else mode)
                        y, codes = _hip.quant_fwd(kind, xl.to(DEV), param.to(DEV), -1, torch.float32,
                                                  chan_mask=mask.to(torch.uint8).to(DEV) if mode == "exact" else mask.to(DEV),
                                                  mask_channel_index=1, out_dtype=dtype if preserve else torch.float32,
                                                  pre_relu=pre_relu, want_codes=not preserve, elision_mask=mode == "exact")
                        out[mode] = (y.cpu(), None if codes is None else codes.cpu())
                    for mode in ("all", "exact"):
                        assert same(out["off"][0], out[mode][0]), (shape, dtype, kind, pre_relu, preserve, xl.stride(), mode)
                        if out["off"][1] is not None:
                            assert torch.equal(out["off"][1], out[mode][1])
                    if not preserve:        # and all are the oracle's
                        h = (x.relu() if pre_relu else x) * mask.view([1, -1] + [1] * (x.dim() - 2))
                        ref = O.scaler_fwd(h, 4, param, -1) if kind == "scaler" else O.decimal_fwd(h, 4, param, -1)
                        assert same(out["all"][0].contiguous(), ref), (shape, dtype, kind, pre_relu)


def test_forward_elision_per_channel_scale_and_masked_rows():
    """per-channel scales together with a channel mask (PARAM_PER_CHANNEL kernels)"""
    for si, shape in enumerate(SHAPES[:5]):
        C = shape[1]
        x = (torch.randn(shape, generator=gen(40 + si)) * 3).bfloat16()
        s = torch.rand(C, 1, generator=gen(50 + si)) + 0.1
        mask = _mask(C, 60 + si)
        out = {}
        for mode in ("off", "all"):
            qs.set_qsparse_options(elide_pruned=mode)
            out[mode] = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), 1, torch.float32, chan_mask=mask.to(DEV),
                                       mask_channel_index=1)[0].cpu()
        assert same(out["off"], out["all"]), shape
        assert same(out["all"], O.scaler_fwd(x * mask.view(1, -1, 1, 1), 8, s, 1)), shape


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


def _nonfinite_case():
    x = torch.randn(4, 16, 8, 8, generator=gen(6)).bfloat16()
    x[0, 1, 0, 0], x[1, 1, 3, 3], x[2, 5, 1, 1] = float("nan"), float("inf"), float("-inf")
    mask = torch.ones(16, dtype=torch.bool)
    mask[1] = mask[5] = mask[9] = False
    return x, mask, torch.tensor([[0.25]])


def test_nan_and_inf_on_pruned_channels_follow_the_reference_unless_all_is_chosen():
    """quirk B15: x * 0 is NaN for a NaN / Inf x, and rounds to INT_MIN.  "off" and the default load such channels in every
    entry point (either layout, with and without gate recording); the opt-in "all" skips them (f32(0) * s)."""
    x, mask, s = _nonfinite_case()
    ref = O.scaler_fwd(x * mask.view(1, -1, 1, 1), 4, s, -1)
    ref_relu = O.scaler_fwd(x.relu() * mask.view(1, -1, 1, 1), 4, s, -1)
    assert ref[0, 1, 0, 0].item() == float(-2 ** 31) * 0.25 and ref_relu[2, 5, 1, 1].item() == 0.0     # relu(-inf) * 0 = 0
    xcl = x.to(DEV).contiguous(memory_format=torch.channels_last)
    for mode in ("off", "forward"):
        qs.set_qsparse_options(elide_pruned=mode)
        for xin in (x.to(DEV), xcl):
            y = _hip.quant_fwd("scaler", xin, s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1)[0]
            assert same(y.cpu().contiguous(), ref), mode
            y, _, gate = _hip.quant_fwd("scaler", xin, s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1,
                                        pre_relu=True, want_gate=True)
            assert same(y.cpu().contiguous(), ref_relu), mode
    qs.set_qsparse_options(elide_pruned="all")
    y_el = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV), mask_channel_index=1)[0].cpu()
    finite = torch.isfinite(x.float())
    assert same(y_el[finite], ref[finite])
    assert y_el[0, 1, 0, 0].item() == 0.0 and y_el[1, 1, 3, 3].item() == 0.0     # documented divergence of the opt-in
    # a NaN scale still propagates through the pruned channels: f32(0) * NaN
    y_nan = _hip.quant_fwd("scaler", x.to(DEV), torch.tensor([[float("nan")]], device=DEV), -1, torch.float32,
                           chan_mask=mask.to(DEV), mask_channel_index=1)[0].cpu()
    assert bool(y_nan[:, 1].isnan().all())


@pytest.mark.parametrize("kind", ["scaler", "decimal"])
def test_elision_mask_marks_the_pruned_channels_that_must_be_loaded(kind):
    """the select writes the elision mask from the per-channel abs-max it reduces anyway (1 kept, 0 pruned and finite, 2 pruned
    with a NaN / Inf); the forward handed that mask skips only the 0 channels and equals the loading path bit for bit."""
    x, mask, s = _nonfinite_case()
    C = 16
    for pre_relu in (False, True):
        h = x.relu() if pre_relu else x
        amax = h.float().abs().amax(dim=(0, 2, 3))                      # NaN propagates, as in the kernels' keys
        em = torch.full((C,), 7, dtype=torch.uint8, device=DEV)
        scale = torch.zeros(1, device=DEV)
        _hip.pq_select(torch.rand(C, generator=gen(8)).to(DEV), None, False, 0, False, 0, mask.to(DEV), amax.to(DEV), True, 0, 4, scale,
                       stat_dtype=torch.bfloat16, elide_mask=em)
        want = [1 if mask[c] else (0 if torch.isfinite(amax[c]) else 2) for c in range(C)]
        assert em.cpu().tolist() == want
        assert want[1] == 2 and want[9] == 0 and want[5] == (0 if pre_relu else 2)      # relu(-inf) = 0: finite
        param = s if kind == "scaler" else torch.tensor([[2.0]])
        out = {}
        for mode in ("off", "exact"):
            qs.set_qsparse_options(elide_pruned="off" if mode == "off" else "forward")
            out[mode] = _hip.quant_fwd(kind, x.to(DEV), param.to(DEV), -1, torch.float32, chan_mask=em if mode == "exact" else mask.to(DEV),
                                       mask_channel_index=1, pre_relu=pre_relu, elision_mask=mode == "exact")[0].cpu()
        assert same(out["off"], out["exact"]), pre_relu
        hm = h * mask.view(1, -1, 1, 1)
        ref = O.scaler_fwd(hm, 4, param, -1) if kind == "scaler" else O.decimal_fwd(hm, 4, param, -1)
        assert same(out["exact"], ref), pre_relu


def _site(act, kind="scaler", **cbkw):
    return nn.Sequential(nn.Sequential(nn.ReLU() if act else nn.Identity(),
                                       qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1,
                                                callback=qs.MagnitudePruningCallback(**cbkw))),
                         qs.quantize(bits=4, channelwise=-1, timeout=1,
                                     callback=qs.ScalerQuantizer() if kind == "scaler" else qs.DecimalQuantizer()))


@pytest.mark.parametrize("frozen", [True, False], ids=["frozen_mask", "refreshing_mask"])
@pytest.mark.parametrize("act", [False, True], ids=["identity", "relu"])
@pytest.mark.parametrize("dtype,kind", [(torch.bfloat16, "scaler"), (torch.float32, "scaler"), (torch.bfloat16, "decimal")])
def test_default_mode_is_exact_on_a_site_with_non_finite_values_on_pruned_channels(act, dtype, kind, frozen):
    """the default elision changed NaN / Inf on PRUNED channels of an NCHW site.  A whole convert-style
    (ReLU ->) prune -> quantize site meets NaN / Inf / -Inf on pruned channels: the fused pair (composite route) == the
    module-by-module GPU path == the CPU path (the reference's own torch arithmetic), in both layouts, with and without autograd:
      * evaluation (no statistics: every element is loaded): f32(INT_MIN) * s at the NaN, quirk B15;
      * training with a live scale: x * 0 is NaN there and the reference's x.abs().max() carries it into the scale -- the
        select, which takes the maximum over kept channels, accounts for it (`pq_scale_key`) -- and marks the channel in the
        elision mask, so the eliding apply kernel loads it.  (With a ScalerQuantizer every output is NaN then; a
        DecimalQuantizer turns the NaN scale into a step of 2^inf, q * inf, and the loaded NaN shows as -inf = INT_MIN * inf
        where a skipped one would give 0 * inf = NaN: the elision mask is what keeps that element right.)"""
    import copy
    assert qs.get_qsparse_option("elide_pruned") == "forward"
    cbkw = {"stop_mask_refresh": 3} if frozen else {}
    fused, plain, cpu = (fuse_prune_quantize_pairs(_site(act, kind, **cbkw).to(DEV)).train(), _site(act, kind, **cbkw).to(DEV).train(),
                         _site(act, kind, **cbkw).train())

    def batch(step):
        return (torch.randn(4, 16, 8, 8, generator=gen(300 + step)) * torch.linspace(0.25, 4, 16).view(1, -1, 1, 1)).to(dtype)

    for step in range(6):
        xs = batch(step)
        y = fused(xs.to(DEV)).cpu()
        assert same(y, plain(xs.to(DEV)).cpu()) and same(y, cpu(xs)), step
    assert fused[1].__dict__.get("_qs_last_route") == ("frozen" if frozen else "live")
    pruned = (~cpu[0][1].mask.view(-1)).nonzero().view(-1).tolist()
    assert len(pruned) >= 4 and same(fused[0][1].mask.cpu(), cpu[0][1].mask)
    step = 6
    for cl in (False, True):
        for grad in (False, True):
            for training in (True, False):
                sites = [copy.deepcopy(m).train(training) for m in (fused, plain, cpu)]     # (a NaN scale stays: fresh copies)
                for rep in range(2):
                    xs = batch(step)
                    step += 1
                    for j, c in enumerate(pruned[:3]):
                        xs[j, c, j, j] = (float("nan"), float("inf"), float("-inf"))[j]
                    xd = xs.to(DEV).contiguous(memory_format=torch.channels_last) if cl else xs.to(DEV)
                    tag = (cl, grad, training, rep)
                    inps = [v.clone().requires_grad_(grad) for v in (xd, xd, xs)]
                    with torch.set_grad_enabled(grad):
                        ys = [m(i) for m, i in zip(sites, inps)]
                    if grad and not training:       # (finite scale: the straight-through clamp has finite bounds)
                        gs = [torch.autograd.grad(y, i, torch.ones_like(y))[0].cpu().contiguous() for y, i in zip(ys, inps)]
                        assert same(gs[0], gs[1]) and same(gs[0], gs[2]), ("gradient", tag)
                    ys = [v.detach().cpu().contiguous() for v in ys]
                    assert same(ys[0], ys[1]) and same(ys[0], ys[2]), tag
                    for m in sites[:2]:
                        assert same(m[1].weight.cpu(), sites[2][1].weight) and same(m[0][1].mask.cpu(), sites[2][0][1].mask), tag
                    w = float(sites[2][1].weight)
                    if training:
                        assert w != w, tag                                        # the reference's scale is NaN from here on
                        if kind == "scaler":
                            assert bool(ys[0].isnan().all()), tag
                        else:
                            assert ys[0][0, pruned[0], 0, 0].item() == float("-inf"), tag
                    elif kind == "scaler":
                        assert w == w and ys[0][0, pruned[0], 0, 0].item() == float(-2 ** 31) * w, tag
                    if training and rep == 0 and not cl and not grad:      # this step elided: what the select wrote for the apply kernel
                        em = sites[0][1].__dict__["_qs_site_plan"].keep[7].cpu().tolist()
                        amax = (xs.relu() if act else xs).float().abs().amax(dim=(0, 2, 3))
                        mask = sites[2][0][1].mask.view(-1)
                        assert em == [1 if mask[c] else (0 if torch.isfinite(amax[c]) else 2) for c in range(16)], tag
                        assert em.count(2) >= 1 and em.count(0) >= 1, tag


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
