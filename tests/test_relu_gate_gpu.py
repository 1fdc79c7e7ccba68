"""The folded ReLU's gate as a bitmap (ABI v8: ``gate_out`` of qs_quant_*_fwd, ``gate`` of qs_quant_ste_relu_bwd; option
``relu_gate``, on by default).  ATen's threshold_backward (the backward of the nn.ReLU that convert() puts in front of a
prune -> quantize site, reference qsparse/convert.py:214-218) needs nothing of x but ``x <= 0``: the forward records that
as one bit per element and the backward reads g and the bitmap instead of g and x.

  * the recording forward returns the same bits as the plain one, in every layout / dtype / ragged geometry the kernels
    distinguish, and the bitmap is exactly packbits(!(x <= 0)) in memory order (NaN passes, -0.0 does not);
  * the backward from the bitmap is bit-identical to the backward from x (NaN / Inf / -0.0 in g and x included),
    also with backward elision ("all");
  * a converted site trains to the same bits with the option on and off, in NCHW and channels_last, and no longer keeps x.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from qsparse_amd import _hip
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda"
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


@pytest.fixture(autouse=True)
def _restore():
    yield
    qs.set_qsparse_options(elide_pruned="forward", preserve_dtype=False, relu_gate=True)


def gen(seed):
    return torch.Generator().manual_seed(seed)


SHAPES = [(4, 16, 8, 8),        # rows of 64: CM_ROW, whole waves
          (3, 24, 14, 14),      # rows of 196 = 4k: the widening kernel's 4-element lanes (nibbles), CM_ELEM for 8-per-lane kernels
          (5, 12, 7, 7),        # rows of 49: CM_ELEM, numel % 8 == 4 -> ragged tail byte
          (2, 8, 3, 1),         # rows shorter than a lane
          (9, 40, 56, 56),      # many full waves
          (6, 33),              # 2-d activation, C % 8 != 0, numel % 8 == 6
          (64, 48),             # 2-d, C % 8 == 0: CM_LAST
          (3, 5, 8, 8),         # whole waves + a partial last wave
          (2, 3, 24, 24),       # rows >= 512 elements (wave-uniform look-up) with a partial last wave
          (1, 1, 1, 5),         # fewer than 8 elements: the tail byte only
          (3, 4, 32, 40)]


def special(x, seed):
    """sprinkle the values the gate must get right: -0.0, +0.0, NaN, +-Inf, denormals"""
    flat = x.view(-1)
    idx = torch.randperm(flat.numel(), generator=gen(seed))[:max(1, flat.numel() // 7)]
    vals = torch.tensor([-0.0, 0.0, float("nan"), float("inf"), float("-inf"), 1e-40, -1e-40], dtype=torch.float32)
    flat[idx] = vals[torch.arange(idx.numel()) % vals.numel()].to(x.dtype)
    return x


def expected_bits(x_mem: torch.Tensor) -> np.ndarray:
    """bit (e & 7) of byte e >> 3 = !(x[e] <= 0) over the memory-order elements"""
    open_ = ~(x_mem.float().reshape(-1).cpu() <= 0)
    return np.packbits(open_.numpy().astype(np.uint8), bitorder="little")


def layouts(shape):
    yield "nchw"
    if len(shape) == 4 and shape[1] > 1 and shape[2] * shape[3] > 1:
        yield "channels_last"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("kind", ["scaler", "decimal"])
def test_recording_forward_and_bitmap(dtype, kind):
    for si, shape in enumerate(SHAPES):
        for layout in layouts(shape):
            for masked in (False, True):
                for out_dtype in (torch.float32, dtype):          # the widening kernel / the generic one (preserve_dtype)
                    C = shape[1]
                    x = special((torch.randn(shape, generator=gen(si)) * 3).to(dtype), si + 100).to(DEV)
                    if layout == "channels_last":
                        x = x.contiguous(memory_format=torch.channels_last)
                    mask = (torch.rand(C, generator=gen(si + 7)) < 0.5).to(DEV) if masked else None
                    param = torch.tensor([0.37], device=DEV) if kind == "scaler" else torch.tensor([2.0], device=DEV)
                    tag = (shape, layout, masked, str(out_dtype))
                    for mode in ("off", "forward", "all"):  # NaN / Inf on a pruned channel: INT_MIN * s when loaded, Q(+0) when elided --
                        qs.set_qsparse_options(elide_pruned=mode)       # the recording forward follows the mode it is called in
                        y0, _ = _hip.quant_fwd(kind, x, param, -1, torch.float32, chan_mask=mask, mask_channel_index=1,
                                               out_dtype=out_dtype, pre_relu=True)
                        y1, _, gate = _hip.quant_fwd(kind, x, param, -1, torch.float32, chan_mask=mask, mask_channel_index=1,
                                                     out_dtype=out_dtype, pre_relu=True, want_gate=True)
                        assert y1.dtype == y0.dtype and y1.stride() == y0.stride(), (tag, mode)
                        a, b = y0.cpu(), y1.cpu()
                        if mode == "all" and masked:
                            # an eliding kernel skips a pruned NaN / Inf or not depending on the lane geometry (documented:
                            # bit-identical for finite x); the recording kernel always counts it as +0.0
                            ok = (mask.view([1, -1] + [1] * (x.dim() - 2)) | torch.isfinite(x)).cpu()
                            a, b = torch.where(ok, a, torch.zeros_like(a)), torch.where(ok, b, torch.zeros_like(b))
                        assert same(a, b), (tag, mode)
                    x_mem = x.permute(0, 2, 3, 1) if gate.channels_last else x
                    assert gate.channels_last == (layout == "channels_last"), tag
                    got = gate.bits.cpu().numpy()
                    want = expected_bits(x_mem)
                    n = x.numel()
                    assert got.shape == want.shape == ((n + 7) // 8,), tag
                    if n % 8:                                       # bits past the last element are unspecified
                        keep = (1 << (n % 8)) - 1
                        got, want = got.copy(), want.copy()
                        got[-1] &= keep
                        want[-1] &= keep
                    assert np.array_equal(got, want), tag


@pytest.mark.parametrize("xdtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("elide", ["forward", "all"])
def test_backward_from_bitmap_equals_backward_from_x(xdtype, elide):
    qs.set_qsparse_options(elide_pruned=elide)
    for si, shape in enumerate(SHAPES):
        for layout in layouts(shape):
            for masked in (False, True):
                for gdtype in (torch.float32, xdtype):
                    C = shape[1]
                    x = special((torch.randn(shape, generator=gen(si)) * 3).to(xdtype), si + 100).to(DEV)
                    g = special((torch.randn(shape, generator=gen(si + 50)) * 2).to(gdtype), si + 200).to(DEV)
                    if layout == "channels_last":
                        x = x.contiguous(memory_format=torch.channels_last)
                        if si % 2:                                   # the gradient may arrive in either layout
                            g = g.contiguous(memory_format=torch.channels_last)
                    mask = (torch.rand(C, generator=gen(si + 7)) < 0.5).to(DEV) if masked else None
                    scale = torch.tensor([0.21], device=DEV)
                    _, _, gate = _hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1,
                                                pre_relu=True, want_gate=True)
                    a = _hip.ste_relu_bwd(g, x, scale, False, -8.0, 7.0, mask)
                    b = _hip.ste_relu_bwd(g, None, scale, False, -8.0, 7.0, mask, gate=gate)
                    tag = (shape, layout, masked, str(gdtype))
                    assert a.dtype == b.dtype == xdtype and a.shape == b.shape and a.stride() == b.stride(), tag
                    assert same(a.cpu().contiguous(), b.cpu().contiguous()), tag


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_mask_apply_records_the_same_bitmap(dtype):
    """the ReLU -> PruneLayer site: max(x, 0) * mask forward (qs_mask_apply), gate * g * mask backward"""
    for si, shape in enumerate(SHAPES):
        for layout in layouts(shape):
            for mode in ("forward", "all"):
                qs.set_qsparse_options(elide_pruned=mode)
                C = shape[1]
                x = special((torch.randn(shape, generator=gen(si)) * 3).to(dtype), si + 100).to(DEV)
                g = special((torch.randn(shape, generator=gen(si + 50)) * 2).to(dtype), si + 200).to(DEV)
                if layout == "channels_last":
                    x = x.contiguous(memory_format=torch.channels_last)
                mask = (torch.rand(C, generator=gen(si + 7)) < 0.5).to(DEV).view([1, C] + [1] * (len(shape) - 2))
                y0 = _hip.mask_apply(x, mask, pre_relu=True)
                y1, gate = _hip.mask_apply(x, mask, pre_relu=True, want_gate=True)
                tag = (shape, layout, mode)
                a, b = y0.cpu(), y1.cpu()
                assert y0.stride() == y1.stride() and a.dtype == b.dtype, tag
                if mode == "all":
                    # an eliding kernel skips a pruned lane or not depending on the geometry: a pruned NaN / Inf / -0.0 gives NaN /
                    # NaN / -0.0 when loaded and +0.0 when skipped; the recording kernel always counts a pruned x as +0.0.  Mode
                    # "all" promises numerical equality for finite inputs, bit equality on kept channels.
                    kept = mask.expand_as(x).cpu()
                    assert same(torch.where(kept, a, torch.zeros_like(a)), torch.where(kept, b, torch.zeros_like(b))), tag
                    fin = (~kept) & torch.isfinite(x).cpu()
                    assert bool((a[fin].float() == b[fin].float()).all()), tag
                else:
                    assert same(a, b), tag
                x_mem = x.permute(0, 2, 3, 1) if gate.channels_last else x
                got, want, n = gate.bits.cpu().numpy().copy(), expected_bits(x_mem).copy(), x.numel()
                if n % 8:
                    got[-1] &= (1 << (n % 8)) - 1
                    want[-1] &= (1 << (n % 8)) - 1
                assert np.array_equal(got, want), tag
                inf = float("inf")
                ga = _hip.ste_relu_bwd(g, x, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=1)
                gb = _hip.ste_relu_bwd(g, None, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=1, gate=gate)
                assert ga.stride() == gb.stride() and same(ga.cpu().contiguous(), gb.cpu().contiguous()), tag


def test_gate_with_a_mask_along_another_dim_of_a_channels_last_tensor():
    """PruneLayer(dimensions={2}) behind a ReLU: the mask varies along H; in channels_last memory that is dim 1 of the NHWC view"""
    x = special(torch.randn(3, 8, 6, 16, generator=gen(5)) * 2, 9).to(DEV).contiguous(memory_format=torch.channels_last)
    g = torch.randn(3, 8, 6, 16, generator=gen(6)).to(DEV)
    mask = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.bool, device=DEV).view(1, 1, 6, 1)
    y, gate = _hip.mask_apply(x, mask, pre_relu=True, want_gate=True)
    assert same(y.cpu(), torch.relu(x.cpu()) * mask.cpu())       # (ATen's CPU relu: the reference path; the GPU one treats -0.0 differently)
    inf = float("inf")
    ga = _hip.ste_relu_bwd(g, x, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=2)
    gb = _hip.ste_relu_bwd(g, None, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=2, gate=gate)
    assert same(ga.cpu().contiguous(), gb.cpu().contiguous())
    want = torch.where(x <= 0, torch.zeros_like(g), g * mask)
    assert same(gb.cpu().contiguous(), want.cpu().contiguous())


def _site(quantize_only=False, prune_only=False):
    if prune_only:
        return fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=2,
                                                                           repetition=2))).to(DEV).train()
    if quantize_only:
        return fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=2))).to(DEV).train()
    site = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=2, repetition=2)),
                         qs.quantize(bits=4, channelwise=-1, timeout=2)).to(DEV).train()
    return fuse_prune_quantize_pairs(site)


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("kind", ["pair", "quantize_only", "prune_only"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_site_trains_to_the_same_bits_with_and_without_the_bitmap(channels_last, kind, dtype):
    runs = {}
    for gate in (False, True):
        qs.set_qsparse_options(relu_gate=gate)
        torch.manual_seed(0)
        site = _site(kind == "quantize_only", kind == "prune_only")
        outs = []
        for step in range(7):
            x = (torch.randn((6, 16, 14, 14), generator=gen(step)) * 2).to(dtype).to(DEV)
            if channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            x.requires_grad_(True)
            y = site(x)
            g = torch.randn(y.shape, generator=gen(step + 30)).to(y.dtype).to(DEV)
            (gx,) = torch.autograd.grad(y, x, g)
            assert gx.dtype == x.dtype and gx.stride() == x.stride()
            outs.append((y.detach().cpu(), gx.cpu()))
        runs[gate] = (outs, {k: v.cpu() for k, v in site.state_dict().items()})
    for (y0, g0), (y1, g1) in zip(runs[False][0], runs[True][0]):
        assert same(y0, y1) and same(g0, g1)
    for k in runs[False][1]:
        assert same(runs[False][1][k], runs[True][1][k]), k


def test_the_relus_input_is_not_kept_for_the_backward():
    """with the bitmap the autograd node holds n/8 bytes instead of x: x may be overwritten before the backward runs"""
    site = _site()
    for _ in range(4):
        site(torch.randn(4, 16, 8, 8, device=DEV))
    leaf = torch.randn(4, 16, 8, 8, device=DEV, requires_grad=True)
    x = leaf * 1.0
    y = site(x)
    assert y.dtype == torch.float32 and site[1]._quantized
    want = torch.autograd.grad(y, leaf, torch.ones_like(y), retain_graph=True)[0]
    with torch.no_grad():
        x.zero_()                              # would raise "modified by an inplace operation" if x were a saved tensor
    got = torch.autograd.grad(y, leaf, torch.ones_like(y))[0]
    assert same(want.cpu(), got.cpu())


def test_gate_needs_pre_relu():
    x = torch.randn(64, device=DEV)
    with pytest.raises(ValueError):
        _hip.quant_fwd("scaler", x, torch.tensor([0.5], device=DEV), -1, torch.float32, want_gate=True)
    lib = _hip.load()
    y = torch.empty_like(x)
    gate = torch.empty(8, dtype=torch.uint8, device=DEV)
    st = lib.qs_quant_scaler_fwd(x.data_ptr(), y.data_ptr(), None, None, 1, 0.5, None, 1, 1, 64, 0, 0, 0, 0, 0, 0, 0, 0,
                                 gate.data_ptr(), None, 0, None, None)
    assert st == -2                            # QS_ERR_ARG: nothing enqueued
