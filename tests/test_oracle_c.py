"""Cross-check of the two oracle restatements: plain C (oracle/qs_oracle.c) vs torch-CPU
(oracle/qs_oracle.py, itself pinned to the reference's golden vectors).  CPU only."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import qs_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = ctypes.POINTER(ctypes.c_float)
I64 = ctypes.c_int64


@pytest.fixture(scope="module")
def clib():
    import __graft_entry__ as ge
    path = ge.build_oracle()
    return ctypes.CDLL(path)


def fp(t):
    return t.contiguous().numpy().ctypes.data_as(F)


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c_quantizers_match(clib, dtype):
    shape = (5, 12, 7, 9)
    outer, C, inner = 5, 12, 63
    x = (torch.randn(shape, generator=gen(1)) * 3).to(dtype)
    xf = x.float().contiguous()
    for per_channel in (False, True):
        s = (torch.rand(C if per_channel else 1, 1, generator=gen(2)) * 0.2 + 0.01)
        y = torch.empty(shape)
        codes = torch.empty(shape, dtype=torch.int32)
        clib.qo_scaler_fwd(fp(xf), fp(s), I64(s.numel()), I64(outer), I64(C), I64(inner), fp(y),
                           codes.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        ci = 1 if per_channel else -1
        assert torch.equal(y, O.scaler_fwd(x, 8, s, ci)) and torch.equal(codes, O.scaler_codes(x, s, ci))
        d = torch.randint(0, 8, (C if per_channel else 1, 1), generator=gen(3)).float()
        clib.qo_decimal_fwd(fp(xf), fp(d), I64(d.numel()), I64(outer), I64(C), I64(inner), fp(y), None)
        assert torch.equal(y, O.decimal_fwd(x, 8, d, ci))
        lo = -torch.rand(C if per_channel else 1, 1, generator=gen(4)) * 2
        lines = torch.cat([lo, lo + torch.rand(lo.shape, generator=gen(5)) * 3], 1).contiguous()
        for fzp in (1, 0):
            clib.qo_line_fwd(fp(xf), fp(lines), I64(lines.shape[0]), 4, fzp, I64(outer), I64(C), I64(inner), fp(y))
            assert torch.equal(y, O.line_fwd(x, 4, lines, ci, bool(fzp)))
        g = torch.randn(shape, generator=gen(6)) * 2
        gx = torch.empty(shape)
        mask = (torch.rand(C, generator=gen(7)) > 0.5)
        clib.qo_ste_bwd(fp(g), fp(s), I64(s.numel()), ctypes.c_float(-8.0), ctypes.c_float(7.0),
                        mask.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), I64(outer), I64(C), I64(inner), fp(gx))
        want = O.ste_bwd(g, 4, s, ci) * mask.view(1, C, 1, 1)
        assert torch.equal(gx, want)
        # a NaN scale (a NaN / Inf input reached it): NaN bounds, torch.clamp returns NaN for every element of that channel; a
        # NaN gradient stays NaN through finite bounds; the reference's `v[v != grad_output] = 0` (v IS grad_output) then turns
        # exactly those NaNs into 0 (fixture F17)
        s_nan, g_nan = s.clone(), g.clone()
        s_nan.view(-1)[0] = float("nan")
        g_nan.view(-1)[5] = float("nan")
        clib.qo_ste_bwd(fp(g_nan), fp(s_nan), I64(s_nan.numel()), ctypes.c_float(-8.0), ctypes.c_float(7.0),
                        mask.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), I64(outer), I64(C), I64(inner), fp(gx))
        want = O.ste_bwd(g_nan, 4, s_nan, ci) * mask.view(1, C, 1, 1)
        assert torch.equal(gx, want) and not bool(gx.isnan().any())
        assert gx.view(-1)[5].item() == 0.0 and (per_channel or bool((gx == 0).all()))


def test_c_staged_mean_matches_aten_order(clib):
    for dtype in (torch.float32, torch.bfloat16):
        for shape, mshape in (((40, 6, 9, 8), (1, 6, 1, 1)), ((300, 70), (1, 70)), ((6, 6, 8, 40), (1, 6, 8, 1)),
                              ((64, 16, 14, 14), (1, 16, 1, 1)), ((7, 5, 3, 9), (1, 5, 1, 1))):
            x = torch.randn(shape, generator=gen(8)).abs().to(dtype)
            cur = x.float().contiguous()
            dims = list(cur.shape)
            for d, (sx, sm) in enumerate(zip(shape, mshape)):
                if sx == sm:
                    continue
                pre, n, post = int(np.prod(dims[:d], dtype=np.int64)), dims[d], int(np.prod(dims[d + 1:], dtype=np.int64))
                out = torch.empty(pre * post)
                clib.qo_mean_dim(fp(cur), I64(pre), I64(n), I64(post), fp(out))
                dims[d] = 1
                cur = out.to(dtype).float().view(dims).contiguous()   # one rounding to the tensor dtype per stage
            assert torch.equal(cur.to(dtype), O.squeeze_mean(x, mshape)), (dtype, shape)


def test_c_channels_last_stage_matches_aten_order(clib):
    """the channels_last first stage (any channel count) against ATen's CPU result on channels_last tensors, one and two
    intra-op threads (from 4 threads on ATen's own result changes for a few small-channel shapes)"""
    threads = torch.get_num_threads()
    try:
        for nthr in (1, min(2, threads)):
            torch.set_num_threads(nthr)
            for dtype in (torch.float32, torch.bfloat16):
                for shape in ((16, 3, 5, 5), (64, 12, 7, 7), (33, 20, 3, 3), (256, 6, 4, 6), (40, 10, 14, 14), (128, 36, 2, 2),
                              (17, 5, 1, 7), (64, 8, 7, 7), (64, 100, 6, 1), (64, 2, 28, 28), (33, 6, 28, 28), (48, 64, 8, 8)):
                    N, C, H, W = shape
                    x = torch.randn(shape, generator=gen(31)).abs().to(dtype).contiguous(memory_format=torch.channels_last)
                    ref = x.mean(0, keepdim=True)                        # the reference's first squeeze stage on this tensor
                    assert ref.is_contiguous()                           # ... is NCHW-contiguous
                    xm = x.float().permute(0, 2, 3, 1).contiguous()      # [N, H, W, C]: the tensor as it lies in memory
                    out = torch.empty(C * H * W)
                    clib.qo_mean_dim_cl(fp(xm), I64(N), I64(H * W), I64(C), fp(out))
                    assert torch.equal(out.to(dtype).view(1, C, H, W), ref), (nthr, dtype, shape)
    finally:
        torch.set_num_threads(threads)


def test_c_mask_and_running_mean(clib):
    imp = torch.rand(5000, generator=gen(9))
    imp = (imp * 40).floor() / 40
    for s in (0.0, 0.5, 0.75):
        k = O.kth_index(s, imp.numel())
        mask = torch.empty(imp.numel(), dtype=torch.uint8)
        clib.qo_mask_from_importance.restype = ctypes.c_float
        clib.qo_mask_from_importance(fp(imp), I64(imp.numel()), I64(k), mask.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        assert torch.equal(mask.bool(), O.mask_from_importance(imp, s))
    state, nv = torch.rand(64, generator=gen(10)), torch.rand(64, generator=gen(11))
    want = (3 * state + nv) / 4
    clib.qo_running_mean(fp(state), fp(nv), I64(64), I64(3))
    assert torch.equal(state, want)
