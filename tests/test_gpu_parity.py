"""GPU parity: the HIP path (through the C ABI of libqsparse_hip.so) against
  (1) the golden vectors recorded from the real reference,
  (2) the CPU oracle on seeded inputs at sizes the oracle finishes in seconds,
  (3) size-independent properties at the BASELINE.json shapes (256x64x56x56 and 256x256x56x56).

Bars (north_star): integer codes, masks, indices and counters bit-exact; floating-point values within
1e-6 relative -- and in fact every float comparison below is asserted bit-exact too, because each
kernel performs the reference's fp32 operator chain with one rounding per operator.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd import _hip
from qsparse_amd.quantize import quantize_with_decimal, quantize_with_line, quantize_with_scaler
from qsparse_amd.sparse import apply_mask
from qsparse_amd.util import squeeze_tensor_to_shape

import test_host_golden as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
REL_TOL = 1e-6   # north_star: floating-point quantized values within 1e-6 relative


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def close(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return bool(((a - b).abs() <= REL_TOL * b.abs()).all())


def test_library_is_loaded_and_versioned():
    lib = _hip.load()
    assert lib.qs_version() == _hip.ABI_VERSION
    assert torch.cuda.is_available()


# ---- (1) golden fixtures on the GPU ------------------------------------------------------------
def test_golden_f1_f2_functional():
    H.run_f1_f2(DEV)


def test_golden_f3_line():
    H.run_f3(DEV)


def test_golden_f4_quantize_layer():
    H.run_f4(DEV)


def test_golden_f5_squeeze():
    H.run_f5(DEV)


def test_golden_f6_mask():
    H.run_f6(DEV)


def test_golden_f7_prune_layer():
    H.run_f7(DEV)


def test_golden_f15_prune_layer_use_gradient():
    H.run_f15(DEV)


@pytest.mark.parametrize("fused", [False, True])
def test_golden_f10_pair(fused):
    H.run_f10(DEV, fused)


@pytest.mark.parametrize("fused", [False, True])
def test_golden_f17_pair_with_non_finite_values_on_pruned_channels(fused):
    """the REFERENCE's recorded outputs for NaN / Inf / -Inf on pruned channels (f32(INT_MIN) * s in evaluation, a NaN scale
    and NaN clamp bounds once a live scale has seen them), module by module and through the fused pair (composite route,
    default elision)"""
    H.run_f17(DEV, fused)


def test_golden_f13_uniform_pruning_callback():
    H.run_f13(DEV)


def test_golden_f16_mnist_recipe_with_layerwise_schedule():
    """the reference's config-1 recipe (examples/mnist.py:193-199) trained on the GPU: module tree, schedule attributes,
    every counter, `_cur_sparsity`, the kept-entry count of every mask at every step and the reference's IndexError (same
    step, same index) -- see run_f16 for what can and cannot be compared across convolution arithmetics"""
    H.run_f16(DEV)


def test_golden_f9_reference_checkpoint_on_gpu():
    """row f3 of SURVEY 8: a checkpoint written by the REFERENCE (fixture F9: quantize(prune(Conv2d)) after 45 steps) is
    preloaded + loaded, moved to the GPU and evaluated there.  State tensors survive `.cuda()` bit for bit; the
    pruned weight the GPU operator hands the convolution is the reference's; quirk B7 (`_quantized` is not
    checkpointed, so a freshly loaded quantizer is a pass-through in eval) is reproduced; and training continues on
    the GPU exactly as it continues on the CPU from the same checkpoint."""
    g, _, schema = H._trees()
    ref_sd = {key: g.get("sd_" + key) for key in schema}
    xt = g.get("eval_x")

    def load(dev):
        m = H._make_conv()
        qs.preload_qsparse_state_dict(m, {k: v.clone() for k, v in ref_sd.items()})
        m.load_state_dict(ref_sd)
        return m.to(dev)

    gpu, cpu = load(DEV), load("cpu")
    for k, v in gpu.state_dict().items():
        assert v.is_cuda and same(v.cpu(), ref_sd[k]), k
    gpu.eval(), cpu.eval()
    w = gpu.weight.detach()
    pruned = gpu.prune(w)
    assert same(pruned.cpu(), ref_sd["weight"] * ref_sd["prune.mask"])
    assert gpu.quantize(pruned) is pruned                      # B7: pass-through until a training step sets _quantized
    y = gpu(xt.to(DEV))
    y_ref = g.get("eval_y_reloaded")
    assert y.shape == y_ref.shape and torch.allclose(y.cpu(), y_ref, rtol=1e-4, atol=1e-5)     # MIOpen vs CPU convolution
    assert same(torch.nn.functional.conv2d(xt, pruned.cpu(), ref_sd["bias"]), y_ref)            # same operands, CPU conv
    # resume training: the weight-side operators see the same parameters on both devices
    gpu.train(), cpu.train()
    for s in range(8):
        x = torch.rand(4, 16, 7, 7, generator=gen(7000 + s))
        gpu(x.to(DEV)), cpu(x)
        sg, sc = gpu.state_dict(), cpu.state_dict()
        for k in sc:
            assert same(sg[k].cpu(), sc[k]), (s, k)
    assert gpu.quantize._quantized and gpu.quantize._n_updates.item() == int(ref_sd["quantize._n_updates"][0]) + 8
    gpu.eval(), cpu.eval()
    assert same(gpu.quantize(gpu.prune(gpu.weight)).cpu(), cpu.quantize(cpu.prune(cpu.weight)))


def test_golden_f14_counters_written_through_data():
    """`.data` writes to GPU-resident counters are seen by the next forward without a per-step sync"""
    H.run_f14(DEV)
    H.run_data_write_fast_forward(DEV)


# ---- (2) oracle parity on seeded inputs -----------------------------------------------------------
SHAPES = [(8, 64, 28, 28), (3, 7, 5, 3), (2, 5, 7, 7), (64, 40), (1, 16, 56, 56), (4099,)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_scaler_fwd_bwd_vs_oracle(dtype, shape):
    x = (torch.randn(shape, generator=gen(1)) * 3).to(dtype)
    gout = torch.randn(shape, generator=gen(2)) * 2
    cases = [(-1, torch.tensor([[0.0371]]) if len(shape) > 1 else torch.tensor([0.0371]))]
    if len(shape) > 1:
        for ci in (0, 1, len(shape) - 1):
            cases.append((ci, torch.rand(shape[ci], 1, generator=gen(3 + ci)) * 0.2 + 0.003))
    for ci, s in cases:
        for bits, flip in ((8, False), (4, True)):
            y_ref = O.scaler_fwd(x, bits, s, ci)
            codes_ref = O.scaler_codes(x, s, ci)
            gx_ref = O.ste_bwd(gout, bits, s, ci, flip, False, dtype)
            xg = x.to(DEV).requires_grad_(True)
            y = quantize_with_scaler(xg, bits, s.to(DEV), ci, False, False, flip)
            y.backward(gout.to(DEV))
            _, codes = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), ci, torch.float32, want_codes=True)
            assert same(codes.cpu(), codes_ref), (shape, ci)          # integer class: bit-exact
            assert close(y, y_ref) and same(y.detach().cpu(), y_ref), (shape, ci)
            assert same(xg.grad.cpu(), gx_ref), (shape, ci, bits)


def test_scaler_codes_around_halfway_points():
    """the forward avoids the division when x*RN(1/s) is provably on the same side of every k+0.5 as
    RN(x/s); hammer the neighbourhood of the half-way points (+-3 ulp) where that proof does not hold and
    the kernel must fall back to the correctly rounded division: codes stay bit-exact."""
    ks = torch.arange(-300, 300, dtype=torch.float64) + 0.5
    for seed in range(6):
        s64 = torch.rand(1, generator=gen(50 + seed), dtype=torch.float64) * (10.0 ** (seed - 3)) + 1e-4
        s = s64.float()
        centre = (ks * s.double()).float()
        xs = [centre]
        up, dn = centre.clone(), centre.clone()
        for _ in range(3):
            up = torch.nextafter(up, torch.full_like(up, float("inf")))
            dn = torch.nextafter(dn, torch.full_like(dn, float("-inf")))
            xs += [up.clone(), dn.clone()]
        x = torch.cat(xs + [torch.randn(4096, generator=gen(60 + seed)) * 50 * s])
        x = torch.cat([x, x.new_zeros((-x.numel()) % 8)])
        _, codes = _hip.quant_fwd("scaler", x.to(DEV), s.view(1, 1).to(DEV), -1, torch.float32, want_codes=True)
        assert same(codes.cpu(), O.scaler_codes(x, s.view(1), -1)), seed
        # per-channel scales take the same route
        xc = x.view(1, 8, -1)
        sc = (torch.rand(8, 1, generator=gen(70 + seed)) + 0.5) * s
        _, codes = _hip.quant_fwd("scaler", xc.to(DEV), sc.to(DEV), 1, torch.float32, want_codes=True)
        assert same(codes.cpu(), O.scaler_codes(xc, sc, 1)), seed


def test_line_levels_around_halfway_points():
    """LineQuantization goes through the same division-free quotient: hammer +-3 ulp around every level boundary
    start + (k + 0.5) * step (training form) and (k + 0.5) * step (evaluation form), tensor-wise and per-channel."""
    for seed in range(5):
        bits = (4, 8, 6, 3, 8)[seed]
        n = 2 ** bits
        lo = -torch.rand(8, 1, generator=gen(80 + seed)) * (10.0 ** (seed - 2)) - 1e-3
        hi = torch.rand(8, 1, generator=gen(90 + seed)) * (10.0 ** (seed - 2)) + 1e-3
        lines = torch.cat([lo, hi], 1)
        step = (hi - lo) / n
        ks = torch.arange(-2, n + 2, dtype=torch.float32) + 0.5
        cols = []
        for base in (lo + ks.view(1, -1) * step, ks.view(1, -1) * step + torch.zeros_like(lo)):
            up, dn = base.clone(), base.clone()
            cols.append(base)
            for _ in range(3):
                up = torch.nextafter(up, torch.full_like(up, float("inf")))
                dn = torch.nextafter(dn, torch.full_like(dn, float("-inf")))
                cols += [up.clone(), dn.clone()]
        x = torch.cat(cols + [torch.randn(8, 512, generator=gen(95 + seed)) * (hi - lo)], 1)
        x = torch.cat([x, x.new_zeros(8, (-x.shape[1]) % 8)], 1).view(1, 8, -1).contiguous()
        for fzp in (True, False):
            y = quantize_with_line(x.to(DEV), bits, lines.to(DEV), 1, False, fzp)
            assert same(y.cpu(), O.line_fwd(x, bits, lines, 1, fzp)), (seed, fzp, "per-channel")
            y = quantize_with_line(x[:, :1].contiguous().to(DEV), bits, lines[:1].to(DEV), -1, False, fzp)
            assert same(y.cpu(), O.line_fwd(x[:, :1].contiguous(), bits, lines[:1], -1, fzp)), (seed, fzp, "tensor-wise")
    # non-finite inputs: NaN propagates (torch.clamp keeps it), +-inf clamp to the line's ends
    xs = torch.tensor([float("nan"), float("inf"), float("-inf"), 3e38, -3e38, 0.3, -0.2, 0.0])
    ln = torch.tensor([[-1.0, 1.5]])
    for fzp in (True, False):
        got, want = quantize_with_line(xs.to(DEV), 4, ln.to(DEV), -1, False, fzp).cpu(), O.line_fwd(xs, 4, ln, -1, fzp)
        assert torch.isnan(got[0]) and torch.isnan(want[0]) and same(got[1:], want[1:]), fzp
    # degenerate rows: step == 0 is replaced by 1e-4 (quantize.py:160)
    flat = torch.tensor([[0.5, 0.5], [-1.0, 1.0]])
    xz = torch.randn(1, 2, 64, generator=gen(99))
    for fzp in (True, False):
        assert same(quantize_with_line(xz.to(DEV), 4, flat.to(DEV), 1, False, fzp).cpu(), O.line_fwd(xz, 4, flat, 1, fzp))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decimal_and_line_vs_oracle(dtype):
    shape = (4, 24, 14, 14)
    x = (torch.randn(shape, generator=gen(5)) * 2).to(dtype)
    for ci, d in ((-1, torch.tensor([[5.0]])), (1, torch.randint(0, 9, (24, 1), generator=gen(6)).float())):
        y = quantize_with_decimal(x.to(DEV), 8, d.to(DEV), ci)
        assert same(y.cpu(), O.decimal_fwd(x, 8, d, ci))
        _, codes = _hip.quant_fwd("decimal", x.to(DEV), d.to(DEV), ci, torch.float32, want_codes=True)
        assert same(codes.cpu(), O.decimal_codes(x, d, ci))
    lo = -torch.rand(24, 1, generator=gen(7)) * 2
    hi = torch.rand(24, 1, generator=gen(8)) * 2 + 0.05
    lines = torch.cat([lo, hi], 1)
    for ci, ln in ((-1, lines[:1]), (1, lines)):
        for fzp in (True, False):
            y = quantize_with_line(x.to(DEV), 4, ln.to(DEV), ci, False, fzp)
            assert same(y.cpu(), O.line_fwd(x, 4, ln, ci, fzp)), (ci, fzp)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_statistics_vs_oracle(dtype):
    for shape, ci in (((16, 32, 28, 28), -1), ((16, 32, 28, 28), 1), ((1, 32, 28, 28), 1), ((48, 20, 3, 3), 0),
                      ((40, 33), 1), ((7,), 0), ((16, 32, 5, 5), 1)):
        x = (torch.randn(shape, generator=gen(11)) * 5).to(dtype)
        am = _hip.absmax(x.to(DEV), ci).cpu()
        ref = x.abs().float().amax() if ci < 0 else x.abs().float().transpose(0, ci).reshape(shape[ci], -1).amax(1)
        assert same(am.view(-1), ref.reshape(-1).float()), (shape, ci)
        mn, mx = _hip.minmax(x.to(DEV), ci)
        xr = x.float().reshape(1, -1) if ci < 0 else x.float().transpose(0, ci).reshape(shape[ci], -1)
        assert same(mn.cpu(), xr.amin(1)) and same(mx.cpu(), xr.amax(1)), (shape, ci)


def test_quantize_layer_running_scale_vs_oracle():
    for kind, cw, shape in (("scaler", -1, (8, 16, 14, 14)), ("decimal", -1, (8, 16, 14, 14)),
                            ("adaptive", 1, (8, 16, 14, 14)), ("adaptive", -1, (8, 16, 14, 14)),
                            ("scaler", 1, (1, 16, 14, 14))):
        layer = qs.quantize(bits=4, channelwise=cw, timeout=2, callback=H.MK[kind]()).to(DEV)
        sim = O.QuantizeSim(kind, 4, cw, 2)
        layer.train()
        for s in range(7):
            if s == 6:
                layer.eval()
            x = ((torch.rand(shape, generator=gen(100 + s)) - 0.4) * (3 + s)).bfloat16()
            y = layer(x.to(DEV))
            y_ref = sim.step(x, training=s < 6)
            assert same(y.cpu(), y_ref), (kind, cw, s)
            assert same(layer.weight.detach().cpu(), sim.weight), (kind, cw, s)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_staged_mean_vs_oracle(dtype):
    """staged means reproduce ATen's CPU summation order: exact equality, all shapes / paths."""
    for shape, mshape in (((32, 24, 28, 28), (1, 24, 1, 1)), ((256, 8, 8, 8), (1, 8, 1, 1)),
                          ((5, 6, 7, 9), (1, 6, 1, 1)), ((5, 6, 7, 9), (5, 6, 1, 1)), ((5, 6, 7, 9), (1, 1, 7, 9)),
                          ((12, 6, 7, 56), (1, 6, 1, 1)), ((300, 70), (1, 70)), ((20, 3, 40, 100), (1, 3, 40, 1)),
                          ((4, 10, 3, 3), (1, 10, 3, 3)), ((6, 16, 12, 12), (6, 16, 12, 12)), ((1030, 16), (1, 16))):
        x = (torch.randn(shape, generator=gen(21)).abs() * torch.linspace(0.25, 4, shape[1]).view(
            [1, -1] + [1] * (len(shape) - 2))).to(dtype)
        out = squeeze_tensor_to_shape(x.to(DEV), mshape)
        assert same(out.cpu(), O.squeeze_mean(x, mshape)), (shape, mshape)


def test_kth_value_and_mask_vs_oracle():
    for n, tie in ((256, False), (2048, True), (40000, False), (600000, True), (2359296, False)):
        imp = torch.rand(n, generator=gen(31))
        if tie:
            imp = (imp * 50).floor() / 50
        imp[::7] *= -1
        for s in (0.0, 0.31, 0.5, 0.75, 0.97):
            m = qs.calculate_mask_given_importance(imp.to(DEV), s)
            assert bool((m.cpu() == O.mask_from_importance(imp, s)).all()), (n, tie, s)
    nanimp = torch.rand(1000, generator=gen(32))
    nanimp[5] = float("nan")
    assert bool((qs.calculate_mask_given_importance(nanimp.to(DEV), 0.5).cpu() == O.mask_from_importance(nanimp, 0.5)).all())
    with pytest.raises(IndexError):
        qs.calculate_mask_given_importance(torch.rand(10, device=DEV), 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_apply_patterns_vs_oracle(dtype):
    shape = (6, 10, 7, 8)
    x = torch.randn(shape, generator=gen(41)).to(dtype)
    gout = torch.randn(shape, generator=gen(42)).to(dtype)
    for mshape in ((1, 10, 1, 1), (6, 10, 1, 1), (1, 10, 7, 1), (6, 1, 7, 1), (1, 1, 1, 8), (6, 10, 7, 8), (1, 1, 1, 1),
                   (6, 1, 1, 8)):
        mask = torch.rand(mshape, generator=gen(43)) > 0.4
        xg = x.to(DEV).requires_grad_(True)
        y = apply_mask(xg, mask.to(DEV))
        y.backward(gout.to(DEV))
        assert same(y.detach().cpu(), x * mask), mshape
        assert same(xg.grad.cpu(), gout * mask), mshape
    with pytest.raises(RuntimeError):
        apply_mask(x.to(DEV), torch.ones(6, 10, 7, 9, dtype=torch.bool, device=DEV))


def test_prune_and_pair_trajectories_vs_oracle():
    """whole-layer trajectories (schedule, running magnitude, refresh, fused pair) on a mid-size
    activation: unfused GPU == fused GPU == oracle, step by step, bit for bit."""
    shape, C = (16, 48, 14, 14), 48
    for dtype in (torch.bfloat16, torch.float32):
        mods, sims = [], (O.PruneSim(0.75, [1], 2, 2, 3, False), O.QuantizeSim("scaler", 4, -1, 3))
        for fused in (False, True):
            pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.75, dimensions={1}, start=2, interval=2,
                                                                        repetition=3)),
                                 qs.quantize(bits=4, channelwise=-1, timeout=3)).to(DEV)
            if fused:
                from qsparse_amd.fused import fuse_prune_quantize_pairs
                fuse_prune_quantize_pairs(pair)
            mods.append(pair.train())
        for s in range(12):
            training = s < 11
            x = (torch.randn(shape, generator=gen(200 + s)).relu() * torch.linspace(0.25, 4, C).view(1, -1, 1, 1)).to(dtype)
            gout = torch.randn(shape, generator=gen(300 + s))
            n_before = sims[0].n_updates
            h_ref = sims[0].step(x, training)
            y_ref = sims[1].step(h_ref, training)
            gx_ref = sims[0].grad(sims[1].grad(gout, dtype), (not training) or n_before >= 2)
            for pair in mods:
                if not training:
                    pair.eval()
                xg = x.to(DEV).requires_grad_(True)
                y = pair(xg)
                y.backward(gout.to(DEV).to(y.dtype))
                pl, ql = pair[0][1], pair[1]
                assert same(y.detach().cpu(), y_ref), (dtype, s)
                assert same(xg.grad.cpu(), gx_ref), (dtype, s)
                assert same(pl.mask.detach().cpu(), sims[0].mask), (dtype, s)
                assert same(ql.weight.detach().cpu(), sims[1].weight), (dtype, s)
                if sims[0].magnitude is not None:
                    assert same(pl.callback.magnitude.detach().cpu(), sims[0].magnitude), (dtype, s)
                assert pl._n_updates.item() == sims[0].n_updates and ql._n_updates.item() == sims[1].n_updates


def test_l0_and_unstructured_and_2d_vs_oracle():
    sim = O.PruneSim(0.5, [1], 1, 1, 2, False, l0=True)
    layer = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2,
                     callback=qs.MagnitudePruningCallback(l0=True)).to(DEV).train()
    for s in range(6):
        x = (torch.randn(4, 12, 6, 6, generator=gen(400 + s)) * torch.linspace(0.1, 2, 12).view(1, -1, 1, 1)).relu()
        assert same(layer(x.to(DEV)).cpu(), sim.step(x))
        assert same(layer.mask.cpu(), sim.mask)
    sim = O.PruneSim(0.6, [0, 1, 2, 3], 1, 1, 2, False)
    layer = qs.prune(sparsity=0.6, dimensions={0, 1, 2, 3}, start=1, interval=1, repetition=2).to(DEV).train()
    for s in range(6):
        x = torch.randn(3, 8, 6, 6, generator=gen(500 + s)).bfloat16()
        assert same(layer(x.to(DEV)).cpu(), sim.step(x))
    sim = O.PruneSim(0.5, [1], 1, 1, 2, False)
    layer = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2).to(DEV).train()
    for s in range(6):
        x = torch.randn(32, 40, generator=gen(600 + s)).bfloat16()
        assert same(layer(x.to(DEV)).cpu(), sim.step(x))


def test_abi_rejects_bad_arguments_loudly():
    # a dtype the kernels are not written for never reaches them: the binding refuses it ...
    x = torch.randn(64, device=DEV, dtype=torch.float64)
    with pytest.raises(_hip.QsparseHipError):
        _hip.dt(x)
    # ... and the operators evaluate such a tensor with the package's ATen expression on the device (tests/test_other_dtypes_gpu.py)
    y64 = quantize_with_scaler(x, 8, torch.tensor([[0.1]], device=DEV))
    assert y64.dtype == torch.float32 and same(y64.cpu(), quantize_with_scaler(x.cpu(), 8, torch.tensor([[0.1]])))
    lib = _hip.load()
    y = torch.empty(64, device=DEV)
    st = lib.qs_quant_scaler_fwd(y.data_ptr() + 4, y.data_ptr(), None, None, 1, 0.1, None, 1, 1, 8, 0, 0, 0, 0, 0, 0, 0, 0, None, None, 0, None, None)
    assert st == -3 and b"aligned" in lib.qs_status_string(st)
    assert lib.qs_quant_scaler_fwd(y.data_ptr(), y.data_ptr(), None, None, 1, 0.1, None, 1, 1, 8, 7, 0, 0, 0, 0, 0, 0, 0, None, None, 0, None, None) == -1
    # an image without a gate bitmap (or from a geometry the gate-recording kernels do not serve) is rejected, nothing enqueued
    img = torch.empty(64, device=DEV, dtype=torch.bfloat16)
    assert lib.qs_quant_scaler_fwd(y.data_ptr(), y.data_ptr(), None, None, 1, 0.1, None, 1, 1, 64, 0, 0, 0, 0, 0, 0, 1, 0, None, img.data_ptr(), 1, None, None) == -2
    # odd storage offsets are re-packed by the binding instead of failing
    base = torch.randn(1001, device=DEV)
    assert same(quantize_with_scaler(base[1:], 8, 0.1).cpu(), O.scaler_fwd(base[1:].cpu(), 8, 0.1))
    assert quantize_with_scaler(torch.empty(0, 4, device=DEV), 8, 0.1).shape == (0, 4)


# ---- (3) BASELINE.json shapes: size-independent properties -----------------------------------------
def _headline_input(n, c, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    x = torch.randn((n, c, 56, 56), generator=g, device=DEV).relu_()
    x *= torch.linspace(0.25, 4.0, c, device=DEV).view(1, c, 1, 1)
    return x.to(dtype)


@pytest.mark.parametrize("n,c,bits", [(256, 64, 8), (256, 256, 4)])
def test_full_size_properties(n, c, bits):
    x = _headline_input(n, c)
    s = (x.abs().amax().float() / 2 ** (bits - 1)).view(1, 1)
    xg = x.clone().requires_grad_(True)
    y = quantize_with_scaler(xg, bits, s)
    assert y.dtype == torch.float32 and y.shape == x.shape
    # (a) every output is an integer multiple of the scale, with |code| <= 2^(bits-1)
    codes = (y / s).round()
    assert torch.equal(codes * s, y) and codes.abs().max().item() <= 2 ** (bits - 1)
    # (b) idempotence: quantizing the quantized tensor changes nothing
    assert torch.equal(quantize_with_scaler(y, bits, s), y)
    # (c) error bound: |y - x| <= s/2 (+ 1 ulp of the product)
    assert ((y - x.float()).abs() <= s * 0.5 * (1 + 1e-6)).all()
    # (d) a slice of the big launch equals the oracle on that slice (same kernel, same arithmetic)
    sl = (slice(n - 2, n), slice(None), slice(None), slice(None))
    assert same(y[sl].detach().cpu(), O.scaler_fwd(x[sl].cpu(), bits, s.cpu(), -1))
    # (e) STE backward: values clamped into the interval, untouched inside; checksum against torch
    g = torch.randn(x.shape, device=DEV)
    y.backward(g)
    lo, hi = O.ste_bounds(bits, s)
    assert xg.grad.dtype == torch.bfloat16
    assert torch.equal(xg.grad, torch.clamp(g, lo.item(), hi.item()).to(torch.bfloat16))
    del y, g, codes
    # (f) channel mask apply: exact zeros on pruned channels, bit-identical elsewhere, backward too
    mask = (torch.arange(c, device=DEV) % 4 == 1).view(1, c, 1, 1)
    xm = x.clone().requires_grad_(True)
    ym = apply_mask(xm, mask)
    assert torch.equal(ym[:, ~mask.view(-1)], torch.zeros_like(ym[:, ~mask.view(-1)]))
    assert torch.equal(ym[:, mask.view(-1)], x[:, mask.view(-1)])
    # (g) per-channel abs-max / tensor abs-max agree with torch reductions (order independent)
    assert torch.equal(_hip.absmax(x, 1), x.abs().amax(dim=(0, 2, 3)).float())
    assert torch.equal(_hip.absmax(x, -1), x.abs().amax().float().view(1))


def test_full_size_fused_pair_step_matches_unfused():
    """headline shape: one live training step of the fused pair == the two layers run separately."""
    x = _headline_input(256, 256)
    g = torch.randn(x.shape, device=DEV)
    outs = []
    for fused in (False, True):
        pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1,
                                                                    repetition=1)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
        if fused:
            from qsparse_amd.fused import fuse_prune_quantize_pairs
            fuse_prune_quantize_pairs(pair)
        for step in range(3):
            xg = x.clone().requires_grad_(True)
            y = pair(xg)
            y.backward(g)
        outs.append((y.detach(), xg.grad, pair[0][1].mask.clone(), pair[1].weight.clone(),
                     pair[0][1].callback.magnitude.clone()))
        del y, xg
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    mask = outs[0][2].view(-1)
    assert mask.sum().item() == 64                      # 75 % of 256 channels pruned
    # magnitude against the oracle's staged mean on the same data (CPU, one channel block at a time is too slow:
    # compare the first 4 channels only)
    m = O.squeeze_mean(x[:, :4].cpu().abs(), (1, 4, 1, 1)).view(-1)
    ref = torch.zeros(4)
    for t in range(3):
        ref = (t * ref + m) / (t + 1)          # sparse.py:89 with the same input three times
    assert torch.equal(outs[0][4].view(-1)[:4].cpu(), ref)


# ---- more shapes / dtypes through the fused pair, and graph capture of the C ABI -----------------------
@pytest.mark.parametrize("shape,dtype", [((6, 24, 7, 7), torch.bfloat16), ((32, 40), torch.float32),
                                         ((5, 16, 4, 8), torch.float16), ((3, 8, 2, 4, 4), torch.bfloat16)])
def test_fused_pair_on_ragged_and_nd_shapes(shape, dtype):
    """7x7 maps (inner not a multiple of 8 -> unfused route), 2-d activations, fp16, 5-d: convert-built pair
    (fused where eligible) against the oracle, state included."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    C = shape[1]
    pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1,
                                                                repetition=2)),
                         qs.quantize(bits=8, channelwise=-1, timeout=1)).to(DEV).train()
    fuse_prune_quantize_pairs(pair)
    ps, qsim = O.PruneSim(0.5, [1], 1, 1, 2, False), O.QuantizeSim("scaler", 8, -1, 1)
    for s in range(6):
        x = (torch.randn(shape, generator=gen(900 + s)) * torch.linspace(0.3, 3, C).view([1, C] + [1] * (len(shape) - 2))).to(dtype)
        gout = torch.randn(shape, generator=gen(950 + s))
        n_before = ps.n_updates
        y_ref = qsim.step(ps.step(x, True), True)
        gx_ref = ps.grad(qsim.grad(gout, dtype), n_before >= 1)
        xg = x.to(DEV).requires_grad_(True)
        y = pair(xg)
        y.backward(gout.to(DEV))
        assert same(y.detach().cpu(), y_ref), (shape, s)
        assert same(xg.grad.cpu(), gx_ref), (shape, s)
        assert same(pair[0][1].mask.cpu(), ps.mask) and same(pair[1].weight.detach().cpu(), qsim.weight), (shape, s)


def test_channels_last_and_sliced_inputs():
    x = torch.randn(4, 16, 8, 8, generator=gen(77))
    s = torch.tensor([[0.05]])
    xcl = x.to(DEV).contiguous(memory_format=torch.channels_last)
    assert same(quantize_with_scaler(xcl, 8, s.to(DEV)).cpu().contiguous(), O.scaler_fwd(x, 8, s))
    view = x.to(DEV)[:, 3:11, ::2]
    assert same(quantize_with_scaler(view, 8, s.to(DEV)).cpu(), O.scaler_fwd(x[:, 3:11, ::2], 8, s))
    mask = torch.rand(1, 8, 1, 1, generator=gen(78)) > 0.5
    assert same(apply_mask(view, mask.to(DEV)).cpu(), x[:, 3:11, ::2] * mask)


def test_abi_calls_are_graph_capturable():
    """the header promises: no allocation, no synchronisation, everything on the caller's stream -- so a
    statistics -> select -> apply -> backward sequence can be captured into a hipGraph and replayed."""
    lib = _hip.load()
    N, C, H, W = 8, 32, 8, 8
    x = (torch.randn(N, C, H, W, generator=gen(5)) * torch.linspace(0.3, 3, C).view(1, C, 1, 1)).bfloat16().to(DEV)
    g = torch.randn(N, C, H, W, generator=gen(6)).to(DEV)
    y, gx = torch.empty(N, C, H, W, device=DEV), torch.empty(N, C, H, W, device=DEV, dtype=torch.bfloat16)
    stage1 = torch.empty(C * H * W, device=DEV, dtype=torch.bfloat16)
    imp = torch.empty(C, device=DEV, dtype=torch.bfloat16)
    amax, mag = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    mask, scale = torch.ones(C, device=DEV, dtype=torch.uint8), torch.zeros(1, device=DEV)

    def sequence(stream):
        assert lib.qs_mean_dim(x.data_ptr(), stage1.data_ptr(), 1, N, C * H * W, 1, 1, 1, None, amax.data_ptr(), 1, H * W, C, stream) == 0
        assert lib.qs_mean_last2(stage1.data_ptr(), imp.data_ptr(), C, H, W, 1, 1, None, None, 1, None, stream) == 0
        assert lib.qs_pq_select(mag.data_ptr(), imp.data_ptr(), 1, C, 1, 0, 1, 16, mask.data_ptr(), amax.data_ptr(), 1, 1, 0, 4,
                                scale.data_ptr(), None, None, None, None, None, None, 1, None, 1, None, stream) == 0
        assert lib.qs_quant_scaler_fwd(x.data_ptr(), y.data_ptr(), None, scale.data_ptr(), 1, 0.0, mask.data_ptr(), N, C, H * W,
                                       1, 0, 0, 0, 0, 0, 0, 1, None, None, 0, None, stream) == 0
        assert lib.qs_quant_ste_bwd(g.data_ptr(), gx.data_ptr(), scale.data_ptr(), 1, 0.0, 0, -8.0, 7.0, 0, mask.data_ptr(),
                                    N, C, H * W, 0, 1, 0, stream) == 0

    sequence(None)
    torch.cuda.synchronize()
    want = (y.clone(), gx.clone(), mask.clone(), scale.clone())
    y.zero_(), gx.zero_(), mask.fill_(1), scale.zero_(), mag.zero_(), amax.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sequence(torch.cuda.current_stream().cuda_stream)
    y.zero_(), gx.zero_(), mask.fill_(1), scale.zero_(), mag.zero_(), amax.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(want, (y, gx, mask, scale)):
        assert torch.equal(a, b)
    ps = O.PruneSim(0.5, [1], 0, 1, 1, False)
    ps.cur_sparsity, ps.t, ps.mask = 0.5, 0, torch.ones(1, C, 1, 1, dtype=torch.bool)
    ps.magnitude = O.magnitude_update(torch.zeros(1, C, 1, 1), x.cpu(), 0)
    assert torch.equal(mask.bool().cpu(), O.mask_from_importance(ps.magnitude, 0.5).view(-1))


# ---- remaining callback options on the GPU ---------------------------------------------------------------
def test_weight_injection_gpu_equals_cpu_bitwise():
    """prune + quantize injected into a conv's weight/bias: the operators only see the parameters, so the GPU
    run must reproduce the CPU run (== reference, test_host_golden) bit for bit, whatever the conv does."""
    import copy
    torch.manual_seed(7)
    base = nn.Conv2d(16, 24, 3)
    states = []
    for dev in ("cpu", DEV):
        conv = copy.deepcopy(base).to(dev)
        conv = qs.quantize(qs.prune(conv, sparsity=0.6, dimensions={0, 1, 2, 3}, start=1, interval=1, repetition=2,
                                    callback=qs.MagnitudePruningCallback(running_average=False)),
                           bits=4, bias_bits=8, timeout=2, channelwise=0)
        conv.train()
        snaps = []
        for s in range(6):
            conv(torch.rand(2, 16, 8, 8, generator=gen(40 + s)).to(dev))
            snaps.append([t.detach().cpu().clone() for t in (conv.weight, conv.bias, conv.prune.mask, conv.quantize.weight,
                                                              conv.quantize_bias.weight)])
        states.append(snaps)
    for a, b in zip(*states):
        for u, v in zip(a, b):
            assert same(u, v)


def test_uniform_gradient_and_groupwise_options_on_gpu():
    # uniform (random) pruning: host RNG picks positions, the mask lives on the GPU
    np.random.seed(0)
    layer = qs.prune(sparsity=0.5, start=1, interval=1, repetition=2, dimensions={0, 1, 2, 3},
                     callback=qs.UniformPruningCallback()).to(DEV).train()
    x = torch.rand(2, 6, 8, 8, generator=gen(1)).to(DEV) + 0.1
    for _ in range(6):
        out = layer(x)
    assert abs((out == 0).float().mean().item() - 0.5) < 2 / out.numel()
    assert torch.equal(out == 0, ~layer.mask)
    # gradient-magnitude pruning: the tensor hook feeds qs_mean_dim / qs_running_mean during backward
    cb = qs.MagnitudePruningCallback(use_gradient=True).to(DEV)
    sim_mag, t = torch.zeros(1, 3, 6, 6), 0
    mask = torch.ones(1, 3, 6, 6, dtype=torch.bool, device=DEV)
    for s in range(5):
        inp = torch.randn(1, 3, 6, 6, generator=gen(10 + s)).to(DEV).requires_grad_(True)
        gout = torch.rand(1, 3, 6, 6, generator=gen(20 + s))
        cb(inp, 0.5, mask).backward(gout.to(DEV))
        # the hook sees the gradient flowing into `inp`'s consumer, i.e. gout * mask(before refresh of this step)
    assert cb.t.item() == 5 and hasattr(cb, "magnitude") and cb.magnitude.is_cuda
    assert abs((~mask).float().mean().item() - 0.5) <= 1 / mask.numel()
    # group-wise quantization: clustering on the host (sklearn), shared scales applied on the GPU
    data = (torch.rand(8, 10, 6, 6, generator=gen(30)) - 0.5) * 4
    ql = qs.quantize(bits=8, timeout=2, channelwise=1, callback=qs.AdaptiveQuantizer(group_num=4, group_timeout=4)).to(DEV)
    qc = qs.quantize(bits=8, timeout=2, channelwise=1, callback=qs.AdaptiveQuantizer(group_num=4, group_timeout=4))
    for _ in range(10):
        yg, yc = ql(data.to(DEV)), qc(data)
    assert same(ql.weight.detach().cpu(), qc.weight.detach())
    assert torch.equal(ql.callback.groups.cpu(), qc.callback.groups)
    assert same(yg.cpu(), yc)


def test_preserve_dtype_extension():
    """opt-in: outputs in the input dtype == the reference's float32 result rounded once; gradients likewise"""
    qs.set_qsparse_options(preserve_dtype=True)
    try:
        for dtype in (torch.bfloat16, torch.float16):
            x = (torch.randn(4, 16, 8, 8, generator=gen(3)) * 2).to(dtype)
            s = torch.tensor([[0.07]])
            g = torch.randn(4, 16, 8, 8, generator=gen(4)).to(dtype)
            for dev in ("cpu", DEV):
                xg = x.detach().clone().to(dev).requires_grad_(True)
                y = quantize_with_scaler(xg, 4, s.to(dev))
                y.backward(g.to(dev))
                assert y.dtype == dtype and xg.grad.dtype == dtype
                assert same(y.detach().cpu(), O.scaler_fwd(x, 4, s).to(dtype))
                lo, hi = O.ste_bounds(4, s)
                assert same(xg.grad.cpu(), torch.clamp(g.float(), lo.item(), hi.item()).to(dtype))
            pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1,
                                                                        repetition=1)),
                                 qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
            from qsparse_amd.fused import fuse_prune_quantize_pairs
            fuse_prune_quantize_pairs(pair)
            ps, qsim = O.PruneSim(0.5, [1], 0, 1, 1, False), O.QuantizeSim("scaler", 4, -1, 1)
            for step in range(4):
                y = pair(x.to(DEV))
                assert y.dtype == dtype and same(y.cpu(), qsim.step(ps.step(x, True), True).to(dtype)), step
    finally:
        qs.set_qsparse_options(preserve_dtype=False)


def test_row_split_statistics_kernel_all_widths(monkeypatch):
    """the row-split stage (R = 2/4/8 waves sharing the rows, chunk sums combined in order) must equal ATen's
    order exactly for every split width, including a ragged channel abs-max riding along."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import os, sys, torch
        sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
        from oracle import qs_oracle as O
        from qsparse_amd import _hip
        from qsparse_amd.util import _staged_mean_hip
        g = torch.Generator().manual_seed(1)
        torch.set_num_threads(1)     # ATen's channels_last order depends on the thread split
        for shape in ((64, 32, 7, 7), (256, 16, 8, 8), (100, 24, 6, 6), (512, 8, 4, 8), (37, 64, 7, 7), (70, 16, 5, 5),
                      (300, 16, 3, 3), (63, 8, 5, 7), (130, 8, 1, 3), (257, 24, 2, 3), (1100, 8, 3, 3), (16, 40, 1, 1), (9, 16, 3, 3),
                      (520, 16, 2, 2)):
            for dt in (torch.bfloat16, torch.float32, torch.float16):
                for cl in (False, True):      # NCHW: qs_mean_dim's kernels; channels_last: qs_mean_dim_cl's
                    x = (torch.randn(shape, generator=g) * torch.linspace(0.3, 3, shape[1]).view(1, -1, 1, 1)).to(dt)
                    if cl:
                        x = x.contiguous(memory_format=torch.channels_last)
                    ref = O.squeeze_mean(x.abs(), (1, shape[1], 1, 1))
                    for am in (torch.zeros(shape[1], device='cuda'), _hip.amax_accumulator(shape[1], 'cuda')):   # dense / one line per channel
                        out = _staged_mean_hip(x.cuda(), [0, 2, 3], take_abs=True, absmax_out=am, absmax_channel_dim=1)
                        assert torch.equal(out.cpu(), ref), (shape, dt, cl)
                        assert torch.equal(_hip.amax_values(am).cpu(), x.abs().float().amax(dim=(0, 2, 3))), (shape, dt, cl)
                        if am.dim() == 2:
                            assert not am[:, 1:].any()
        print('ok')
    """)
    # (split "1" + depth: the unsplit kernel with 16 / 32 rows in flight per wave)
    # (channels_last: the workgroup kernel also with 16 waves and with narrow waves, QS_CL_LANES)
    for split, depth, lanes in (("0", "0", "0"), ("2", "0", "0"), ("4", "0", "16"), ("8", "0", "32"), ("16", "0", "0"), ("16", "0", "16"),
                                ("1", "16", "64"), ("1", "32", "0")):
        env = dict(os.environ, QS_MEAN_SPLIT=split, QS_MEAN_DEPTH=depth, QS_CL_LANES=lanes)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "ok" in r.stdout, (split, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reduction_kernels_all_layouts(dtype):
    """abs-max / min-max: tensor-wise, rows (inner >= 64, sliced over the outer range), columns (inner < 64,
    vector and scalar variants), ragged sizes, single rows -- all order independent, hence exact."""
    for shape, ci in (((64, 48, 9, 9), 1), ((200, 8, 16, 16), 1), ((3, 5, 64), 1), ((1000, 16), 1), ((33, 7), 1),
                      ((16, 2048, 7, 7), 1), ((5, 3, 2, 2), 1), ((4, 6, 10, 10), 2), ((512, 512, 3, 3), 0), ((1, 8, 1, 1), 1),
                      ((9, 4, 130), 0), ((70000,), -1), ((3, 1, 5), -1), ((37, 12, 14, 14), 1), ((50, 1024, 14, 14), 1), ((21, 5, 3, 5), 1),
                      # few columns, many rows: the two-stage path (2-d inputs, channels_last activations)
                      ((5000, 2048), 1), ((4097, 24), 1), ((300, 8), 1), ((100000, 64), 1), ((1024, 2056), 1), ((9000, 512), 1), ((8192, 256), 1)):
        x = (torch.randn(shape, generator=gen(sum(shape))) * 3).to(dtype)
        xr = x.float().reshape(1, -1) if ci < 0 else x.float().transpose(0, ci).reshape(shape[ci], -1)
        assert same(_hip.absmax(x.to(DEV), ci).cpu(), xr.abs().amax(1)), (shape, ci)
        mn, mx = _hip.minmax(x.to(DEV), ci)
        assert same(mn.cpu(), xr.amin(1)) and same(mx.cpu(), xr.amax(1)), (shape, ci)
    for shape in ((64, 64, 14, 14), (16, 40, 9, 9), (33, 256, 7, 7)):          # channels_last, per channel, accumulating
        x = (torch.randn(shape, generator=gen(sum(shape))) * 3).to(dtype)
        xcl = x.contiguous(memory_format=torch.channels_last).to(DEV)
        want = x.float().abs().amax(dim=(0, 2, 3))
        assert same(_hip.absmax(xcl, 1).cpu(), want), shape
        buf = torch.full((shape[1],), 0.5, device=DEV)
        assert same(_hip.absmax(xcl, 1, accumulate_into=buf).cpu(), torch.maximum(want, torch.tensor(0.5))), shape
        assert same(_hip.absmax(xcl, 1, pre_relu=True).cpu(), torch.relu(x).float().amax(dim=(0, 2, 3))), shape
        mn, mx = _hip.minmax(xcl, 1)
        assert same(mn.cpu(), x.float().amin(dim=(0, 2, 3))) and same(mx.cpu(), x.float().amax(dim=(0, 2, 3))), shape


@pytest.mark.parametrize("C", [200, 256, 257, 1000, 1024, 1536, 2048, 2049, 4096])
def test_fused_select_all_channel_counts(C):
    """the C-sized select step has three code paths (256-thread, rank select up to 2048 channels, radix select
    beyond): the fused pair must match the oracle for channel counts on both sides of every boundary."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    shape = (4, C, 2, 4)
    pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.7, dimensions={1}, start=0, interval=1,
                                                                repetition=1)),
                         qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
    fuse_prune_quantize_pairs(pair)
    ps, qsim = O.PruneSim(0.7, [1], 0, 1, 1, False), O.QuantizeSim("scaler", 4, -1, 1)
    for s in range(4):
        x = (torch.randn(shape, generator=gen(C + s)) * (torch.rand(C, generator=gen(7 * C + s)) + 0.1).view(1, C, 1, 1)).bfloat16()
        y = pair(x.to(DEV))
        assert same(y.cpu(), qsim.step(ps.step(x, True), True)), (C, s)
        assert same(pair[0][1].mask.cpu(), ps.mask) and same(pair[1].weight.detach().cpu(), qsim.weight), (C, s)
        assert same(pair[0][1].callback.magnitude.cpu(), ps.magnitude), (C, s)


def test_relu_fold_is_bit_identical_to_materialised_relu():
    """convert() finds ReLU -> prune -> quantize; with the fold the ReLU is applied inside the kernels.  Outputs,
    input gradients (incl. the threshold_backward gate and signed zeros) and all state must equal the unfolded
    run and the oracle, for the pruning-only phase, the active phase and evaluation."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    for dtype, shape in ((torch.bfloat16, (8, 32, 8, 8)), (torch.float32, (4, 16, 7, 7)), (torch.float32, (3, 8, 6, 6)), (torch.bfloat16, (5, 16, 14, 14)), (torch.float16, (6, 24, 4, 8))):
        runs = []
        for fold in (True, False):
            qs.set_qsparse_options(fold_relu=fold)
            pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1,
                                                                    repetition=2)),
                                 qs.quantize(bits=4, channelwise=-1, timeout=2)).to(DEV).train()
            fuse_prune_quantize_pairs(pair)
            ps, qsim = O.PruneSim(0.5, [1], 1, 1, 2, False), O.QuantizeSim("scaler", 4, -1, 2)
            outs = []
            for s in range(7):
                training = s < 6
                if not training:
                    pair.eval()
                x = (torch.randn(shape, generator=gen(1000 + s)) * torch.linspace(0.3, 3, shape[1]).view(1, -1, 1, 1)).to(dtype)
                x.view(-1)[:4] = torch.tensor([0.0, -1e-3, 1.0, -1.0]).to(dtype)   # (torch's own CPU and GPU ReLU disagree on relu(-0.0))
                gout = torch.randn(shape, generator=gen(1100 + s))
                xg = x.to(DEV).requires_grad_(True)
                y = pair(xg)
                y.backward(gout.to(DEV).to(y.dtype))
                n_before = ps.n_updates
                h = torch.relu(x)
                y_ref = qsim.step(ps.step(h, training), training)
                gh = ps.grad(qsim.grad(gout.to(y_ref.dtype), dtype), (not training) or n_before >= 1)
                gx_ref = torch.where(x <= 0, torch.zeros_like(gh), gh)
                assert same(y.detach().cpu(), y_ref), (dtype, fold, s)
                assert same(xg.grad.cpu(), gx_ref), (dtype, fold, s)
                assert same(pair[0][1].mask.cpu(), ps.mask) and same(pair[1].weight.detach().cpu(), qsim.weight)
                outs.append((y.detach().clone(), xg.grad.clone()))
            runs.append(outs)
        for (ya, ga), (yb, gb) in zip(*runs):
            assert torch.equal(ya, yb) and same(ga.cpu(), gb.cpu())
    qs.set_qsparse_options(fold_relu=True)


@pytest.mark.parametrize("kind", ["scaler", "decimal"])
def test_relu_quantize_site_fold_is_bit_identical(kind):
    """convert() builds Sequential(ReLU, QuantizeLayer) for a quantize-only activation site; with the fold relu(x)
    is never materialised (abs-max of max(x,0), y = Q(max(x,0)), gated STE backward).  Outputs, gradients and state
    equal the module-by-module run and the oracle: identity phase, active phase, evaluation."""
    from qsparse_amd.fused import FusedActQuantize, fuse_prune_quantize_pairs
    cbs = {"scaler": qs.ScalerQuantizer, "decimal": qs.DecimalQuantizer}
    for dtype, shape in ((torch.bfloat16, (8, 32, 8, 8)), (torch.float32, (4, 16, 7, 7)), (torch.float32, (3, 8, 6, 6)), (torch.bfloat16, (5, 16, 14, 14)), (torch.float16, (3, 50))):
        runs = []
        for fold in (True, False):
            qs.set_qsparse_options(fold_relu=fold)
            site = nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=2, callback=cbs[kind]())).to(DEV).train()
            fuse_prune_quantize_pairs(site)
            assert type(site) is FusedActQuantize and str(site).startswith("Sequential(")
            qsim = O.QuantizeSim(kind, 4, -1, 2)
            outs = []
            for s in range(7):
                training = s < 6
                if not training:
                    site.eval()
                x = (torch.randn(shape, generator=gen(2000 + s)) * 2).to(dtype)
                x.view(-1)[:4] = torch.tensor([0.0, -1e-3, 1.0, -1.0]).to(dtype)
                gout = torch.randn(shape, generator=gen(2100 + s))
                xg = x.to(DEV).requires_grad_(True)
                y = site(xg)
                y.backward(gout.to(DEV).to(y.dtype))
                y_ref = qsim.step(torch.relu(x), training)
                gh = qsim.grad(gout.to(y_ref.dtype), dtype)
                gx_ref = torch.where(x <= 0, torch.zeros_like(gh), gh)
                assert same(y.detach().cpu(), y_ref), (dtype, fold, s)
                assert same(xg.grad.cpu(), gx_ref), (dtype, fold, s)
                assert same(site[1].weight.detach().cpu(), qsim.weight) and int(site[1]._n_updates) == qsim.n_updates
                outs.append((y.detach().clone(), xg.grad.clone()))
            runs.append(outs)
        for (ya, ga), (yb, gb) in zip(*runs):
            assert torch.equal(ya, yb) and same(ga.cpu(), gb.cpu())
    qs.set_qsparse_options(fold_relu=True)


def test_absmax_of_folded_relu():
    """qs_absmax(pre_relu=1) == max|relu(x)| for every reduction layout (tensor-wise, rows, columns)."""
    for dtype in (torch.bfloat16, torch.float32):
        for shape, ci in (((64, 48, 9, 9), -1), ((64, 48, 9, 9), 1), ((200, 8, 16, 16), 1), ((1000, 16), 1), ((33, 7), 1),
                          ((5, 4096), -1)):
            x = torch.randn(shape, generator=gen(77)).to(dtype)
            got = _hip.absmax(x.to(DEV), ci, pre_relu=True).cpu()
            r = torch.relu(x).float()
            ref = r.amax().view(1) if ci < 0 else r.transpose(0, ci).reshape(shape[ci], -1).amax(1)
            assert torch.equal(got, ref), (dtype, shape, ci)
        neg = -torch.rand(4, 8, 8, 8).to(dtype) - 0.1
        assert float(_hip.absmax(neg.to(DEV), -1, pre_relu=True)) == 0.0


def test_relu_prune_site_fold_is_bit_identical():
    """convert() builds Sequential(ReLU, PruneLayer) for a prune-only activation site; with the fold relu(x) is never
    materialised (importance of max(x,0), y = max(x,0)*mask, backward gate*g*mask).  Outputs, gradients, masks,
    magnitudes and counters equal the module-by-module run and the oracle, before the start step, while the schedule
    ramps, in steady state and in evaluation."""
    from qsparse_amd.fused import FusedActPrune, fuse_prune_quantize_pairs
    for dtype, shape in ((torch.bfloat16, (8, 32, 8, 8)), (torch.float32, (4, 16, 7, 7)), (torch.float32, (3, 8, 6, 6)), (torch.bfloat16, (5, 16, 14, 14)), (torch.float16, (6, 24, 4, 8))):
        runs = []
        for fold in (True, False):
            qs.set_qsparse_options(fold_relu=fold)
            site = nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=2, interval=1, repetition=2)).to(DEV).train()
            fuse_prune_quantize_pairs(site)
            assert type(site) is FusedActPrune and str(site).startswith("Sequential(")
            ps = O.PruneSim(0.5, [1], 2, 1, 2, False)
            outs = []
            for s in range(8):
                training = s < 7
                if not training:
                    site.eval()
                x = (torch.randn(shape, generator=gen(3000 + s)) * torch.linspace(0.3, 3, shape[1]).view(1, -1, 1, 1)).to(dtype)
                x.view(-1)[:4] = torch.tensor([0.0, -1e-3, 1.0, -1.0]).to(dtype)
                gout = torch.randn(shape, generator=gen(3100 + s)).to(dtype)
                xg = x.to(DEV).requires_grad_(True)
                y = site(xg)
                y.backward(gout.to(DEV))
                n_before = ps.n_updates
                y_ref = ps.step(torch.relu(x), training)
                gh = ps.grad(gout, (not training) or n_before >= 2)
                gx_ref = torch.where(x <= 0, torch.zeros_like(gh), gh)
                assert same(y.detach().cpu(), y_ref), (dtype, fold, s)
                assert same(xg.grad.cpu(), gx_ref), (dtype, fold, s)
                assert same(site[1].mask.cpu(), ps.mask) and int(site[1]._n_updates) == ps.n_updates, (dtype, fold, s)
                if ps.magnitude is not None:
                    assert same(site[1].callback.magnitude.cpu(), ps.magnitude) and int(site[1].callback.t) == ps.t
                outs.append((y.detach().clone(), xg.grad.clone()))
            runs.append(outs)
        for (ya, ga), (yb, gb) in zip(*runs):
            assert torch.equal(ya, yb) and same(ga.cpu(), gb.cpu())
    qs.set_qsparse_options(fold_relu=True)


def test_out_of_range_codes_follow_the_cpu_conversion():
    """a zero scale (all-zero tensor) or a huge quotient leaves the int32 range; ATen's CPU cast then yields INT_MIN
    (x86 integer indefinite) and the reference's output is float(INT_MIN) * s -- e.g. -0.0 everywhere for s == 0.
    The kernels reproduce that instead of the GPU's saturating conversion."""
    x = torch.tensor([0.0, -0.0, 1.0, -1.0, 3e38, -3e38, float("inf"), float("-inf"), float("nan"), 5.0, 1e-30, 7.5] + [0.25] * 20)
    for s in (0.0, 1e-38, 1e-30, 0.5):
        sc = torch.tensor([[s]])
        y, codes = _hip.quant_fwd("scaler", x.to(DEV), sc.to(DEV), -1, torch.float32, want_codes=True)
        assert same(codes.cpu(), O.scaler_codes(x, sc.view(1), -1)), s
        assert same(y.cpu(), O.scaler_fwd(x, 8, sc.view(1))), s
    for d in (0.0, 100.0, 127.0, -3.0):
        dc = torch.tensor([[d]])
        y, codes = _hip.quant_fwd("decimal", x.to(DEV), dc.to(DEV), -1, torch.float32, want_codes=True)
        assert same(codes.cpu(), O.decimal_codes(x, dc.view(1), -1).view(-1)), d
        assert same(y.cpu(), O.decimal_fwd(x, 8, dc.view(1)).view(-1)), d
    # an activation that is zero everywhere: the layer's scale becomes 0 and stays usable
    q_gpu, q_cpu = (qs.quantize(bits=4, channelwise=-1, timeout=1) for _ in range(2))
    q_gpu.to(DEV)
    z = torch.zeros(2, 4, 3, 3)
    for step in range(3):
        xin = z if step < 2 else z + torch.randn(2, 4, 3, 3, generator=gen(7))
        assert same(q_gpu(xin.to(DEV)).cpu(), q_cpu(xin)), step


@pytest.mark.parametrize("shape,dtype,site,fold", [
    ((1, 16), torch.bfloat16, "relu_pair", False), ((1, 64), torch.float16, "pair", True), ((1, 24), torch.float32, "relu_pair", True),
    ((1, 16, 1, 1), torch.float16, "relu_pair", True), ((1, 130, 1), torch.float16, "relu_pair", False),
    ((1, 32, 7, 7), torch.bfloat16, "pair", True), ((1, 32, 8, 8), torch.bfloat16, "relu_pair", True),
    ((1, 3), torch.float16, "relu_p", True), ((1, 6), torch.float32, "relu_q", True), ((2, 3), torch.bfloat16, "q", True),
    ((3, 2, 1, 1), torch.float32, "pair", True), ((1, 5), torch.float32, "relu_p", True)])
def test_sites_with_a_batch_of_one_and_tiny_tensors(shape, dtype, site, fold):
    """regressions found by tests/fuzz/fuzz_parity.py: a batch of one leaves the fused pair without a statistics stage in
    front of the channel dim (the abs-max must not ride along), tensors of fewer than 8 elements, masks that cover
    every element under the ReLU fold."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "tests", "fuzz", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    try:
        for kind, bits, sparsity, start, timeout in (("scaler", 4, 0.25, 0, 2), ("decimal", 8, 0.5, 1, 1), ("scaler", 2, 0.5, 2, 3)):
            r = fz.run_site(shape, dtype, site, kind, bits, sparsity, start, 1, 2, timeout, fold, 6, 5)
            assert r in ("ok", None), r
    finally:
        qs.set_qsparse_options(fold_relu=True)


def test_tensor_wise_bias_keeps_the_reference_output_shape():
    """a (1,1) tensor-wise parameter broadcasts a 1-d input (a bias) to (1, C) in the reference; nn.Linear accepts that
    bias, nn.Conv2d raises -- on the GPU exactly as on the CPU."""
    x = torch.randn(12, generator=gen(3))
    for fn, p in ((quantize_with_scaler, torch.tensor([[0.05]])), (quantize_with_decimal, torch.tensor([[5.0]]))):
        y_cpu, y_gpu = fn(x, 8, p, -1), fn(x.to(DEV), 8, p.to(DEV), -1)
        assert y_cpu.shape == (1, 12) and same(y_gpu.cpu(), y_cpu)
        xg = x.to(DEV).requires_grad_(True)
        fn(xg, 8, p.to(DEV), -1).backward(torch.ones(1, 12, device=DEV))
        assert xg.grad.shape == (12,)
    assert quantize_with_line(x.to(DEV), 8, torch.tensor([[-1.0, 1.0]], device=DEV), -1).shape == (12,)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    lin = qs.quantize(nn.Linear(16, 12), bits=4, channelwise=-1, timeout=1, bias_bits=8).to(DEV).train()
    conv = qs.quantize(nn.Conv2d(3, 8, 3), bits=4, channelwise=-1, timeout=1, bias_bits=8).to(DEV).train()
    for _ in range(3):
        assert lin(torch.randn(4, 16, device=DEV)).shape == (4, 12)
    assert lin.bias.shape == (1, 12)
    conv(torch.randn(2, 3, 8, 8, device=DEV))               # identity phase: the raw 1-d bias
    with pytest.raises(RuntimeError):
        conv(torch.randn(2, 3, 8, 8, device=DEV))           # (1, 8) bias, as in the reference


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_channels_last_activations_are_used_in_place(dtype):
    """dense channels_last (NHWC) activations -- what MIOpen's convolutions prefer -- are addressed in memory order
    without an NCHW copy: outputs and gradients come back channels_last, element-wise results equal the NCHW ones,
    and the channel statistics follow ATen's summation order for that layout (reproduced for up to 32 CPU threads;
    the reference's own 128-thread result differs from its 8-thread one)."""
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        # (H*W % 4 positions go through the row-sum-order kernel: n / 4 >= 32 exercises its carries, n % 4 its last rows)
        for shape in ((16, 32, 14, 14), (33, 64, 7, 7), (40, 8, 5, 6), (64, 16, 3, 9), (130, 16, 3, 3), (3, 8, 3, 3), (259, 8, 1, 3)):
            N, C, H, W = shape
            x = (torch.randn(shape, generator=gen(11)) * torch.linspace(0.3, 3, C).view(1, C, 1, 1)).to(dtype)
            xcl = x.contiguous(memory_format=torch.channels_last)
            # statistics: staged mean to the channels and the fused per-channel abs-max
            am = _hip.amax_accumulator(C, DEV)
            from qsparse_amd.util import _staged_mean_hip
            imp = _staged_mean_hip(xcl.to(DEV), [0, 2, 3], take_abs=True, absmax_out=am, absmax_channel_dim=1)
            assert same(imp.cpu(), O.squeeze_mean(xcl.abs(), (1, C, 1, 1))), shape
            assert same(_hip.amax_values(am).cpu(), x.abs().float().amax(dim=(0, 2, 3))), shape
            assert same(squeeze_tensor_to_shape(xcl.to(DEV), (1, C, 1, 1)).cpu(), squeeze_tensor_to_shape(xcl, (1, C, 1, 1)))
            # element-wise kernels: same values as NCHW, channels_last out, per-channel mask and tensor-wise scale
            mask = torch.rand(C, generator=gen(12)) > 0.4
            s = torch.tensor([[0.07]])
            for relu in (False, True):
                y_cl, _ = _hip.quant_fwd("scaler", xcl.to(DEV), s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV),
                                         mask_channel_index=1, pre_relu=relu)
                y_nc, _ = _hip.quant_fwd("scaler", x.to(DEV), s.to(DEV), -1, torch.float32, chan_mask=mask.to(DEV),
                                         mask_channel_index=1, pre_relu=relu)
                assert y_cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(y_cl, y_nc), (shape, relu)
            g = torch.randn(shape, generator=gen(13)).contiguous(memory_format=torch.channels_last)
            gx_cl = _hip.ste_relu_bwd(g.to(DEV), xcl.to(DEV), s.to(DEV), False, -8.0, 7.0, mask.to(DEV))
            gx_nc = _hip.ste_relu_bwd(g.contiguous().to(DEV), x.to(DEV), s.to(DEV), False, -8.0, 7.0, mask.to(DEV))
            assert gx_cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(gx_cl, gx_nc), shape
            m4 = mask.view(1, C, 1, 1)
            y = apply_mask(xcl.to(DEV), m4.to(DEV))
            assert y.is_contiguous(memory_format=torch.channels_last) and same(y.cpu(), xcl * m4), shape
            assert same(_hip.absmax(xcl.to(DEV), 1).cpu(), x.abs().float().amax(dim=(0, 2, 3)))
        # a whole fused site: channels_last in, channels_last out and grad, same numbers as the NCHW run
        from qsparse_amd.fused import fuse_prune_quantize_pairs
        runs = []
        for cl in (False, True):
            pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                                 qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
            fuse_prune_quantize_pairs(pair)
            outs = []
            for step in range(4):
                x = (torch.randn(16, 32, 8, 8, generator=gen(20 + step)) * torch.linspace(0.3, 3, 32).view(1, 32, 1, 1)).to(dtype)
                xg = (x.contiguous(memory_format=torch.channels_last) if cl else x).to(DEV).requires_grad_(True)
                y = pair(xg)
                y.backward(torch.ones_like(y))
                if cl:
                    assert y.is_contiguous(memory_format=torch.channels_last) and xg.grad.is_contiguous(memory_format=torch.channels_last)
                outs.append((y.detach().cpu().contiguous(), xg.grad.cpu().contiguous(), pair[0][1].mask.cpu().clone()))
            runs.append(outs)
        if dtype == torch.bfloat16:      # bf16 means absorb the 1-ulp order difference between the two layouts
            for (ya, ga, ma), (yb, gb, mb) in zip(*runs):
                assert torch.equal(ma, mb) and torch.equal(ya, yb) and torch.equal(ga, gb)
    finally:
        torch.set_num_threads(threads)


def test_fp16_scale_quotient_is_rounded_to_fp16_and_relu_keeps_negative_zero():
    """two fuzz finds: (1) the reference divides max|x| by 2^(bits-1) in x's dtype, so a small fp16 maximum lands on an
    fp16 subnormal (1e-3 / 128 -> 7.8082e-6, not 7.8157e-6); (2) ATen's CPU relu returns -0.0 for -0.0, and so does
    the ReLU folded into the prune site."""
    x = torch.zeros(2, 8, 4, 4, dtype=torch.float16)
    x[0, 1, 0, 0] = -1e-3
    x[1, 3, 2, 1] = 7e-4
    for fused in (False, True):
        q_cpu, q_gpu = (qs.quantize(bits=8, channelwise=-1, timeout=1) for _ in range(2))
        p_cpu, p_gpu = (qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1) for _ in range(2))
        site_cpu = nn.Sequential(nn.Sequential(nn.Identity(), p_cpu), q_cpu) if fused else q_cpu
        site_gpu = (nn.Sequential(nn.Sequential(nn.Identity(), p_gpu), q_gpu) if fused else q_gpu).to(DEV)
        if fused:
            from qsparse_amd.fused import fuse_prune_quantize_pairs
            fuse_prune_quantize_pairs(site_gpu)
        for _ in range(3):
            y_cpu, y_gpu = site_cpu(x), site_gpu(x.to(DEV))
            assert same(y_gpu.cpu(), y_cpu)
        assert same(q_gpu.weight.detach().cpu(), q_cpu.weight.detach()) and float(q_cpu.weight) < 7.82e-6
    from qsparse_amd.fused import FusedActPrune, fuse_prune_quantize_pairs
    z = torch.tensor([[-0.0, 0.0, -1.0, 2.0]] * 2).view(2, 4, 1, 1).half()
    sites = [fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1)))
             for _ in range(2)]
    sites[1].to(DEV)
    assert type(sites[1]) is FusedActPrune
    warm = torch.rand(2, 4, 1, 1).half() + 0.5      # the first call initialises the layer and runs module by module
    sites[0](warm), sites[1](warm.to(DEV))            # (torch's own GPU ReLU returns +0.0 for -0.0, its CPU ReLU -0.0)
    for _ in range(2):
        y_cpu, y_gpu = sites[0](z), sites[1](z.to(DEV))
        assert same(y_gpu.cpu(), y_cpu)


def test_multi_tensor_weight_kernels_match_the_per_tensor_ones():
    """qs_multi_absmax / qs_multi_scale_update / qs_multi_quant_fwd / qs_multi_ste_bwd over one device-resident table of 70
    tensors -- tensor-wise and per channel along the first, a middle and the last dim, ragged sizes, 4-byte aligned starts,
    rows that only quantize, a saturating row, a weight / bias pair sharing one counter -- against qs_absmax / qs_scale_update
    / qs_quant_*_fwd / qs_quant_ste_bwd one tensor at a time."""
    shapes = [(1,), (5,), (8,), (9,), (63,), (64,), (100,), (1000,), (4096,), (36864,), (147456,), (65,), (12345,)] * 3
    # (shape, channel dim): conv weights along dim 1 and 0, linear weights along dim 1 (the last), a bias along dim 0,
    # 1x1 convolutions (tall and narrow), a single-channel dim, a 3-d tensor along its middle dim
    per_channel = [((64, 32, 3, 3), 1), ((64, 32, 3, 3), 0), ((10, 512), 1), ((48,), 0), ((2048, 96, 1, 1), 1), ((7, 5, 3), 1),
                   ((16, 1, 5, 5), 1), ((3, 700), 1), ((700, 3), 1), ((513, 257), 0), ((4, 9, 1, 1), 1), ((128, 64, 1, 1), 0),
                   ((33, 17, 2), 2), ((1, 40), 1), ((40, 1), 0)]
    specs = [(sh, -1) for sh in shapes] + per_channel * 2
    specs = specs[:70]
    n = len(specs)
    pool = torch.empty(sum(int(np.prod(sh)) + 16 for sh, _ in specs), device=DEV)      # tensors carved at 4-byte aligned starts
    wd, off = [], 0
    for i, (sh, ci) in enumerate(specs):
        numel = int(np.prod(sh))
        start = off + (i % 4)                                     # only every fourth tensor is 16-byte aligned
        w = pool[start:start + numel].view(sh)
        w.copy_((torch.randn(sh, generator=gen(300 + i)) * (0.05 + 0.01 * i)).to(DEV))
        wd.append(w)
        off = start + numel + 8
    bits = [2 + (i % 7) for i in range(n)]
    ts = [i % 4 for i in range(n)]
    geo = [(1, 1, w.numel()) if ci < 0 or w.shape[ci] == 1 else
           (int(np.prod(w.shape[:ci])), w.shape[ci], int(np.prod(w.shape[ci + 1:]))) for w, (sh, ci) in zip(wd, specs)]
    chans = [g[1] for g in geo]
    train = [i % 9 != 4 for i in range(n)]                        # some rows only quantize (evaluation hand-outs)
    sat = {11: (-3, 2), 40: (0, 7)}                                # two saturating rows
    for decimal in (False, True):
        scales0 = [torch.rand(c, 1, generator=gen(400 + i)) * 0.1 + 0.01 for i, c in enumerate(chans)]
        # reference: one tensor at a time through the per-tensor entry points
        ref_scales, ref_y, ref_gx = [], [], []
        for i, (w, (sh, ci)) in enumerate(zip(wd, specs)):
            ci_eff = ci if chans[i] > 1 else -1
            s = scales0[i].clone().to(DEV)
            wc = w.contiguous().clone()                           # (16-byte aligned copy for the per-tensor entry points)
            if train[i]:
                t = ts[50] + 1 if i == 51 else ts[i]              # row 51 plays the bias of row 50: shared counter, t + 1
                _hip.scale_update(_hip.absmax(wc, ci_eff), s.view(-1), t, bits[i])
            ref_scales.append(s)
            param = _hip.decimal_from_scale(s.view(-1)).view(-1, 1) if decimal else s
            y, _ = _hip.quant_fwd("decimal" if decimal else "scaler", wc, param if chans[i] > 1 else param.view(1, 1), ci_eff,
                                  torch.float32, saturate=sat.get(i))
            ref_y.append(y)
            g = torch.randn(sh, generator=gen(500 + i)).to(DEV)
            ref_gx.append(_hip.ste_bwd(g, param if chans[i] > 1 else param.view(1, 1), decimal, ci_eff, -4.0, 3.0, False, torch.float32))
        total_c = sum(chans)
        amax, decs, backup = (torch.zeros(total_c, device=DEV) for _ in range(3))
        scales = [s0.clone().to(DEV) for s0 in scales0]
        t_devs = [torch.tensor([ts[i]], dtype=torch.int64, device=DEV) for i in range(n)]
        t_devs[51] = t_devs[50]                                   # the shared counter
        ts_before = [int(t.item()) for t in t_devs]
        bumps = torch.zeros(n, dtype=torch.int32, device=DEV)
        rows, offs, o, c0 = [], [], 0, 0
        for i, w in enumerate(wd):
            r = _hip.MultiRow()
            sl = slice(c0, c0 + chans[i])
            r.x, r.scale, r.amax, r.backup = w.data_ptr(), scales[i].data_ptr(), amax[sl].data_ptr(), backup[sl].data_ptr()
            r.decimal = decs[sl].data_ptr() if decimal else None
            r.t_dev, r.bump = t_devs[i].data_ptr(), bumps[i:i + 1].data_ptr()
            r.numel, r.y_off = w.numel(), o
            r.outer, r.C, r.inner = geo[i]
            r.train, r.is_decimal, r.t_offset = int(train[i]), int(decimal), int(i == 51)
            r.code_lo, r.code_hi = sat.get(i, (1, 0))
            r.denom = float(2 ** (bits[i] - 1))
            rows.append(r)
            offs.append(o)
            o += (w.numel() + 63) // 64 * 64
            c0 += chans[i]
        table = _hip.MultiTable(rows, wd[0].device)
        assert table.channels == total_c
        flat = torch.empty(o, device=DEV)
        _hip.multi_absmax(table)
        _hip.multi_scale_update(table)
        if decimal:      # rows that do not train keep their scale: their decimals come from the caller
            for i in range(n):
                if not train[i]:
                    decs[sum(chans[:i]):sum(chans[:i + 1])] = _hip.decimal_from_scale(scales[i].view(-1))
        _hip.multi_quant_fwd(table, flat, advance=True)
        for i, w in enumerate(wd):
            assert torch.equal(scales[i], ref_scales[i]), (decimal, i, specs[i])
            assert torch.equal(flat[offs[i]:offs[i] + w.numel()].view(w.shape), ref_y[i]), (decimal, i, specs[i])
            sl = slice(sum(chans[:i]), sum(chans[:i + 1]))
            if train[i]:
                assert torch.equal(backup[sl], scales0[i].view(-1).to(DEV)), i           # what the update replaced
        assert not amax.any()
        assert bumps.tolist() == [int(tr) for tr in train]
        for i in range(n):
            moved = int(t_devs[i].item()) - ts_before[i]
            assert moved == (2 if i in (50, 51) else int(train[i])), (i, moved)
        # the grouped STE backward, per channel where the step is
        gs = [torch.randn(sh, generator=gen(500 + i)).to(DEV) for i, (sh, _) in enumerate(specs)]
        gx = [torch.empty_like(g) for g in gs]
        steps = [(decs[sum(chans[:i]):sum(chans[:i + 1])] if decimal else scales[i]) for i in range(n)]
        _hip.multi_ste_bwd(n, _hip.ptr_array(gs), _hip.ptr_array(gx), _hip.ptr_array(steps), _hip.i64_array([g.numel() for g in gs]),
                           _hip.f32_array([-4.0] * n), _hip.f32_array([3.0] * n), decimal, wd[0].device,
                           channels=_hip.i32_array(chans), inners=_hip.i64_array([g[2] for g in geo]))
        for i in range(n):
            assert torch.equal(gx[i], ref_gx[i]), (decimal, i, specs[i])


@pytest.mark.parametrize("C", [48, 300, 2500])
def test_select_on_gathered_records_equals_combine_then_select(C):
    """data-parallel exchange: qs_pq_select given the all-gathered [world, 2C] records must do exactly what
    qs_stats_combine followed by qs_pq_select on the combined vectors does (and what the rank-ordered CPU formula says)."""
    world, k = 3, C // 3
    g = gen(900 + C)
    stages = [torch.rand(C, generator=g) + 0.01 for _ in range(world)]
    amaxes = [torch.rand(C, generator=g) * 4 for _ in range(world)]
    gathered = torch.cat([torch.cat([s, a]) for s, a in zip(stages, amaxes)]).to(DEV)
    acc = torch.zeros(C)
    for s in stages:
        acc = acc + s
    want_stage, want_amax = acc / world, torch.stack(amaxes).amax(0)

    def run(use_gathered):
        mag = torch.linspace(0.1, 1.0, C).to(DEV)
        mask = torch.ones(C, dtype=torch.bool, device=DEV)
        scale = torch.full((1, 1), 0.25, device=DEV)
        am = _hip.amax_accumulator(C, DEV)
        if use_gathered:
            _hip.pq_select(mag, None, True, 2, True, k, mask, am, True, 2, 4, scale, gathered=gathered, world=world)
        else:
            stage = _hip.stats_combine(gathered, world, C, True, am)
            assert same(stage.cpu(), want_stage) and same(_hip.amax_values(am).cpu(), want_amax)
            _hip.pq_select(mag, stage, True, 2, True, k, mask, am, True, 2, 4, scale)
        assert float(_hip.amax_values(am).abs().max()) == 0.0          # accumulator left clean either way
        return mag.cpu(), mask.cpu(), scale.cpu()

    a, b = run(False), run(True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    mag0 = torch.linspace(0.1, 1.0, C)
    want_mag = (2 * mag0 + want_stage) / 3
    assert same(a[0], want_mag)
    want_mask = want_mag >= want_mag.sort().values[k]
    assert torch.equal(a[1], want_mask)
    new = want_amax[want_mask].max() / 8                       # 4 bits
    assert same(a[2].view(-1), ((2 * torch.tensor(0.25) + new) / 3).view(-1))


def test_channels_last_statistics_any_channel_count_masks_and_l0_vs_oracle():
    """channels_last activations whose batch dim is reduced first are summed in place, in the order ATen's CPU path uses
    for that layout (reference util.py:92-99 on a channels_last tensor): channel counts that are not a multiple of 8,
    masks other than the channel mask ((1,C,H,W), (1,C,H,1), (1,1,H,W), (1,C,1,W)) and the L0 variant
    (sparse.py:85-86) -- none of them pays an NCHW copy any more, all bit-identical to the oracle run with one CPU thread
    (from 4 threads on ATen's own result for a few small-channel shapes such as (64, 3, 14, 14) differs from its 1-thread result)."""
    from qsparse_amd.util import squeeze_tensor_to_shape
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for si, shape in enumerate(((16, 3, 5, 5), (64, 12, 7, 7), (33, 20, 3, 3), (40, 10, 14, 14), (128, 36, 2, 2), (17, 5, 1, 7),
                                    (64, 100, 6, 1), (48, 24, 9, 9), (32, 64, 8, 8))):
            N, C, H, W = shape
            for dt in (torch.float32, torch.bfloat16, torch.float16):
                x = (torch.randn(shape, generator=gen(900 + si)) * torch.linspace(0.3, 3, C).view(1, -1, 1, 1)).to(dt)
                xcl = x.contiguous(memory_format=torch.channels_last)
                for mshape in ((1, C, 1, 1), (1, C, H, W), (1, C, H, 1), (1, 1, H, W), (1, C, 1, W), (1, 1, 1, 1)):
                    ref = O.squeeze_mean(xcl.abs(), mshape)
                    got = squeeze_tensor_to_shape(xcl.to(DEV).abs(), mshape)
                    assert got.shape == ref.shape and same(got.cpu().contiguous(), ref.contiguous()), (shape, dt, mshape)
        # whole PruneLayer trajectories on channels_last inputs: C % 8 != 0, L0 magnitudes, a (1, C, H, W) mask
        for kw, dims, shape in ((dict(), {1}, (8, 12, 6, 6)), (dict(l0=True), {1}, (8, 12, 6, 6)), (dict(l0=True), {1}, (8, 16, 6, 6)),
                                (dict(), {1, 2, 3}, (8, 6, 5, 5))):
            for dt in (torch.float32, torch.bfloat16):
                sim = O.PruneSim(0.5, sorted(dims), 1, 1, 2, False, l0=kw.get("l0", False))
                layer = qs.prune(sparsity=0.5, dimensions=dims, start=1, interval=1, repetition=2,
                                 callback=qs.MagnitudePruningCallback(**kw)).to(DEV).train()
                for s in range(6):
                    x = (torch.randn(shape, generator=gen(950 + s)) * torch.linspace(0.2, 2, shape[1]).view(1, -1, 1, 1)).relu().to(dt)
                    xcl = x.contiguous(memory_format=torch.channels_last)
                    y = layer(xcl.to(DEV))
                    y_ref = sim.step(xcl)
                    assert same(y.cpu().contiguous(), y_ref.contiguous()), (kw, dims, dt, s)
                    assert same(layer.mask.cpu(), sim.mask), (kw, dims, dt, s)
                    if sim.magnitude is not None:
                        assert same(layer.callback.magnitude.cpu(), sim.magnitude), (kw, dims, dt, s)
    finally:
        torch.set_num_threads(threads)


def test_channels_last_statistics_with_the_batch_dim_kept_vs_oracle():
    """channels_last activations whose batch dim is NOT reduced -- every batch of one (dim 0 equals the mask's), and per-sample
    masks (N, C, 1, 1): ATen reduces H of the NHWC tensor in place, in an order of its own (a last float32 bit away from the
    NCHW copy's); `_staged_mean_hip` reproduces it through qs_mean_dim_cl, one launch per sample
    (tests/test_aten_contract.py pins the identity it rests on).  Bit-identical to the oracle at one CPU thread: the plain
    squeeze, L0, and whole PruneLayer / fused-pair trajectories at batch one."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    from qsparse_amd.util import squeeze_tensor_to_shape
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for si, shape in enumerate(((1, 64, 14, 14), (1, 3, 28, 28), (1, 256, 56, 56), (1, 130, 5, 9), (1, 8, 7, 7), (3, 64, 14, 14),
                                    (5, 12, 9, 17), (1, 24, 2, 2), (1, 16, 40, 3))):
            N, C, H, W = shape
            for dt in (torch.float32, torch.bfloat16, torch.float16):
                x = (torch.randn(shape, generator=gen(1200 + si)) * torch.linspace(0.3, 3, C).view(1, -1, 1, 1)).to(dt)
                xcl = x.contiguous(memory_format=torch.channels_last)
                for mshape in ((N, C, 1, 1), (N, 1, H, W), (N, 1, 1, 1), (N, C, 1, W), (N, 1, 1, W), (N, 1, H, 1)):
                    ref = O.squeeze_mean(xcl.abs(), mshape)
                    got = squeeze_tensor_to_shape(xcl.to(DEV).abs(), mshape)
                    assert got.shape == ref.shape and same(got.cpu().contiguous(), ref.contiguous()), (shape, dt, mshape)
        # channels_last_3d activations, batch reduced first
        for si, shape in enumerate(((8, 16, 3, 4, 8), (5, 32, 2, 7, 7), (16, 3, 4, 4, 4), (3, 130, 1, 5, 9))):
            for dt in (torch.float32, torch.bfloat16):
                x = (torch.randn(shape, generator=gen(1230 + si)) * torch.linspace(0.3, 3, shape[1]).view(1, -1, 1, 1, 1)).to(dt)
                xcl = x.contiguous(memory_format=torch.channels_last_3d)
                for mshape in ((1, shape[1], 1, 1, 1), (1, shape[1]) + shape[2:], (1, 1, 1, 1, 1)):
                    ref = O.squeeze_mean(xcl.abs(), mshape)
                    got = squeeze_tensor_to_shape(xcl.to(DEV).abs(), mshape)
                    assert got.shape == ref.shape and same(got.cpu().contiguous(), ref.contiguous()), (shape, dt, mshape)
        for kw, shape in ((dict(), (1, 64, 14, 14)), (dict(l0=True), (1, 24, 6, 6)), (dict(), (1, 130, 5, 9))):
            for dt in (torch.float32, torch.bfloat16):
                sim = O.PruneSim(0.5, [1], 1, 1, 2, False, l0=kw.get("l0", False))
                layer = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2,
                                 callback=qs.MagnitudePruningCallback(**kw)).to(DEV).train()
                for s in range(6):
                    x = (torch.randn(shape, generator=gen(1250 + s)) * torch.linspace(0.2, 2, shape[1]).view(1, -1, 1, 1)).relu().to(dt)
                    xcl = x.contiguous(memory_format=torch.channels_last)
                    y = layer(xcl.to(DEV))
                    y_ref = sim.step(xcl)
                    assert same(y.cpu().contiguous(), y_ref.contiguous()), (kw, shape, dt, s)
                    assert same(layer.mask.cpu(), sim.mask), (kw, shape, dt, s)
                    if sim.magnitude is not None:
                        assert same(layer.callback.magnitude.cpu(), sim.magnitude), (kw, shape, dt, s)
        # the fused ReLU -> prune -> quantize site at batch one
        for dt in (torch.float32, torch.bfloat16):
            shape = (1, 64, 14, 14)
            site = fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1,
                                                                                               repetition=2)),
                                                           qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train())
            ps, qsim = O.PruneSim(0.5, [1], 1, 1, 2, False), O.QuantizeSim("scaler", 4, -1, 1)
            for s in range(6):
                x = (torch.randn(shape, generator=gen(1300 + s)) * torch.linspace(0.2, 2, 64).view(1, -1, 1, 1)).to(dt)
                xcl = x.contiguous(memory_format=torch.channels_last)
                y = site(xcl.to(DEV))
                y_ref = qsim.step(ps.step(torch.relu(xcl), True).contiguous(), True)
                assert same(y.cpu().contiguous(), y_ref.contiguous()), (dt, s)
                assert same(site[0][1].mask.cpu(), ps.mask), (dt, s)
                if ps.magnitude is not None:
                    assert same(site[0][1].callback.magnitude.cpu(), ps.magnitude), (dt, s)
                assert qsim.weight is None or same(site[1].weight.detach().cpu(), qsim.weight), (dt, s)
    finally:
        torch.set_num_threads(threads)


def test_folded_relu_propagates_nan_like_the_module_by_module_path():
    """ATen's relu keeps NaN (clamp_min = max(0, x)); the folded kernels must not turn a diverged activation into a clean
    zero.  A NaN input gives the same (NaN-carrying) scale, output and gradient with and without the fold -- the fused
    pair, the ReLU -> quantize site and the ReLU -> prune site."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    shape = (4, 16, 8, 8)
    for si, make in enumerate((
            lambda: nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=1)),
            lambda: nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                                  qs.quantize(bits=4, channelwise=-1, timeout=1)),
            lambda: nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)))):
        runs = []
        # (fold, elide_pruned): the default mode is exact -- the folded site records the ReLU gate and loads every element, the
        # unfolded NCHW pair elides through the select's elision mask, which has a pruned NaN channel loaded -- so all four agree
        for fold, mode in ((True, "forward"), (False, "off"), (True, "off"), (False, "forward")):
            qs.set_qsparse_options(fold_relu=fold, elide_pruned=mode)
            site = fuse_prune_quantize_pairs(make().to(DEV).train())
            outs = []
            for s in range(4):
                x = torch.randn(shape, generator=gen(3100 + s)).bfloat16()
                if s == 2:
                    x[1, 3, 2, 2] = float("nan")        # a kept or pruned channel, whichever the mask says
                    x[0, 5, 0, 0] = float("nan")
                xg = x.to(DEV).requires_grad_(True)
                y = site(xg)
                y.backward(torch.ones_like(y))
                outs.append((y.detach().cpu(), xg.grad.cpu()))
            runs.append(outs)
        qs.set_qsparse_options(fold_relu=True, elide_pruned="forward")
        for s, ((ya, ga), (yb, gb), (yc, gc), (yd, gd)) in enumerate(zip(*runs)):
            assert same(ya, yb) and same(ga, gb) and same(ya, yc) and same(ga, gc) and same(ya, yd) and same(ga, gd), s
        if si != 1:     # (in the pair a NaN channel has NaN magnitude, is pruned, and never reaches the scale)
            assert bool(runs[0][2][0].isnan().any())      # the NaN is visible in the step that received it


def test_fused_sites_with_hooks_on_children_run_module_by_module():
    """forward (pre-)hooks registered on the PruneLayer / QuantizeLayer / ReLU children of a convert-built site keep
    firing on the GPU: a hooked site falls back to q(p(act(x))) -- same numbers, the fused path just is not taken."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    shape = (4, 16, 8, 8)

    def build():
        torch.manual_seed(0)
        return fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train())

    plain, hooked = build(), build()
    seen = {"relu": 0, "prune": 0, "quant": 0, "pre": 0}
    hooked[0][0].register_forward_hook(lambda m, i, o: seen.__setitem__("relu", seen["relu"] + 1))
    hooked[0][1].register_forward_hook(lambda m, i, o: seen.__setitem__("prune", seen["prune"] + 1))
    hooked[1].register_forward_hook(lambda m, i, o: seen.__setitem__("quant", seen["quant"] + 1))
    hooked[1].register_forward_pre_hook(lambda m, i: seen.__setitem__("pre", seen["pre"] + 1))
    for s in range(5):
        x = torch.randn(shape, generator=gen(3200 + s)).bfloat16().to(DEV)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = plain(xa), hooked(xb)
        ya.backward(torch.ones_like(ya)), yb.backward(torch.ones_like(yb))
        assert same(ya.detach().cpu(), yb.detach().cpu()) and same(xa.grad.cpu(), xb.grad.cpu()), s
        assert same(plain[0][1].mask.cpu(), hooked[0][1].mask.cpu())
    assert seen == {"relu": 5, "prune": 5, "quant": 5, "pre": 5}


def test_channel_mismatch_in_eval_raises_runtime_error_like_the_reference():
    """reference tests/test_sparse.py:75-96: a channel mask meets an input with another channel count in evaluation mode
    -> RuntimeError (the broadcast of x * mask), also on the fused GPU path (it used to be an AssertionError)."""
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    pair = fuse_prune_quantize_pairs(nn.Sequential(
        nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
        qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train())
    for s in range(3):
        pair(torch.randn(4, 16, 8, 8, generator=gen(3300 + s)).to(DEV))
    pair.eval()
    assert pair(torch.randn(2, 16, 5, 5).to(DEV)).shape == (2, 16, 5, 5)        # other spatial size: the channel mask broadcasts (quirk B17)
    with pytest.raises(RuntimeError):
        pair(torch.randn(2, 12, 8, 8).to(DEV))
    lone = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1).to(DEV).train()
    for s in range(3):
        lone(torch.randn(4, 16, 8, 8, generator=gen(3400 + s)).to(DEV))
    lone.eval()
    with pytest.raises(RuntimeError):
        lone(torch.randn(2, 12, 8, 8).to(DEV))


def test_minmax_propagates_nan_like_torch():
    """AdaptiveQuantizer's statistics are `rows.min(dim=1)` / `rows.max(dim=1)` (reference quantize.py:410-411): torch
    propagates a NaN into BOTH; so do the kernels (v_minimum3_f32 / v_maximum3_f32), on every route -- tensor-wise, per channel
    NCHW (rows, columns, the big tensors' column walk), channels_last, 2-d"""
    for shape, ci in (((64, 48, 9, 9), 1), ((3, 5, 64), 1), ((1000, 16), 1), ((70000,), -1), ((16, 128, 64, 64), 1), ((37, 12, 14, 14), 1),
                      ((8192, 256), 1), ((32, 1024, 8, 8), 1)):
        for dtype in (torch.float32, torch.bfloat16):
            for cl in ((False, True) if len(shape) == 4 else (False,)):
                x = (torch.randn(shape, generator=gen(4242)) * 3).to(dtype)
                flat = x.view(-1)
                flat[5], flat[flat.numel() // 2], flat[3] = float("nan"), float("nan"), -0.0
                xg = x.to(DEV).contiguous(memory_format=torch.channels_last) if cl else x.to(DEV)
                mn, mx = _hip.minmax(xg, ci)
                xf = x.float()
                if ci < 0:
                    want_mn, want_mx = xf.min().view(1), xf.max().view(1)
                else:
                    rows = xf.transpose(0, ci).reshape(shape[ci], -1)
                    want_mn, want_mx = rows.min(dim=1).values, rows.max(dim=1).values
                for got, want in ((mn.cpu(), want_mn), (mx.cpu(), want_mx)):
                    assert torch.equal(got.isnan(), want.isnan()) and want.isnan().any(), (shape, ci, dtype, cl)
                    assert torch.equal(got[~want.isnan()], want[~want.isnan()]), (shape, ci, dtype, cl)


def test_minmax_key_accumulation_equals_the_float_route():
    """qs_minmax(accumulate) leaves order-preserving keys in persistent buffers and qs_lines_update(from_keys) converts and
    resets them: the running (min, max) lines must carry the same bits as the four-launch route (key initialisation,
    reduction, key -> float, running mean) over several steps -- tensor-wise and per channel, NCHW (rows, columns and the big
    tensors' column walk) and channels_last, with -0.0 and NaN in the data."""
    cases = (((64, 48, 9, 9), 1), ((3, 5, 64), 1), ((1000, 16), 1), ((16, 2048, 7, 7), 1), ((4, 6, 10, 10), 2), ((70000,), -1),
             ((37, 12, 14, 14), 1), ((50, 1024, 14, 14), 1), ((8192, 256), 1), ((16, 128, 64, 64), 1), ((32, 1024, 8, 8), 1))
    for si, (shape, ci) in enumerate(cases):
        for dtype in (torch.float32, torch.bfloat16):
            for cl in ((False, True) if len(shape) == 4 and ci == 1 else (False,)):
                n = shape[ci] if ci >= 0 else 1
                keys = _hip.minmax_key_buffers(n, DEV)
                la, lb = torch.zeros(n, 2, device=DEV), torch.zeros(n, 2, device=DEV)
                for t in range(3):
                    x = (torch.randn(shape, generator=gen(7000 + 10 * si + t)) * 3).to(dtype)
                    if t == 1:
                        x.view(-1)[0] = -0.0
                    if t == 2 and x.numel() > 100:
                        x.view(-1)[77] = float("nan")
                    xg = x.to(DEV)
                    if cl:
                        xg = xg.contiguous(memory_format=torch.channels_last)
                    kmn, kmx = _hip.minmax(xg, ci, accumulate_into=keys)
                    _hip.lines_update(kmn, kmx, la, t + 1, from_keys=True)
                    mn, mx = _hip.minmax(xg, ci)
                    _hip.lines_update(mn, mx, lb, t + 1)
                    assert same(la.cpu(), lb.cpu()), (shape, ci, dtype, cl, t)
                    assert bool((keys[0] == -1).all()) and not keys[1].any(), (shape, ci, dtype, cl, t)     # neutral again


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_channels_last_reduction_that_starts_with_w_follows_atens_order(dtype):
    """a mask that keeps N, C and H (`prune(dimensions={0, 1, 2})`): squeeze_tensor_to_shape reduces W alone, which ATen sums
    in row-sum order for a channels_last tensor (reference util.py:92-99 on such an input; qs_mean_cl_w, ABI v20)"""
    import qsparse_amd as qs
    from qsparse_amd.util import squeeze_tensor_to_shape
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        g = torch.Generator().manual_seed(2)
        for N, C, H, W in [(2, 8, 5, 7), (4, 16, 14, 14), (3, 24, 7, 56), (2, 64, 28, 28), (8, 3, 9, 33), (2, 5, 3, 100), (2, 8, 4, 300),
                           (2, 16, 2, 9), (1, 8, 5, 17)]:
            x = (torch.randn(N, C, H, W, generator=g) * 3).to(dtype).contiguous(memory_format=torch.channels_last)
            want = squeeze_tensor_to_shape(x, [N, C, H, 1])
            got = squeeze_tensor_to_shape(x.cuda(), [N, C, H, 1])
            assert got.dtype == want.dtype and torch.equal(got.cpu(), want), (N, C, H, W)
        # the prune layer end to end: importance, running magnitude, mask
        outs = []
        for dev in ("cpu", "cuda"):
            p = qs.prune(sparsity=0.5, dimensions={0, 1, 2}, start=1, interval=1, repetition=2).to(dev).train()
            gg = torch.Generator().manual_seed(3)
            rec = []
            for _ in range(5):
                x = (torch.randn(4, 8, 6, 10, generator=gg) * 2).to(dtype).contiguous(memory_format=torch.channels_last).to(dev)
                rec.append(p(x).cpu())
            outs.append(rec + [p.mask.cpu(), p.callback.magnitude.cpu()])
        for i, (a, b) in enumerate(zip(*outs)):
            assert torch.equal(a, b), i
    finally:
        torch.set_num_threads(threads)


def _permuted(x, perm):
    """the same values, dense in memory in the dim order `perm` (outermost first)"""
    inv = [perm.index(i) for i in range(x.dim())]
    return x.permute(perm).contiguous().permute(inv)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_reduction_of_any_dense_layout_follows_atens_order(dtype):
    """squeeze_tensor_to_shape (reference util.py:92-99) of a dense tensor in an arbitrary dim order -- a transposed weight, a
    permuted activation, NDHWC with the batch kept: ATen reduces it where it lies, in an order that depends on the layout
    (util.aten_reduce_plan, pinned on the CPU in tests/test_aten_contract.py); qs_mean_strided (ABI v22) executes that order on
    the GPU instead of reducing a contiguous copy in the copy's order"""
    import random

    import qsparse_amd as qs
    from qsparse_amd.util import squeeze_tensor_to_shape
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        rng, g = random.Random(11), torch.Generator().manual_seed(11)
        differs = 0
        for it in range(160):
            nd = rng.choice([2, 3, 4, 4, 5])
            shape = [rng.choice([1, 2, 3, 5, 7, 8, 9, 16, 33, 40, 64]) for _ in range(nd)]
            while int(torch.tensor(shape).prod()) > 60000:
                shape[rng.randrange(nd)] = 2
            perm = list(range(nd))
            rng.shuffle(perm)
            x = _permuted((torch.randn(shape, generator=g) * 3).to(dtype), perm)
            if it % 7 == 3 and shape[-1] > 2:           # a view that is not dense: `x.abs()` lays it out densely in its stride order
                x = x[..., 1:]
            target = [1 if (s > 1 and rng.random() < 0.5) else s for s in x.shape]
            want = squeeze_tensor_to_shape(x.abs(), target)
            got = squeeze_tensor_to_shape(x.cuda().abs(), target)
            assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got.cpu(), want), (shape, perm, target, it)
            differs += int(not torch.equal(want, squeeze_tensor_to_shape(x.abs().contiguous(), target)))
        # (the order of the contiguous copy, what the GPU path used until ABI v21, is another one: a last float32 bit, which the
        # rounding to a 16-bit result hides)
        assert differs >= 10 or dtype != torch.float32
        # the shapes the route exists for: [B, C, T] activations that are transposes of [B, T, C]; transposed linear weights
        for shape, perm, target in [((8, 96, 50), (0, 2, 1), (1, 96, 1)), ((4, 128, 33), (0, 2, 1), (4, 128, 1)),
                                    ((256, 384), (1, 0), (256, 1)), ((256, 384), (1, 0), (1, 384)),
                                    ((2, 16, 4, 6, 6), (0, 2, 3, 4, 1), (2, 16, 1, 1, 1)), ((3, 10, 12, 14), (2, 3, 0, 1), (1, 10, 1, 1)),
                                    ((4, 16, 1, 9), (0, 2, 3, 1), (4, 16, 1, 1)), ((4, 16, 1, 40), (0, 2, 3, 1), (1, 16, 1, 1))]:
            x = _permuted((torch.randn(shape, generator=g) * 3).to(dtype), list(perm))
            want = squeeze_tensor_to_shape(x.abs(), list(target))
            assert torch.equal(squeeze_tensor_to_shape(x.cuda().abs(), list(target)).cpu(), want), (shape, perm, target)
        # the prune layer end to end on a transposed activation: importance (its L0 first step included), running magnitude, mask
        outs = []
        for dev in ("cpu", "cuda"):
            p = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2).to(dev).train()
            gg = torch.Generator().manual_seed(3)
            rec = []
            for _ in range(5):
                x = (torch.randn(6, 20, 24, generator=gg) * 2).to(dtype).transpose(1, 2).to(dev)       # [B, C, T] view of [B, T, C]
                assert not x.is_contiguous()
                y = p(x)
                assert y.stride() == x.stride()         # `x * mask` comes back in x's layout (TensorIterator), on both devices
                rec.append(y.cpu())
            outs.append(rec + [p.mask.cpu(), p.callback.magnitude.cpu()])
        for i, (a, b) in enumerate(zip(*outs)):
            assert torch.equal(a, b), i
    finally:
        torch.set_num_threads(threads)


@pytest.mark.gpu
def test_results_come_back_in_the_layout_the_reference_returns():
    """the reference's operators are element-wise ATen chains (sparse.py:116, quantize.py:109-117): their result is laid out
    densely in the input's stride order, whatever that is.  The kernels address contiguous and channels_last tensors in place;
    any other order goes through a contiguous copy and `_hip.laid_out_like` puts the result back -- values, layout and the
    gradient agree with the CPU path"""
    import qsparse_amd as qs
    def pair():         # Sequential(Sequential(ReLU, PruneLayer), QuantizeLayer): the converted site of the --pq recipe
        net = qs.convert(torch.nn.Sequential(torch.nn.ReLU()), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2),
                         activation_layers=[torch.nn.ReLU], log=False)
        return qs.convert(net, qs.quantize(bits=6, timeout=1, channelwise=-1), activation_layers=[torch.nn.ReLU], log=False)

    def sites():
        yield "prune", lambda: qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2)
        yield "quantize", lambda: qs.quantize(bits=6, timeout=1, channelwise=-1)
        yield "quantize_cw", lambda: qs.quantize(bits=6, timeout=1, channelwise=1, callback=qs.AdaptiveQuantizer())
        yield "quantize_decimal", lambda: qs.quantize(bits=6, timeout=1, channelwise=-1, callback=qs.DecimalQuantizer())
        yield "pair", pair
        yield "act_quantize", lambda: qs.convert(torch.nn.Sequential(torch.nn.ReLU()), qs.quantize(bits=6, timeout=1, channelwise=-1),
                                                 activation_layers=[torch.nn.ReLU], log=False)

    inputs = [((6, 20, 24), (0, 2, 1)), ((4, 10, 6, 8), (0, 3, 2, 1)), ((4, 10, 6, 8), (2, 0, 3, 1)), ((3, 8, 5), (1, 0, 2)),
              ((4, 16, 6, 8), (0, 2, 3, 1))]
    for name, make in sites():
        for shape, perm in inputs:
            res = []
            for dev in ("cpu", "cuda"):
                torch.manual_seed(0)
                m = make().to(dev).train()
                gg = torch.Generator().manual_seed(7)
                rec = []
                for step in range(4):
                    if step == 3:
                        m.eval()
                    x = _permuted(torch.randn(shape, generator=gg) * 2, list(perm)).to(dev).requires_grad_(True)
                    y = m(x)
                    y.backward(torch.randn(shape, generator=gg).to(dev))
                    rec += [y.detach().cpu(), x.grad.cpu(), torch.tensor([st for st, n in zip(y.stride(), y.shape) if n > 1])]
                res.append(rec)
            for i, (a, b) in enumerate(zip(*res)):
                assert torch.equal(a, b), (name, shape, perm, i)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_long_rows_reduced_along_their_own_direction(dtype):
    """the inner reduction (the last dim of a contiguous tensor: a row mask over a Linear weight, the W stage of maps too large
    for the fused last-two-dims launch, the unit-stride dim of a transposed tensor) with half a wave per row
    (mean_inner_wave_kernel, n >= 64): ATen's vectorised inner sum, bit for bit (reference util.py:92-99)"""
    from qsparse_amd.util import squeeze_tensor_to_shape
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        g = torch.Generator().manual_seed(4)
        for pre in (1, 2, 3, 64, 257):
            for n in (63, 64, 65, 71, 96, 127, 128, 255, 256, 1000, 4096, 5003, 70000):
                if pre * n > 3_000_000:
                    continue
                x = (torch.randn(pre, n, generator=g) * 3).to(dtype)
                want = squeeze_tensor_to_shape(x.abs(), [pre, 1])
                got = squeeze_tensor_to_shape(x.cuda().abs(), [pre, 1])
                assert got.dtype == want.dtype and torch.equal(got.cpu(), want), (pre, n)
        # inside a 4-d staged mean: W = 300 maps do not fit the fused last-two-dims launch (LDS), the W stage is an inner reduction
        x = (torch.randn(2, 3, 230, 300, generator=g) * 3).to(dtype)
        for target in ([1, 3, 1, 1], [2, 3, 230, 1], [2, 1, 1, 1]):
            assert torch.equal(squeeze_tensor_to_shape(x.cuda().abs(), target).cpu(), squeeze_tensor_to_shape(x.abs(), target)), target
    finally:
        torch.set_num_threads(threads)
