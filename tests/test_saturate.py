"""Row N1 of VERDICT r03: the saturating forward -- north_star's "fused scale -> round -> clamp", SURVEY section 7 "a flag
defaulting to reference behaviour".

The reference spells the clamp out and loses it: `q.float().clamp_(lo, hi)` acts on a temporary (qsparse/quantize.py:56-62,
110-116), so its output never saturates (quirk B1, pinned by the golden fixtures F1 / F2).  With the public switch --
`set_qsparse_options(saturate=True)`, `ScalerQuantizer(saturate=True)`, `quantize_with_*(..., saturate=True)` -- the codes are
`clamp(q, lo, hi)` with `lo, hi = -2^(bits-1)+notch, 2^(bits-1)-1+notch` (or `0, 2^bits-1` with `use_uint`), i.e. that very
line with the assignment it lacks.  Expected values here are torch's own arithmetic on the CPU
(`clamp(round(x / s).int(), lo, hi).float() * s`; the decimal quantizer truncates); the GPU runs go through the kernels'
`saturate` branch on every route that quantizes: the functional API, lone layers (`qs_quantize_step`), the fused
prune -> quantize pair with and without the ReLU fold (`qs_site_fwd`, `_FusedApply`), the multi-tensor weight path
(`qs_multi_quant_fwd`), and the integer export (int8 / packed int4).
"""
import copy

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd.export import QuantizedTensor
from qsparse_amd.quantize import DecimalQuantizer, ScalerQuantizer, code_range, quantize_with_decimal, quantize_with_scaler

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def gen(seed):
    return torch.Generator().manual_seed(seed)


def same(a, b):
    """bit-for-bit (NaNs compare equal, -0.0 differs from +0.0)"""
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.dtype.is_floating_point:
        return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.float().view(torch.int32), b.float().view(torch.int32))
    return a.shape == b.shape and torch.equal(a, b)


def _view(p, x, ci):
    if not isinstance(p, torch.Tensor) or p.numel() == 1:
        return p
    shape = [1] * x.dim()
    shape[ci] = -1
    return p.view(shape)


def torch_scaler(x, bits, s, ci, lo, hi):
    s = _view(s, x, ci)
    q = (x / s).round().int().clamp(lo, hi)
    return q.float() * s, q


def torch_decimal(x, bits, d, ci, lo, hi):
    d = _view(d, x, ci)
    q = (x * 2.0 ** d).int().clamp(lo, hi)
    return q.float() * 2.0 ** -d, q


def test_code_range():
    assert code_range(4, 0, False, True) == (-8, 7) and code_range(4, 1, False, True) == (-7, 8)
    assert code_range(8, 0, True, True) == (0, 255) and code_range(8, 0, False, False) is None
    assert code_range(8) is None                                     # the option's default: the reference's behaviour
    qs.set_qsparse_options(saturate=True)
    try:
        assert code_range(8) == (-128, 127) and code_range(8, saturate=False) is None
        assert ScalerQuantizer().code_range(4) == (-8, 7) and ScalerQuantizer(saturate=False).code_range(4) is None
    finally:
        qs.set_qsparse_options(saturate=False)
    assert ScalerQuantizer(flip_axis=True, saturate=True).code_range(4) == (-7, 8)


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
def test_functional_saturation_equals_torch_arithmetic(dev, dtype, layout):
    x = (torch.randn(6, 16, 9, 8, generator=gen(1)) * 3).to(dtype)
    flat = x.view(-1)
    flat[:8] = torch.tensor([1e6, -1e6, float("inf"), float("-inf"), float("nan"), 0.0, -0.0, 0.7]).to(dtype)
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    per_chan = torch.linspace(0.05, 0.6, 16).view(-1, 1)
    dec = torch.tensor([3.0, 4, 5, 2, 1, 0, 6, 7, 3, 3, 2, 4, 1, 2, 3, 4])
    for bits in (4, 8):
        for flip, uint in ((False, False), (True, False), (False, True)):
            lo, hi = code_range(bits, 1 if flip else 0, uint, True)
            for s, ci in ((torch.tensor([[0.25]]), -1), (per_chan, 1)):
                want, wq = torch_scaler(x.float() if isinstance(s, torch.Tensor) else x, bits, s, ci, lo, hi)
                y, q = quantize_with_scaler(x.to(dev), bits, s.to(dev), ci, use_uint=uint, flip_axis=flip, return_codes=True,
                                            saturate=True)
                assert same(q, wq) and same(y, want), (bits, flip, uint, ci)
                assert int(q.min()) >= lo and int(q.max()) <= hi and int(q.max()) == hi and int(q.min()) == lo
                # and the default stays the reference's: the same call without the switch does not clamp
                y0, q0 = quantize_with_scaler(x.to(dev), bits, s.to(dev), ci, use_uint=uint, flip_axis=flip, return_codes=True)
                assert int(q0.max()) > hi
            for d, ci in ((2, -1), (dec, 1)):
                want, wq = torch_decimal(x.float(), bits, d, ci, lo, hi)
                y, q = quantize_with_decimal(x.to(dev), bits, d.to(dev) if isinstance(d, torch.Tensor) else d, ci, use_uint=uint,
                                             flip_axis=flip, return_codes=True, saturate=True)
                assert same(q, wq) and same(y, want), (bits, flip, uint, ci)
    # the backward is the reference's, saturated or not: gradient VALUES clamped to [lo * s, hi * s] (quantize.py:120-131)
    xg = x.float().to(dev).requires_grad_(True)
    g = torch.randn(x.shape, generator=gen(2)).to(dev)
    quantize_with_scaler(xg, 4, 0.25, saturate=True).backward(g)
    assert same(xg.grad, g.clamp(-8 * 0.25, 7 * 0.25))


def _steps(n, shape, dtype, seed=5):
    g = gen(seed)
    return [((torch.randn(shape, generator=g) * torch.linspace(0.3, 3, shape[1]).view(1, -1, 1, 1)).to(dtype),
             torch.randn(shape, generator=g)) for _ in range(n)]


def _run(net, data, dev, fmt=torch.contiguous_format):
    net = net.to(dev).train()
    outs = []
    for x, g in data:
        xd = x.detach().clone().to(dev).contiguous(memory_format=fmt).requires_grad_(True)
        y = net(xd)
        y.backward(g.to(dev))
        outs.append((y.detach().cpu(), xd.grad.detach().cpu()))
    return outs, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fmt", [torch.contiguous_format, torch.channels_last])
@pytest.mark.parametrize("quantizer", [ScalerQuantizer, DecimalQuantizer])
def test_sites_saturate_on_the_gpu_exactly_like_the_cpu_path(dtype, fmt, quantizer):
    """convert-built sites -- ReLU -> prune -> quantize (the fused pair with the ReLU fold), ReLU -> quantize, a lone quantizer
    behind a convolution-less Identity -- with `saturate` on: the HIP routes (composite site call, `_FusedApply`,
    `qs_quantize_step`) against the CPU path's module-by-module torch arithmetic, outputs, gradients and state bit for bit"""
    def build():
        torch.manual_seed(0)
        net = nn.Sequential(nn.ReLU(), nn.Identity(), nn.ReLU())
        net = qs.convert(net, qs.prune(sparsity=0.5, start=2, interval=1, repetition=2), activation_layers=[nn.ReLU],
                         excluded_activation_layer_indexes=[(nn.ReLU, [1])], log=False)
        net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=2, callback=quantizer(saturate=True)),
                         activation_layers=[nn.ReLU, nn.Identity], log=False)
        return net

    data = _steps(7, (6, 16, 10, 10), dtype)
    cpu, cpu_state = _run(build(), data, "cpu", fmt)
    gpu, gpu_state = _run(build(), data, "cuda", fmt)
    for (ya, ga), (yb, gb) in zip(cpu, gpu):
        assert same(ya, yb) and same(ga, gb)
    for k in cpu_state:
        assert same(cpu_state[k], gpu_state[k]), k
    # it did saturate: the last output holds at most 2^4 distinct levels per quantizer scale, the largest code being 7
    net = build().to("cuda").train()
    for x, _ in data:
        y = net(x.to("cuda").contiguous(memory_format=fmt))
    s = float(net[2][1].weight)
    q = (y / s).round() if quantizer is ScalerQuantizer else None
    if q is not None:
        assert float(q.max()) == 7.0
    # and through the global option instead of the quantizer's own flag
    qs.set_qsparse_options(saturate=True)
    try:
        def build_plain():
            torch.manual_seed(0)
            net = nn.Sequential(nn.ReLU(), nn.Identity(), nn.ReLU())
            net = qs.convert(net, qs.prune(sparsity=0.5, start=2, interval=1, repetition=2), activation_layers=[nn.ReLU],
                             excluded_activation_layer_indexes=[(nn.ReLU, [1])], log=False)
            return qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=2, callback=quantizer()),
                              activation_layers=[nn.ReLU, nn.Identity], log=False)
        opt, _ = _run(build_plain(), data, "cuda", fmt)
    finally:
        qs.set_qsparse_options(saturate=False)
    for (ya, ga), (yb, gb) in zip(gpu, opt):
        assert same(ya, yb) and same(ga, gb)


def _weight_net(quantizer, bits=4, channelwise=-1, **kw):
    torch.manual_seed(1)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.Flatten(), nn.Linear(8 * 36, 5))
    return qs.convert(net, qs.quantize(bits=bits, channelwise=channelwise, timeout=1, callback=quantizer(saturate=True)),
                      weight_layers=[nn.Conv2d, nn.Linear], log=False, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("quantizer", [ScalerQuantizer, DecimalQuantizer])
def test_weight_batcher_saturates_like_the_inline_path(quantizer):
    x = torch.randn(4, 3, 6, 6, generator=gen(7))
    results = []
    for dev, kw in (("cpu", {}), ("cuda", {"batch_weights": False}), ("cuda", {})):
        net = _weight_net(quantizer, **kw).to(dev).train()
        if dev == "cuda" and not kw:
            assert net.__dict__.get("_qs_weight_batcher") is not None
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        for _ in range(4):
            opt.zero_grad()
            net(x.to(dev)).square().mean().backward()
            opt.step()
        net.eval()
        with torch.no_grad():
            w = [m.weight.detach().cpu() for m in net if hasattr(m, "quantize")]
        results.append((w, {k: v.detach().cpu() for k, v in net.state_dict().items() if "quantize" in k}))
    (w_cpu, s_cpu), (w_inline, s_inline), (w_batched, s_batched) = results
    for a, b in zip(w_inline, w_batched):
        assert same(a, b)
    for k in s_inline:
        assert same(s_inline[k], s_batched[k]), k
    # every effective weight holds codes in [-8, 7] only
    for k, w in zip([k for k in s_batched if k.endswith("quantize.weight")], w_batched):
        s = float(s_batched[k])
        step = s if quantizer is ScalerQuantizer else 2.0 ** -round(torch.log2(torch.tensor(1 / s)).item())
        codes = (w / step).round()
        assert float(codes.max()) <= 7 and float(codes.min()) >= -8
        if quantizer is ScalerQuantizer:         # (a power-of-two step may leave the largest element below the top code)
            assert float(codes.max()) == 7


@pytest.mark.parametrize("dev", DEVICES)
def test_export_int8_and_packed_int4_under_saturation(dev):
    x = torch.randn(4, 3, 6, 6, generator=gen(8)).to(dev)
    for bits in (8, 4):
        net = _weight_net(ScalerQuantizer, bits=bits, channelwise=0).to(dev).train()
        for _ in range(3):
            net(x)
        net.eval()
        ex = qs.export_integer(net)
        assert len(ex) == 3
        for rec in ex.values():
            t = rec.weight
            lo, hi = int(t.codes.min()), int(t.codes.max())
            assert -2 ** (bits - 1) <= lo and hi == 2 ** (bits - 1) - 1          # the largest element sits ON the top code
            assert same(t.int8().to(torch.int32), t.codes)
            if bits == 4:
                packed = t.int4_packed()
                assert packed.dtype == torch.uint8 and packed.numel() == (t.codes.numel() + 1) // 2
                assert same(QuantizedTensor.unpack_int4(packed, t.codes.numel()).view(t.codes.shape), t.codes)
            assert same(t.dequantize(), dict(net.named_modules())[rec.path].weight)
    # without the switch the reference's quirk B1 is what an export meets: code +2^(bits-1) does not fit
    torch.manual_seed(1)
    plain = qs.quantize(nn.Conv2d(3, 8, 3), bits=8, channelwise=-1, timeout=1).to(dev).train()
    for _ in range(3):
        plain(x)
    (rec,) = qs.export_integer(plain).values()
    assert int(rec.weight.codes.abs().max()) == 128
    if int(rec.weight.codes.max()) == 128:
        with pytest.raises(OverflowError):
            rec.weight.int8()
    four = copy.deepcopy(rec.weight)
    four.codes = torch.tensor([8, -8, 3], dtype=torch.int32)
    with pytest.raises(OverflowError):
        four.int4_packed()
    four.codes = torch.tensor([0, 15, 8, 7, 1], dtype=torch.int32)          # unsigned nibbles (use_uint under saturation)
    assert same(QuantizedTensor.unpack_int4(four.int4_packed(), 5, signed=False), four.codes)
