"""Row f4 of SURVEY section 8: the inference-side step -- integer export (codes + scales / decimals / zero points,
masks) through a PUBLIC entry point, and `fuse_bn` on the device.

The reference has no export API; it pins the property an export must have (tests/test_quantize.py:73-101: the
float-simulated 8-bit convolution equals int32 arithmetic on the codes, built on qsparse/quantize.py:44-63).  Here the
codes are the kernels' own `codes` output (`quantize_with_*(..., return_codes=True)`, `qs.export_integer(model)`), checked
against the oracle's `scaler_codes` / `decimal_codes` and through that int32-convolution property.  Every test body runs
on the CPU path (`-m "not gpu"`) and on the HIP path (`-m gpu`).
"""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from golden_io import Golden, same
from oracle import qs_oracle as O
from qsparse_amd.quantize import (AdaptiveQuantizer, DecimalQuantizer, ScalerQuantizer, quantize_with_decimal,
                                  quantize_with_line, quantize_with_scaler)

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dev", DEVICES)
def test_return_codes_of_the_functional_api_equal_the_oracle(dev):
    for dtype in (torch.float32, torch.bfloat16):
        x = (torch.randn(5, 12, 7, 6, generator=gen(1)) * 3).to(dtype)
        x.view(-1)[:6] = torch.tensor([0.05, -0.05, 0.15, -0.15, 0.25, 1e6]).to(dtype)       # half-way quotients, a huge one
        sc = torch.linspace(0.05, 0.4, 12).view(-1, 1)
        for scale, ci in ((torch.tensor([[0.1]]), -1), (sc, 1)):
            y, q = quantize_with_scaler(x.to(dev), 4, scale.to(dev), ci, return_codes=True)
            assert q.dtype == torch.int32 and same(q.cpu(), O.scaler_codes(x, scale, ci))
            assert same(y.cpu(), O.scaler_fwd(x, 4, scale, ci))
        dec = torch.tensor([3.0, 4, 5, 2, 1, 0, 6, 7, 3, 3, 2, 4])
        for d, ci in ((5, -1), (dec, 1)):
            y, q = quantize_with_decimal(x.to(dev), 8, d.to(dev) if isinstance(d, torch.Tensor) else d, ci, return_codes=True)
            assert q.dtype == torch.int32 and same(q.cpu(), O.decimal_codes(x, d, ci)) and same(y.cpu(), O.decimal_fwd(x, 8, d, ci))
    # line quantizer: level index in [0, N - 1]; values rebuilt from the integers alone, both zero-point forms
    x = torch.randn(4, 6, 5, 5, generator=gen(2)) * 2
    lines = torch.stack([torch.linspace(-2, -0.5, 6), torch.linspace(0.7, 2.5, 6)], dim=1)
    step = ((lines[:, 1] - lines[:, 0]) / 16).view(1, -1, 1, 1)
    for fzp in (True, False):
        y, idx = quantize_with_line(x.to(dev), 4, lines.to(dev), 1, float_zero_point=fzp, return_codes=True)
        assert idx.dtype == torch.int32 and int(idx.min()) >= 0 and int(idx.max()) <= 15
        assert same(y.cpu(), O.line_fwd(x, 4, lines, 1, float_zero_point=fzp))
        i = idx.cpu().float()
        rebuilt = i * step + lines[:, 0].view(1, -1, 1, 1) if fzp else (i + (lines[:, 0].view(1, -1, 1, 1) / step).round()) * step
        assert same(rebuilt, y.cpu())
    # gradients still flow through the first output only
    xg = x.clone().to(dev).requires_grad_(True)
    y, q = quantize_with_scaler(xg, 8, 0.05, return_codes=True)
    assert not q.requires_grad
    y.backward(torch.ones_like(y))
    assert xg.grad is not None and float(xg.grad.abs().sum()) > 0


def _train(layer, x, steps):
    layer.train()
    for _ in range(steps):
        layer(x)
    return layer.eval()


@pytest.mark.parametrize("dev", DEVICES)
def test_export_integer_scaler_weights_with_pruning_and_bias(dev):
    torch.manual_seed(3)
    conv = nn.Conv2d(6, 10, 3)
    layer = qs.quantize(qs.prune(conv, sparsity=0.5, dimensions={0, 1, 2, 3}, start=1, interval=1, repetition=1),
                        bits=8, channelwise=0, timeout=2, bias_bits=12, callback=ScalerQuantizer()).to(dev)
    x = torch.rand(2, 6, 9, 9, generator=gen(4)).to(dev)
    _train(layer, x, 5)
    before = copy.deepcopy(layer.state_dict())
    ex = qs.export_integer(layer)
    assert list(ex) == [""] and ex[""].module == "Conv2d"
    w, b = ex[""].weight, ex[""].bias
    raw, mask = layer._parameters["weight"].detach().cpu(), layer.prune.mask.detach().cpu()
    scale = layer.quantize.weight.detach().cpu()
    assert w.kind == "scaler" and w.bits == 8 and w.channel_index == 0 and same(w.mask.cpu(), mask)
    assert w.codes.dtype == torch.int32 and same(w.codes.cpu(), O.scaler_codes(raw * mask, scale, 0))
    assert same(w.scale.cpu(), scale) and abs(float((~mask).float().mean()) - 0.5) < 0.01
    assert same(w.dequantize().cpu(), layer.weight.detach().cpu())            # the integers alone rebuild the effective weight
    # quirk B1 (no forward saturation): the largest |element| of a row maps to code +-2^(bits-1); +128 does not fit int8
    if int(w.codes.max()) > 127:
        with pytest.raises(OverflowError):
            w.int8()
    else:
        assert same(w.int8().cpu(), w.codes.cpu().to(torch.int8))
    assert b.kind == "scaler" and b.bits == 12
    assert same(b.codes.cpu().view(-1), O.scaler_codes(layer._parameters["bias"].detach().cpu(), layer.quantize_bias.weight.detach().cpu(), 0).view(-1))
    assert same(b.dequantize().cpu().view(-1), layer.bias.detach().cpu().view(-1))
    # exporting touches no state and leaves the mode alone
    after = layer.state_dict()
    assert all(same(before[k].cpu(), after[k].cpu()) for k in before) and not layer.training
    layer.train()
    qs.export_integer(layer)
    assert layer.training and layer.quantize.training and layer.prune.training
    assert all(same(before[k].cpu(), layer.state_dict()[k].cpu()) for k in before)


@pytest.mark.parametrize("dev", DEVICES)
def test_int32_convolution_on_exported_codes_equals_the_float_simulation(dev):
    """the reference's own property (tests/test_quantize.py:73-101), with the integers taken from `export_integer`"""
    ni, no, timeout = 7, 6, 5
    inp = torch.randint(-128, 127, size=(3, 10, 16, 16), generator=gen(0))
    inp_float = (inp.float() / 2 ** ni).to(dev)
    torch.manual_seed(0)
    qconv = qs.quantize(nn.Conv2d(10, 30, 3, bias=False), bits=8, timeout=timeout, channelwise=0,
                        callback=DecimalQuantizer()).to(dev)
    qconv.train()
    for _ in range(timeout + 1):
        qconv(inp_float)
    out_float, out_codes = quantize_with_decimal(qconv(inp_float), 8, no, return_codes=True)
    e = qs.export_integer(qconv)[""].weight
    assert e.kind == "decimal" and e.decimal.dtype == torch.int32 and e.decimal.numel() == 30
    w_int, decimal = e.codes.cpu(), e.decimal.cpu().view(-1)
    assert same(w_int, O.decimal_codes(qconv._parameters["weight"].detach().cpu(), decimal.float().view(-1, 1), 0))
    assert same(e.dequantize().cpu(), qconv.weight.detach().cpu())
    out_int = F.conv2d(inp.int(), w_int)
    for i in range(out_int.shape[1]):
        out_int[:, i] = (out_int[:, i].float() / 2 ** (ni + decimal[i] - no)).int()
    assert torch.equal(out_codes.cpu(), out_int)                      # the integers themselves ...
    assert torch.equal(out_float.detach().cpu(), out_int.float() / 2 ** no)        # ... and the reference's float criterion


@pytest.mark.parametrize("dev", DEVICES)
def test_export_integer_adaptive_and_groupwise(dev):
    torch.manual_seed(5)
    lin = qs.quantize(nn.Linear(24, 16), bits=6, channelwise=0, timeout=1, callback=AdaptiveQuantizer()).to(dev)
    x = torch.rand(4, 24, generator=gen(6)).to(dev)
    _train(lin, x, 4)
    e = qs.export_integer(lin)[""].weight
    assert e.kind == "line" and e.zero_point.dtype == torch.int32 and e.step.numel() == 16
    assert int(e.codes.min()) >= 0 and int(e.codes.max()) <= 63
    assert same(e.values.cpu(), lin.weight.detach().cpu()) and same(e.dequantize().cpu(), lin.weight.detach().cpu())
    assert same(e.values.cpu(), O.line_fwd(lin._parameters["weight"].detach().cpu(), 6, lin.quantize.weight.detach().cpu(), 0, False))
    # group-wise scales (sklearn clustering on the host, reference quantize.py:352-366): the export shows the SHARED scales
    torch.manual_seed(7)
    conv = qs.quantize(nn.Conv2d(4, 12, 3), bits=8, channelwise=0, timeout=1,
                       callback=DecimalQuantizer(group_num=3, group_timeout=2)).to(dev)
    with torch.no_grad():
        conv._parameters["weight"] *= torch.logspace(-2, 1, 12).view(-1, 1, 1, 1).to(dev)
    _train(conv, torch.rand(2, 4, 8, 8, generator=gen(8)).to(dev), 5)
    e = qs.export_integer(conv)[""].weight
    assert e.kind == "decimal" and len(set(e.decimal.view(-1).tolist())) <= 3
    assert same(e.dequantize().cpu(), conv.weight.detach().cpu())


@pytest.mark.parametrize("dev", DEVICES)
def test_export_integer_of_a_converted_network(dev):
    from examples.models import convert_pq, resnet18
    torch.manual_seed(0)
    model = convert_pq(resnet18(10, True, 8), sparsity=0.5, bits=4, prune_start=1, prune_interval=1, repetition=1,
                       quant_timeout=1).to(dev).train()
    x = torch.randn(4, 3, 32, 32, generator=gen(9)).to(dev)
    for _ in range(4):
        model(x).sum().backward()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ex = qs.export_integer(model)
    weights = {k: v for k, v in ex.items() if v.weight is not None}
    acts = {k: v for k, v in ex.items() if v.activation is not None}
    n_layers = sum(1 for m in model.modules() if isinstance(m, (nn.Conv2d, nn.Linear)))
    assert len(weights) == n_layers and model.training
    modules = dict(model.named_modules())
    for path, rec in weights.items():
        layer = modules[path]
        layer.eval()
        assert same(rec.weight.dequantize().cpu(), layer.weight.detach().cpu()), path
        assert int(rec.weight.codes.abs().max()) <= 8                 # 4 bits, no saturation: the largest element maps to +8
        layer.train()
    kinds = {v.activation["operator"] for v in acts.values()}
    assert kinds == {"quantize", "prune"}
    masks = [v.activation["mask"] for v in acts.values() if v.activation["operator"] == "prune"]
    assert masks and all(m.dtype == torch.bool and 0.3 <= float(m.float().mean()) <= 0.75 for m in masks)
    assert all(same(sd[k].cpu(), v.cpu()) for k, v in model.state_dict().items())      # nothing moved


def test_export_skips_operators_that_never_quantized():
    conv = qs.quantize(nn.Conv2d(3, 4, 3), bits=8, channelwise=-1, timeout=100)
    conv.train()
    conv(torch.rand(1, 3, 8, 8))
    assert qs.export_integer(conv) == {}
    assert qs.export_integer(nn.Sequential(nn.Conv2d(3, 4, 3), nn.ReLU())) == {}


# ---- fuse_bn on the device (reference qsparse/fuse.py:76-163, tests/test_fuse.py:7-130) -----------------------------------
def _bn_nets():
    return {
        "conv": nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5)),
        "linear": nn.Sequential(nn.Linear(12, 7, bias=False), nn.BatchNorm1d(7)),
        "deconv": nn.Sequential(nn.ConvTranspose2d(3, 5, 3), nn.BatchNorm2d(5)),
        "nested": nn.Sequential(nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)), nn.ReLU(),
                                nn.Sequential(nn.Conv2d(4, 4, 3)), nn.BatchNorm2d(4), nn.ReLU(),
                                nn.Sequential(nn.BatchNorm2d(4), nn.ConvTranspose2d(4, 2, 3), nn.BatchNorm2d(2))),
    }


@pytest.mark.gpu
def test_f11_fuse_bn_on_the_gpu():
    """the golden networks moved to the GPU before folding: same tree and keys; folded parameters equal the reference's
    within one rounding of the device's sqrt / divide (the algebra is C-sized ATen arithmetic on whatever device holds the
    parameters); fused evaluation == unfused evaluation within the reference's own 1e-5"""
    g = Golden("f11_fuse_bn")
    nets = _bn_nets()
    for c in g.cases:
        name, net = c["name"], nets[c["name"]]
        net.load_state_dict({k: torch.as_tensor(g.get(f"{name}_in_{k}")) for k in net.state_dict()})
        net = net.cuda().eval()
        x = g.get(name + "_x").cuda()
        before = net(x)
        fused = qs.fuse_bn(net, log=False)
        assert str(fused) == c["tree"], name            # (the nested case keeps the BatchNorm that PRECEDES its layer, as the reference does)
        sd = fused.state_dict()
        assert list(sd.keys()) == c["out_keys"], name
        for k, v in sd.items():
            want = torch.as_tensor(g.get(f"{name}_out_{k}"))
            assert v.is_cuda and torch.allclose(v.cpu(), want, rtol=3e-7, atol=1e-9), (name, k)
        y = fused(x)
        assert torch.allclose(y, before, atol=1e-5)
        assert torch.allclose(y.cpu(), g.get(name + "_y"), atol=1e-5)


@pytest.mark.gpu
def test_fuse_bn_then_quantize_on_the_gpu_equals_the_cpu_run():
    """the inference recipe end to end on the device: train conv+BN, fold, wrap the folded layer in a weight quantizer,
    export -- against the same recipe on the CPU path"""
    outs = {}
    for dev in ("cpu", "cuda"):
        torch.manual_seed(11)
        net = nn.Sequential(nn.Conv2d(4, 8, 3), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 8, 3), nn.BatchNorm2d(8))
        x = torch.randn(6, 4, 12, 12, generator=gen(12))
        net.train()
        net(x)                                   # running statistics on the host: identical starting point for both devices
        net = net.to(dev).eval()
        gt = net(x.to(dev))
        fused = qs.fuse_bn(net, log=False)
        assert "batchnorm" not in str(fused).lower() and torch.allclose(fused(x.to(dev)), gt, atol=1e-5)
        q = qs.convert(fused, qs.quantize(bits=8, channelwise=0, timeout=1, callback=DecimalQuantizer()),
                       weight_layers=[nn.Conv2d], log=False)
        q.train()
        for _ in range(3):
            q(x.to(dev))
        q.eval()
        outs[dev] = (q(x.to(dev)).detach().cpu(), {k: (v.weight.codes.cpu(), v.weight.decimal.cpu()) for k, v in qs.export_integer(q).items()})
    assert outs["cpu"][1].keys() == outs["cuda"][1].keys() and len(outs["cpu"][1]) == 2
    for k in outs["cpu"][1]:
        # the folded weights may differ by an ulp between devices (sqrt / divide), codes then differ in at most a few entries
        dc = (outs["cpu"][1][k][0] != outs["cuda"][1][k][0]).float().mean().item()
        assert dc <= 0.01 and torch.equal(outs["cpu"][1][k][1], outs["cuda"][1][k][1]), (k, dc)
    assert torch.allclose(outs["cpu"][0], outs["cuda"][0], atol=2e-2)
