"""The float-simulated 8-bit convolution equals true int32 arithmetic -- the property the reference pins in
tests/test_quantize.py:73-101 (it relies on DecimalQuantization truncating and on power-of-two scales).
Here the integer codes come straight from the kernels' `codes` output on the GPU variant."""
import pytest
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from qsparse_amd.quantize import DecimalQuantizer, quantize_with_decimal

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def _run(dev):
    ni, no, timeout = 7, 6, 5
    g = torch.Generator().manual_seed(0)
    inp = torch.randint(-128, 127, size=(3, 10, 16, 16), generator=g)
    inp_float = (inp.float() / 2 ** ni).to(dev)
    torch.manual_seed(0)
    qconv = qs.quantize(torch.nn.Conv2d(10, 30, 3, bias=False), bits=8, timeout=timeout, channelwise=0,
                        callback=DecimalQuantizer()).to(dev)
    qconv.train()
    for _ in range(timeout + 1):
        qconv(inp_float)
    out_float = quantize_with_decimal(qconv(inp_float), 8, no).detach().cpu()

    decimal = (1 / qconv.quantize.weight).nan_to_num(posinf=1, neginf=1).log2().round().int().cpu()
    w_int = (qconv.weight.detach().cpu() * (2.0 ** decimal).view(-1, 1, 1, 1)).int()
    if dev != "cpu":
        from qsparse_amd import _hip
        # the kernel's own integer codes of the raw weight are the same integers
        _, codes = _hip.quant_fwd("decimal", qconv._parameters["weight"].detach(), decimal.float().to(dev), 0,
                                  torch.float32, want_codes=True)
        assert torch.equal(codes.cpu(), w_int)
    out_int = F.conv2d(inp.int(), w_int)
    for i in range(out_int.shape[1]):
        out_int[:, i] = (out_int[:, i].float() / 2 ** (ni + decimal[i] - no)).int()
    return out_float, out_int.float() / 2 ** no


def test_float_simulation_equals_integer_arithmetic_cpu():
    a, b = _run("cpu")
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_float_simulation_equals_integer_arithmetic_gpu():
    a, b = _run("cuda")
    # the GPU convolution accumulates in a different order, but every product and partial sum here is an exact
    # dyadic rational below 2^24, so the float convolution is exact and the property still holds bit for bit
    assert torch.equal(a, b)
