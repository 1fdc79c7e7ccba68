"""Two training forwards before one backward (siamese / contrastive recipes): what each forward's backward clamps with.

The reference's ``ScalerQuantization`` saves the scale PARAMETER itself (quantize.py:108; the statistics update writes it
through ``.data``, :504), so the backward of the FIRST forward clamps with the scale the SECOND forward left behind.  Its
``DecimalQuantizer`` computes a fresh decimal tensor per call (quantize.py:312-325) and ``DecimalQuantization`` saves that one
(:41): the first forward's backward clamps with the first forward's decimal.  The fused sites and the multi-tensor weight
path must do the same; the checker is this package's CPU path (the op-by-op mirror of the reference that the golden
fixtures pin), run on the same inputs."""
import copy
import math

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd.quantize import QuantizeLayer

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def _site(kind, quantizer):
    cb = qs.DecimalQuantizer() if quantizer == "decimal" else qs.ScalerQuantizer()
    q = qs.quantize(bits=4, channelwise=-1, timeout=1, callback=cb)
    net = nn.Sequential(nn.ReLU())
    if kind == "pair":
        net = qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=1, dimensions={1}),
                         activation_layers=[nn.ReLU], log=False)
    return qs.convert(net, q, activation_layers=[nn.ReLU], log=False)


@pytest.mark.parametrize("quantizer", ["scaler", "decimal"])
@pytest.mark.parametrize("kind", ["relu_q", "pair"])
def test_backward_of_the_first_of_two_forwards(kind, quantizer):
    torch.manual_seed(0)
    cpu = _site(kind, quantizer).train()
    gpu = copy.deepcopy(cpu).cuda()
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, 8, 6, 6, generator=g) * s for s in (1.0, 1.0, 1.0, 0.05, 9.0)]   # the last two are octaves apart
    outs = {}
    for name, net, dev in (("cpu", cpu, "cpu"), ("gpu", gpu, "cuda")):
        for x in xs[:3]:                                  # identity step, then live steps
            net(x.clone().to(dev).requires_grad_()).sum().backward()
        xa, xb = xs[3].clone().to(dev).requires_grad_(), xs[4].clone().to(dev).requires_grad_()
        q = [m for m in net.modules() if isinstance(m, QuantizeLayer)][0]
        ya = net(xa)
        scale_a = float(q.weight)
        yb = net(xb)                                      # moves scale / decimal / mask before ya's backward runs
        scale_b = float(q.weight)
        ga = torch.randn(ya.shape, generator=torch.Generator().manual_seed(2)) * 6.0
        gb = torch.randn(yb.shape, generator=torch.Generator().manual_seed(3)) * 6.0
        ya.backward(ga.to(dev).clone())
        yb.backward(gb.to(dev).clone())
        outs[name] = [t.detach().cpu() for t in (ya, yb, xa.grad, xb.grad)]
        outs[name + "_state"] = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    assert outs["cpu_state"].keys() == outs["gpu_state"].keys()
    for k, v in outs["cpu_state"].items():
        assert torch.equal(v, outs["gpu_state"][k]), k
    for i, (a, b) in enumerate(zip(outs["cpu"], outs["gpu"])):
        assert a.shape == b.shape and torch.equal(a, b), i
    # the first forward's backward saturated at the bound its quantizer's Function saved: the live scale for Scaler (the
    # second forward's), the first forward's own decimal for Decimal -- octaves apart here
    assert scale_b > 3 * scale_a
    bound = 8 * scale_b if quantizer == "scaler" else 8 * 2.0 ** -round(math.log2(1 / scale_a))
    gxa = outs["gpu"][2]
    assert float(gxa.abs().max()) == pytest.approx(bound, rel=1e-6) and int((gxa.abs() == gxa.abs().max()).sum()) > 3


@pytest.mark.parametrize("quantizer", ["scaler", "decimal"])
def test_weight_path_backward_of_the_first_of_two_forwards(quantizer):
    """the multi-tensor weight path against the layer-by-layer one, bit for bit"""
    res = []
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            torch.manual_seed(0)
            cb = qs.DecimalQuantizer() if quantizer == "decimal" else None
            net = nn.Sequential(nn.Linear(16, 32), nn.Tanh(), nn.Linear(32, 8))
            net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1, callback=cb), weight_layers=[nn.Linear],
                             log=False).cuda().train()
            assert (net.__dict__.get("_qs_weight_batcher") is not None) == batched
            g = torch.Generator().manual_seed(4)
            for _ in range(3):
                net(torch.randn(5, 16, generator=g).cuda()).sum().backward()
            net.zero_grad()
            ya = net(torch.randn(5, 16, generator=g).cuda())
            scale_a = float(net[0].quantize.weight)
            with torch.no_grad():                         # the weights grow by octaves between the two forwards
                for prm in net.parameters():
                    if prm.requires_grad:
                        prm.mul_(6.0)
            yb = net(torch.randn(5, 16, generator=g).cuda())
            scale_b = float(net[0].quantize.weight)
            (ya * torch.randn(ya.shape, generator=g).cuda()).sum().backward()
            first = [p.grad.detach().clone() for p in net.parameters() if p.requires_grad]
            (yb * torch.randn(yb.shape, generator=g).cuda()).sum().backward()
            both = [p.grad.detach().clone() for p in net.parameters() if p.requires_grad]
            res.append((ya.detach(), yb.detach(), first, both, {k: v.detach().clone() for k, v in net.state_dict().items()}))
        finally:
            qs.set_qsparse_options(batch_weights=True)
    (ya0, yb0, f0, b0, s0), (ya1, yb1, f1, b1, s1) = res
    assert torch.equal(ya0, ya1) and torch.equal(yb0, yb1)
    for a, b in zip(f0 + b0, f1 + b1):
        assert torch.equal(a, b)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    # the first layer's weight gradient of the first forward saturates at the bound the reference's Function saved: the live
    # (second forward's) scale for Scaler, the first forward's own decimal for Decimal
    assert scale_b > 2 * scale_a
    bound = 8 * scale_b if quantizer == "scaler" else 8 * 2.0 ** -round(math.log2(1 / scale_a))
    assert float(f0[0].abs().max()) == pytest.approx(bound, rel=1e-6) and int((f0[0].abs() == f0[0].abs().max()).sum()) > 8
