"""GPU tensors of dtypes the HIP kernels are not written for (float64, integers).  The reference is dtype-agnostic ATen
(quantize.py:109-117: a float64 input is divided in float64, rounded, and STILL comes out as float32; sparse.py:116, 263: `x * mask`
in x's dtype; util.py:92-99: the staged mean in x's dtype), so "drops into any PyTorch model unchanged" includes `model.double()`.
Such tensors evaluate the package's own ATen expression on the device (`_hip.on_hip`): the code of the CPU path -- which
tests/fuzz/fuzz_reference.py holds against the real reference, float64 included -- never the oracle, never a host round trip.
Element-wise results are bit-identical to the CPU's (IEEE float64 division / rint on both); float64 sums may differ from the
CPU's summation order in the last bit, which the float32 state (magnitude) and the masks of these cases do not see."""
import copy

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from qsparse_amd import _hip

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def gen(s):
    return torch.Generator().manual_seed(s)


def _both(make, xs, grads=True):
    outs = []
    for dev in ("cpu", "cuda"):
        m = make().to(dev).train()
        rec = []
        for x in xs:
            xd = x.detach().clone().to(dev)
            if grads and xd.is_floating_point():
                xd.requires_grad_(True)
            y = m(xd)
            rec.append(y.detach().cpu())
            if grads and xd.requires_grad:
                y.backward(torch.ones_like(y) * 0.37)
                rec.append(xd.grad.cpu())
        rec += [v.detach().cpu() for v in m.state_dict().values()]
        outs.append(rec)
    return outs


@pytest.mark.parametrize("kind", ["scaler", "decimal", "adaptive"])
@pytest.mark.parametrize("channelwise", [-1, 1])
def test_float64_activations_through_every_quantizer(kind, channelwise):
    cb = {"scaler": qs.ScalerQuantizer, "decimal": qs.DecimalQuantizer, "adaptive": qs.AdaptiveQuantizer}[kind]
    n = 1 if (channelwise == 1 and kind != "adaptive") else 6            # (batched channel-wise Scaler / Decimal raises, as the reference)
    xs = [torch.randn(n, 8, 5, 7, generator=gen(s), dtype=torch.float64) * 3 for s in range(5)]
    a, b = _both(lambda: qs.quantize(bits=6, channelwise=channelwise, timeout=1, callback=cb()), xs)
    assert len(a) == len(b)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and same(u, v), i
    # Scaler / Decimal: float32 out of a float64 input (`q.float() * scaler`, quantize.py:117), float64 gradient; the line
    # quantizer keeps the input's dtype (quantize.py:160-181)
    assert a[2].dtype == (torch.float64 if kind == "adaptive" else torch.float32) and a[3].dtype == torch.float64


@pytest.mark.parametrize("dims", [{1}, {0, 1}, {1, 2, 3}])
@pytest.mark.parametrize("policy", [dict(), dict(running_average=False), dict(mask_refresh_interval=2, stop_mask_refresh=4)])
def test_float64_activations_through_the_prune_layer(dims, policy):
    xs = [torch.randn(6, 8, 5, 7, generator=gen(10 + s), dtype=torch.float64) * torch.linspace(0.2, 3, 8, dtype=torch.float64).view(1, -1, 1, 1)
          for s in range(7)]
    a, b = _both(lambda: qs.prune(sparsity=0.5, dimensions=dims, start=1, interval=1, repetition=2,
                                  callback=qs.MagnitudePruningCallback(**policy)), xs)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and same(u, v), i
    assert a[12].dtype == torch.float64 and (a[12] == 0).any()               # the last step's output: float64, pruned


def test_integer_activations():
    xs = [torch.randint(-50, 50, (4, 8, 5, 5), generator=gen(s), dtype=torch.int32) for s in range(4)]
    a, b = _both(lambda: qs.quantize(bits=4, channelwise=-1, timeout=1), xs, grads=False)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and same(u, v), i
    # a prune layer averages |x| (util.py:97): ATen has no integer mean -- the reference raises, on either device, and so does this
    for dev in ("cpu", "cuda"):
        p = qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1).to(dev).train()
        with pytest.raises(RuntimeError, match="mean"):
            p(xs[0].to(dev))
    # ... while a frozen / evaluating one only multiplies (sparse.py:263)
    for dev in ("cpu", "cuda"):
        p = qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1).to(dev).train()
        p(xs[0].float().to(dev))
        p(xs[1].float().to(dev))
        p.eval()
        y = p(xs[2].to(dev))
        assert y.dtype == torch.int32 and torch.equal(y.cpu(), xs[2] * p.mask.cpu())


def test_a_double_precision_network_trains_on_the_gpu():
    """`model.double()` with the operators that keep their input's dtype -- pruned weights and activations (`x * mask`), the line
    quantizer on activations (the Scaler / Decimal quantizers return float32, which a float64 convolution refuses in the
    reference as here): state and effective weights against the CPU run (the convolutions themselves round differently on the
    two devices: the parameters move by a seeded pseudo-update)"""
    def build():
        torch.manual_seed(0)
        net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(), nn.Flatten(),
                            nn.Linear(8 * 36, 5))
        net = qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=1), weight_layers=[nn.Conv2d],
                         activation_layers=[nn.ReLU], log=False)
        net = qs.convert(net, qs.quantize(bits=8, timeout=1, channelwise=-1, callback=qs.AdaptiveQuantizer()), activation_layers=[nn.ReLU],
                         log=False)
        return net.double()

    results = []
    for dev in ("cpu", "cuda"):
        net = build().to(dev).train()
        g = gen(5)
        for s in range(5):
            x = torch.randn(4, 3, 6, 6, generator=g, dtype=torch.float64).to(dev)
            net.zero_grad()
            out = net(x)
            assert out.dtype == torch.float64
            out.sum().backward()
            with torch.no_grad():
                for prm in net.parameters():
                    if prm.requires_grad:
                        prm.add_((torch.randn(prm.shape, generator=g, dtype=torch.float64) * 0.02).to(dev))
        net.eval()
        eff = [m.weight.detach().cpu() for m in net.modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
        results.append((eff, {k: v.detach().cpu() for k, v in net.state_dict().items() if "mask" in k or "_n_updates" in k or k.endswith(".t")}))
    (ea, sa), (eb, sb) = results
    for i, (u, v) in enumerate(zip(ea, eb)):
        assert u.dtype == v.dtype == torch.float64 and same(u, v), ("effective weight", i)
    for k in sa:                    # (the WEIGHT masks depend on the parameters alone; the activations' on convolution outputs, which
        if ".prune." in k:          # the two devices round differently)
            assert same(sa[k], sb[k]), k


def test_supported_dtypes_still_take_the_kernels(monkeypatch):
    calls = []
    real = _hip.quantize_step
    monkeypatch.setattr(_hip, "quantize_step", lambda *a, **k: (calls.append(a[0].dtype), real(*a, **k))[1])
    for dtype in (torch.float32, torch.bfloat16, torch.float16, torch.float64):
        q = qs.quantize(bits=8, channelwise=-1, timeout=1).cuda().train()
        for s in range(3):
            q(torch.randn(4, 8, 4, 4, generator=gen(s)).to(dtype).cuda())
    assert torch.float64 not in calls and {torch.float32, torch.bfloat16, torch.float16} <= set(calls)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.int32])
def test_power_of_two_steps_are_exact_on_the_device(dtype):
    """`2.0 ** decimal` (reference quantize.py:52-53) is exact on the CPU; ATen's device pow is not (powf(2, 16) = 65535.996 on
    gfx950: x = -0.5 with 16 fractional bits came out one code short, found by the float64 cases of the fuzz) -- `quantize._pow2`"""
    for d in range(-20, 31):
        dec = torch.tensor([float(d), float(d + 1)])
        x = (torch.tensor([[-0.5, 3.0], [0.75, -7.0]]) if dtype.is_floating_point else torch.tensor([[-1, 3], [5, -7]])).to(dtype)
        g = torch.tensor([[1e-9, -1e9], [0.5, 2.0 ** (7 - d - 1)]])
        res = []
        for dev in ("cpu", "cuda"):
            xi = x.detach().clone().to(dev).requires_grad_(dtype.is_floating_point)
            y = qs.quantize_with_decimal(xi, 8 if d > -20 else 16, dec.to(dev), channel_index=1)
            res.append(y.detach().cpu())
            if dtype.is_floating_point:
                y.backward(g.to(dev))
                res.append(xi.grad.cpu())
        half = len(res) // 2
        for a, b in zip(res[:half], res[half:]):
            assert torch.equal(a, b), (d, a, b)
