"""Folding the rectifier-type activations other than nn.ReLU.

`convert` puts its operators behind whatever activation modules the caller names (reference qsparse/convert.py:214-218):
nn.ReLU in the BASELINE networks, nn.ReLU6 in MobileNet-style ones, nn.Hardtanh / nn.LeakyReLU elsewhere.  The fused sites absorb
all four: the activation's output is never written, statistics and forward read its input (`pre_relu` = a `qs_activation`
handle), the forward records the one bit per element its backward needs, the backward applies it -- `hardtanh_backward`: 0
outside (a, b); `leaky_relu_backward`: the gradient times the slope where x <= 0.  Everything is compared bit for bit

  * with the CPU path of the same modules (ATen's own activation, `x * mask`, the quantizer: the reference's operator chain), and
  * on a MobileNet-style inverted-residual network with the oracle's state machines fed the tensors each site really received.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from oracle import qs_oracle as O
from qsparse_amd import _hip
from qsparse_amd.quantize import DecimalQuantizer, QuantizeLayer, ScalerQuantizer

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)

ACTS = {"relu6": lambda inplace: nn.ReLU6(inplace=inplace), "hardtanh": lambda inplace: nn.Hardtanh(-0.75, 1.5, inplace=inplace),
        "hardtanh_odd": lambda inplace: nn.Hardtanh(0.1, 0.7, inplace=inplace),     # bounds bf16 / fp16 cannot represent
        "leaky": lambda inplace: nn.LeakyReLU(0.1, inplace=inplace), "relu": lambda inplace: nn.ReLU(inplace=inplace)}


def gen(seed):
    return torch.Generator().manual_seed(seed)


def same(a, b):
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.view(torch.uint8).view(-1), b.view(torch.uint8).view(-1))


def _site(kind, act, inplace, quantizer=ScalerQuantizer):
    net = nn.Sequential(ACTS[act](inplace))
    types = [type(net[0])]
    if kind in ("pair", "act_p"):
        net = qs.convert(net, qs.prune(sparsity=0.5, start=2, interval=1, repetition=2, dimensions={1}), activation_layers=types, log=False)
    if kind in ("pair", "act_q"):
        net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=2, callback=quantizer()), activation_layers=types, log=False)
    return net


def _run(net, dev, data, fmt, inplace):
    net = net.to(dev).train()
    outs = []
    for step, (x, g) in enumerate(data):
        if step == len(data) - 1:
            net.eval()
        x0 = x.detach().clone().to(dev).contiguous(memory_format=fmt).requires_grad_(True)
        # (an in-place activation needs a non-leaf input, as behind a convolution.  A clone, not `x0 * 1.0`: ATen's own fp16 GPU
        # multiply is compiled to v_fma_mixlo_f16(g, 1.0, +0) and returns +0.0 for a -0.0 gradient -- the CPU's keeps the sign)
        h = x0.clone() if inplace else x0
        y = net(h)
        y.backward(g.to(dev).to(y.dtype))
        outs.append((y.detach().cpu(), x0.grad.detach().cpu(), h.detach().cpu()))
    return outs, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("fmt", [torch.contiguous_format, torch.channels_last])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("act", ["relu6", "hardtanh", "hardtanh_odd", "leaky"])
@pytest.mark.parametrize("kind", ["pair", "act_q", "act_p"])
def test_folded_activation_sites_equal_the_cpu_path(kind, act, inplace, dtype, fmt):
    if dtype == torch.float16 and act != "hardtanh_odd":
        pytest.skip("fp16 runs the clamp whose bounds it cannot represent; ATen's fp16 leaky_relu differs between its own CPU and GPU kernels")
    g = gen(11)
    data = []
    for step in range(8):
        x = (torch.randn(6, 16, 9, 8, generator=g) * torch.linspace(0.4, 3.0, 16).view(1, -1, 1, 1)).to(dtype)
        if step >= 2:     # (the inactive steps 0 and 1 are ATen's own device kernels, whose -0.0 differs from the CPU's)
            x.view(-1)[5:13] = torch.tensor([-0.0, 0.0, 6.0, 1.5, -0.75, 7.0, -3.0, 1e-30]).to(dtype)     # the gates' boundary values
            x.view(-1)[13:19] = torch.tensor([0.1, 0.7, 0.69921875, 0.10009765625, 0.099609375, 0.703125]).to(dtype)
        data.append((x, torch.randn(6, 16, 9, 8, generator=g)))
    cpu, cpu_state = _run(_site(kind, act, inplace), "cpu", data, fmt, inplace)
    # the site IS folded: the activation module's forward is not called on the GPU once the operators are active (counted
    # through the instance's `forward`: a hook on a child would make the site fall back to module by module)
    calls, real = [], {}
    net_gpu = _site(kind, act, inplace).to("cuda")
    for m in net_gpu.modules():
        if type(m) in (nn.ReLU6, nn.Hardtanh, nn.LeakyReLU):
            real[m] = m.forward
            m.forward = (lambda mod: (lambda x: (calls.append(1), real[mod](x))[1]))(m)
    gpu, gpu_state = _run(net_gpu, "cuda", data, fmt, inplace)
    for step, ((ya, ga, ha), (yb, gb, hb)) in enumerate(zip(cpu, gpu)):
        assert same(ya, yb), ("output", step)
        assert same(ga, gb), ("input gradient", step)
        if inplace and kind != "act_p":
            assert same(ha, hb), ("the modified tensor", step)
        elif inplace:     # the prune-only site has no write-back kernel: ATen's own in-place pass on the device, whose clamp
            assert torch.equal(ha.float(), hb.float()), ("the modified tensor", step)      # turns -0.0 into +0.0 (the CPU's keeps it)
    assert cpu_state.keys() == gpu_state.keys()
    for k in cpu_state:
        assert same(cpu_state[k], gpu_state[k]), k
    # steps 0 and 1: the operators are inactive (module by module: 2 calls); afterwards everything is folded -- except an
    # IN-PLACE clamp whose bounds the dtype cannot represent: its saturated values (0.10009765625, 0.69921875 in bf16) lie inside
    # (0.1, 0.7), so the gate cannot be read off the rectified tensor and the module keeps running by itself (fused._foldable_relu)
    unfolded = act == "hardtanh_odd" and inplace and dtype != torch.float32
    assert len(calls) == (8 if unfolded else 2), calls


@pytest.mark.parametrize("quantizer", [DecimalQuantizer])
@pytest.mark.parametrize("act", ["relu6", "leaky"])
def test_folded_activation_with_the_decimal_quantizer(act, quantizer):
    g = gen(12)
    data = [((torch.randn(4, 8, 6, 6, generator=g) * 3).bfloat16(), torch.randn(4, 8, 6, 6, generator=g)) for _ in range(6)]
    cpu, cpu_state = _run(_site("pair", act, False, quantizer), "cpu", data, torch.contiguous_format, False)
    gpu, gpu_state = _run(_site("pair", act, False, quantizer), "cuda", data, torch.contiguous_format, False)
    for (ya, ga, _), (yb, gb, _) in zip(cpu, gpu):
        assert same(ya, yb) and same(ga, gb)
    for k in cpu_state:
        assert same(cpu_state[k], gpu_state[k]), k


def test_activation_handles_are_interned_descriptors():
    a = _hip.activation(_hip.ACT_HARDTANH, 0.0, 6.0)
    assert a >= 2 and a == _hip.activation(_hip.ACT_HARDTANH, 0.0, 6.0) and a != _hip.activation(_hip.ACT_HARDTANH, -1.0, 1.0)
    assert _hip.activation(_hip.ACT_LEAKY, 0.1) != a and _hip.act_spec(a) == (_hip.ACT_HARDTANH, 0.0, 6.0)
    lib = _hip.load()
    assert lib.qs_activation(7, 0.0, 0.0) < 0 and lib.qs_activation(_hip.ACT_HARDTANH, 2.0, 1.0) < 0      # unknown kind; a > b
    assert lib.qs_activation(_hip.ACT_RELU, 0.0, 0.0) == 1 and lib.qs_activation(0, 0.0, 0.0) == 0
    x = torch.randn(64, device="cuda")
    with pytest.raises(_hip.QsparseHipError):
        _hip.absmax(x, -1, pre_relu=200)          # a handle nobody made


class InvertedResidual(nn.Module):
    """MobileNetV2's block: 1x1 expand, 3x3 depthwise, 1x1 project, ReLU6 twice (out of place here: the oracle replay hangs
    full backward hooks on the sites, which PyTorch does not allow around an in-place module; the in-place form is the
    subject of test_folded_activation_sites_equal_the_cpu_path)"""

    def __init__(self, cin, cout, stride, expand):
        super().__init__()
        hidden = cin * expand
        self.use_res = stride == 1 and cin == cout
        self.conv = nn.Sequential(
            nn.Conv2d(cin, hidden, 1, bias=False), nn.BatchNorm2d(hidden), nn.ReLU6(),
            nn.Conv2d(hidden, hidden, 3, stride, 1, groups=hidden, bias=False), nn.BatchNorm2d(hidden), nn.ReLU6(),
            nn.Conv2d(hidden, cout, 1, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        return x + self.conv(x) if self.use_res else self.conv(x)


class TinyMobileNet(nn.Module):
    def __init__(self, classes=10):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.ReLU6())
        self.blocks = nn.Sequential(InvertedResidual(16, 16, 1, 2), InvertedResidual(16, 24, 2, 4), InvertedResidual(24, 24, 1, 4),
                                    InvertedResidual(24, 32, 2, 4))
        self.head = nn.Sequential(nn.Conv2d(32, 64, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU6())
        self.fc = nn.Linear(64, classes)

    def forward(self, x):
        return self.fc(self.head(self.blocks(self.stem(x))).mean((2, 3)))


class _Site:
    """oracle twin of one ReLU6 -> prune -> quantize site (or ReLU6 -> quantize): what the site's forward / backward hooks check"""

    def __init__(self, name, module, log):
        self.name, self.log = name, log
        first, self.q = module[0], module[1]
        self.p = first[1] if isinstance(first, nn.Sequential) else None
        self.psim = O.PruneSim(0.5, [1], 2, 2, 2, False) if self.p is not None else None
        self.qsim = O.QuantizeSim("scaler", 4, -1, 3)
        self.saved, self.steps = None, 0

    def fwd(self, module, inputs, output):
        if not module.training:
            return
        x = inputs[0].detach().cpu()                  # AFTER the forward: an in-place activation has rectified it -- idempotent
        a = F.hardtanh(x, 0.0, 6.0)
        n_before = self.psim.n_updates if self.psim else 0
        h = self.psim.step(a, True) if self.psim else a
        y_ref = self.qsim.step(h.contiguous(), True)
        tag = (self.name, self.steps)
        assert same(output.detach().cpu().contiguous(), y_ref.contiguous()), ("output", tag)
        if self.psim:
            assert same(self.p.mask, self.psim.mask), ("mask", tag)
            if self.psim.magnitude is not None:
                assert same(self.p.callback.magnitude, self.psim.magnitude), ("magnitude", tag)
        assert same(self.q.weight, self.qsim.weight), ("scale", tag)
        self.saved = (x, n_before)
        self.steps += 1

    def bwd(self, module, grad_input, grad_output):
        if self.saved is None or grad_input[0] is None:
            return
        x, n_before = self.saved
        g = grad_output[0].detach().cpu().contiguous()
        gin = self.qsim.grad(g, x.dtype)
        if self.psim:
            gin = self.psim.grad(gin, n_before >= self.psim.start)
        gin = torch.where((x > 0) & (x < 6), gin, torch.zeros_like(gin))      # hardtanh_backward (h in (0, 6) <=> x in (0, 6))
        assert same(grad_input[0].detach().cpu().contiguous(), gin.contiguous()), ("input gradient", self.name, self.steps - 1)
        self.log.append((self.name, self.steps - 1, float((gin == 0).float().mean())))


@pytest.mark.parametrize("channels_last", [False, True])
def test_mobilenet_style_network_sites_vs_oracle(channels_last):
    torch.manual_seed(0)
    net = TinyMobileNet()
    net = qs.convert(net, qs.prune(sparsity=0.5, dimensions={1}, start=2, interval=2, repetition=2), activation_layers=[nn.ReLU6],
                     excluded_activation_layer_indexes=[(nn.ReLU6, [-1])], log=False)
    net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=3), activation_layers=[nn.ReLU6],
                     weight_layers=[nn.Conv2d, nn.Linear], log=False).cuda().train()
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    log, sites = [], []
    for name, m in net.named_modules():
        if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[1], QuantizeLayer) and not isinstance(m[0], QuantizeLayer):
            s = _Site(name, m, log)
            m.register_forward_hook(s.fwd)
            m.register_full_backward_hook(s.bwd)
            sites.append(s)
    assert len(sites) == 10 and sum(s.psim is not None for s in sites) == 9
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    g = gen(3)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for step in range(8):
            x = torch.randn(16, 3, 32, 32, generator=g).cuda()
            if channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            y = torch.randint(0, 10, (16,), generator=g).cuda()
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(net(x).float(), y)
            loss.backward()
            opt.step()
            assert torch.isfinite(loss).item()
    finally:
        torch.set_num_threads(threads)
    assert all(s.steps == 8 for s in sites) and len(log) >= 8 * 9
    kept = [float(s.p.mask.float().mean()) for s in sites if s.p is not None]
    assert all(0.4 <= k <= 0.8 for k in kept), kept


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["relu", "relu6", "leaky"])
@pytest.mark.parametrize("policy", ["no_avg", "uniform", "dims01"])
def test_inplace_activation_in_front_of_a_site_that_materialises_it_after_all(policy, act, dtype):
    """An owned in-place activation is deferred to the site's kernels (`fused.py::_with_owned_relu`).  A prune operator whose
    policy the kernels do not cover -- no running average, the uniform policy, masks over two dims -- applies the activation
    through ATen instead, on the alias: the deferred in-place pass must happen BEFORE that node records its input (found by
    tests/fuzz/fuzz_cpu_gpu.py: "modified by an inplace operation" in the backward).  Compared with the CPU path."""
    def build():
        cb = {"no_avg": lambda: qs.MagnitudePruningCallback(running_average=False),
              "uniform": lambda: qs.UniformPruningCallback(),
              "dims01": lambda: qs.MagnitudePruningCallback()}[policy]()
        net = nn.Sequential(ACTS[act](True))
        return qs.convert(net, qs.prune(sparsity=0.4, start=1, interval=1, repetition=2, dimensions={0, 1} if policy == "dims01" else {1},
                                        callback=cb), activation_layers=[type(net[0])], log=False)
    import numpy as np
    g = gen(21)
    data = [((torch.randn(4, 8, 6, 6, generator=g) * 3).to(dtype), torch.randn(4, 8, 6, 6, generator=g)) for _ in range(6)]
    np.random.seed(5)
    cpu, cpu_state = _run(build(), "cpu", data, torch.contiguous_format, True)
    np.random.seed(5)
    gpu, gpu_state = _run(build(), "cuda", data, torch.contiguous_format, True)
    for step, ((ya, ga, ha), (yb, gb, hb)) in enumerate(zip(cpu, gpu)):
        assert torch.equal(ya.float(), yb.float()) and ya.dtype == yb.dtype, ("output", step)
        assert torch.equal(ga.float(), gb.float()) and ga.dtype == gb.dtype, ("input gradient", step)
        assert torch.equal(ha.float(), hb.float()), ("the modified tensor", step)
    for k in cpu_state:
        assert same(cpu_state[k], gpu_state[k]), k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("policy", ["no_avg", "uniform"])
def test_inplace_clamp_with_bounds_the_dtype_cannot_represent_in_front_of_a_site_the_kernels_do_not_cover(policy, dtype):
    """found by the round-5 fuzz campaign: nn.Hardtanh(0.1, 0.7, inplace=True) on a bf16 tensor saturates at 0.10009765625 /
    0.69921875 -- values INSIDE (0.1, 0.7) -- so a gate read off the rectified tensor lets the gradient of clamped elements
    through.  Such an activation is not owned in place (`_hip.bounds_representable`); the results are the CPU path's"""
    cb = qs.MagnitudePruningCallback(running_average=False) if policy == "no_avg" else qs.UniformPruningCallback()
    def site():
        net = nn.Sequential(nn.Hardtanh(0.1, 0.7, inplace=True))
        return qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=2, dimensions={1}, callback=__import__("copy").deepcopy(cb)),
                          activation_layers=[nn.Hardtanh], log=False)
    g = gen(31)
    data = [((torch.randn(6, 16, 9, 8, generator=g) * 0.8).to(dtype), torch.randn(6, 16, 9, 8, generator=g)) for _ in range(6)]
    import numpy as np
    np.random.seed(0)
    cpu, _ = _run(site(), "cpu", data, torch.contiguous_format, True)
    np.random.seed(0)
    gpu, _ = _run(site().to("cuda"), "cuda", data, torch.contiguous_format, True)
    for step, ((ya, ga, ha), (yb, gb, hb)) in enumerate(zip(cpu, gpu)):
        assert same(ya, yb) and torch.equal(ga.float(), gb.float()), step
