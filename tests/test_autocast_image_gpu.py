"""The autocast image (qsparse_amd/fused.py; the DEFAULT since round 5, `set_qsparse_options(autocast_image=False)` opts out): under torch.autocast a fused ReLU -> prune -> quantize site hands
its first convolution / linear consumer the bf16 image of its float32 output and takes that consumer's bf16 gradient as it
is.  The claim is that every VALUE stays the reference's (quantize.py:109-131 + autocast's casts + autograd's float32
accumulation); what is checked:

  * the backward kernel with two gradient streams against torch's own arithmetic (float32 add, clamp, mask, ReLU gate, cast),
    bit for bit, over layouts / dtypes / ragged shapes / with and without the float32 stream;
  * whole sites with real consumers -- a linear layer (deterministic GEMM), a second float32 consumer (residual add), a
    second low-precision consumer -- option on against option off: outputs, consumer outputs, input gradients bit for bit;
  * the image is taken exactly once, never after an in-place write, never outside autocast; a site whose consumer is no
    autocast matmul stops producing it; evaluation mode works; the output is a Tensor subclass only while it carries one."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from golden_io import same
from qsparse_amd import _hip
from qsparse_amd.fused import AutocastImageTensor, fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
DEV = "cuda"
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float32, torch.float16])
def test_backward_kernel_with_two_gradient_streams_equals_autograd_arithmetic(xdt):
    for shape in ((6, 16, 8, 8), (3, 24, 7, 7), (5, 8, 14, 14), (2, 40, 3, 5), (4, 64, 1, 1)):
        for cl in (False, True):
            for with_g32 in (True, False):
                for g2dt in ((torch.bfloat16, torch.float16) if xdt == torch.float32 else (xdt,)):
                    C = shape[1]
                    fmt = torch.channels_last if cl else torch.contiguous_format
                    x = (torch.randn(shape, generator=gen(1)) * 2).to(xdt)
                    x.view(-1)[:3] = torch.tensor([0.0, -0.0, float("nan")], dtype=xdt)
                    x = x.to(DEV).contiguous(memory_format=fmt)
                    mask = (torch.rand(C, generator=gen(2)) > 0.4).to(DEV)
                    scale = torch.tensor([[0.37]], device=DEV)
                    # the forward records the gate bitmap in the layout it addressed
                    _, _, gate = _hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1, pre_relu=True,
                                                want_gate=True)
                    g32 = None
                    if with_g32:
                        g32 = torch.randn(shape, generator=gen(3)) * 3
                        g32.view(-1)[5] = float("nan")
                        g32 = g32.to(DEV).contiguous(memory_format=fmt)
                    g16 = (torch.randn(shape, generator=gen(4)) * 3).to(g2dt).to(DEV).contiguous(memory_format=fmt)
                    lo, hi = -8.0, 7.0
                    gx = _hip.ste_relu_bwd(g32, None, scale, False, lo, hi, mask, gate=gate, g2=g16)
                    total = g16.float() if g32 is None else g32 + g16.float()            # autograd's accumulation
                    lo_b, hi_b = (scale * lo).item(), (scale * hi).item()                # float32 products, as quantize.py:123-129
                    ref = torch.clamp(total, lo_b, hi_b) * mask.view(1, -1, 1, 1)
                    ref = torch.where(x > 0, ref, torch.zeros_like(ref)).to(xdt)         # threshold_backward, then the cast to x's dtype
                    ref = torch.where(x != x, (torch.clamp(total, lo_b, hi_b) * mask.view(1, -1, 1, 1)).to(xdt), ref)   # NaN passes the gate
                    a, b = gx.cpu(), ref.cpu()          # (a NaN only has to be a NaN: payload and sign depend on the instruction mix)
                    assert gx.dtype == xdt and torch.equal(a.isnan(), b.isnan()), (shape, cl, with_g32, g2dt)
                    assert same(torch.where(a.isnan(), torch.zeros_like(a), a), torch.where(b.isnan(), torch.zeros_like(b), b)), \
                        (shape, cl, with_g32, g2dt)


class Net(nn.Module):
    """site -> linear over the last dim (a deterministic GEMM under autocast) [+ a float32 consumer] [+ a second bf16 consumer]"""

    def __init__(self, C, W, residual, second):
        super().__init__()
        self.site = fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1)))
        torch.manual_seed(5)
        self.w1 = nn.Parameter(torch.randn(12, W) * 0.3)
        self.w2 = nn.Parameter(torch.randn(9, W) * 0.3)
        self.residual, self.second = residual, second
        self.seen = []

    def forward(self, x):
        y = self.site(x)
        self.seen.append(type(y))
        out = F.linear(y, self.w1).float().sum(-1)
        if self.second:
            out = out + F.linear(y, self.w2).float().sum(-1)
        if self.residual:
            out = out + (y * 0.5).sum(-1)
        return y, out


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("residual,second", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("shape,xdt", [((6, 16, 8, 8), torch.bfloat16), ((3, 16, 32, 32), torch.bfloat16), ((4, 24, 7, 7), torch.bfloat16),
                                       ((5, 8, 14, 14), torch.float32), ((2, 16, 32, 32), torch.float32), ((6, 16, 8, 8), torch.float16)])
def test_sites_with_real_consumers_are_value_identical(cl, residual, second, shape, xdt):
    """(rows of 64 / 1024 elements, ragged 7x7 maps whose image is a cast of y, float32 inputs as behind a residual add: every
    route by which the image is made -- by the forward kernel itself or by a cast -- and consumed)"""
    runs = []
    C, W = shape[1], shape[3]
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        try:
            net = Net(C, W, residual, second).to(DEV).train()
            trace = []
            for s in range(5):
                x = (torch.randn(shape, generator=gen(20 + s)) * torch.linspace(0.3, 3, C).view(1, -1, 1, 1)).to(xdt).to(DEV)
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last)
                x.requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y, out = net(x)
                out.sum().backward()
                trace += [y.detach().as_subclass(torch.Tensor).clone(), out.detach().clone(), x.grad.clone(), net.w1.grad.clone()]
                net.zero_grad()
            runs.append((trace, net.seen, net.site[0][1].mask.clone(), net.site[1].weight.clone()))
        finally:
            qs.set_qsparse_options(autocast_image=True)
    (ta, seen_a, ma, sa), (tb, seen_b, mb, sb) = runs
    assert all(t is torch.Tensor for t in seen_a)
    assert seen_b[0] is torch.Tensor and all(t is AutocastImageTensor for t in seen_b[2:])      # from the first ACTIVE step on
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), ("trace", i)
    assert torch.equal(ma, mb) and torch.equal(sa, sb)


def test_the_image_is_taken_once_and_only_when_it_is_still_valid():
    qs.set_qsparse_options(autocast_image=True)
    try:
        site = fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1))).to(DEV).train()
        w = torch.randn(5, 8, device=DEV)

        def run(consume):
            x = torch.randn(4, 16, 8, 8, device=DEV).bfloat16().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = site(x)
                return y, consume(y)


        for _ in range(3):
            run(lambda y: F.linear(y, w))
        # taken exactly once
        y, _ = run(lambda y: F.linear(y, w))
        assert type(y) is AutocastImageTensor and "_qs_image" not in y.__dict__
        # an in-place write invalidates it: the consumer casts the WRITTEN values itself
        def inplace(y):
            y.mul_(2.0)
            return F.linear(y, w)
        y, out = run(inplace)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert torch.equal(out, F.linear(y.detach().as_subclass(torch.Tensor), w))
        # a gradient hook on the output cancels the image: the hook then sees the WHOLE gradient, as without the extension
        seen = []
        def hooked(y):
            y.register_hook(lambda g: seen.append(g.clone()))
            return F.linear(y, w)
        y, out = run(hooked)
        assert "_qs_image" not in y.__dict__
        out.float().sum().backward()
        assert len(seen) == 1 and seen[0].dtype == torch.float32 and float(seen[0].abs().sum()) > 0
        # outside autocast nothing is substituted (and nothing is made)
        x = torch.randn(4, 16, 8, 8, device=DEV).bfloat16()
        assert type(site(x)) is torch.Tensor
        # a site whose consumer is no autocast matmul stops making images after the first unused one
        lone = fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1))).to(DEV).train()
        kinds = []
        for _ in range(6):
            x = torch.randn(4, 16, 8, 8, device=DEV).bfloat16().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = lone(x)
                kinds.append(type(y))
                F.adaptive_avg_pool2d(y, 1).sum().backward()
        assert kinds[-1] is torch.Tensor and AutocastImageTensor in kinds
        # evaluation mode: forward only, the image still saves the consumer's cast
        site = fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1))).to(DEV).train()
        for _ in range(3):
            run(lambda y: F.linear(y, w))
        site.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            x = torch.randn(4, 16, 8, 8, device=DEV).bfloat16()
            y = site(x)
            assert type(y) is AutocastImageTensor
            a = F.linear(y, w)
            qs.set_qsparse_options(autocast_image=False)
            assert torch.equal(a, F.linear(site(x), w))
    finally:
        qs.set_qsparse_options(autocast_image=True)


def test_whole_step_graph_capture_with_the_image_equals_eager():
    """the subclass dispatch is host-side only: a captured and replayed training step (graphs.GraphedStep) with the extension
    on gives the eager results"""
    from qsparse_amd import graphs
    outs = []
    try:
        for graphed in (False, True):
            qs.set_qsparse_options(autocast_image=True, graph_safe=True)
            net = Net(16, 8, True, False).to(DEV).train()

            def train_step(x):
                x = x.detach().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y, out = net(x)
                (gx,) = torch.autograd.grad(out.sum(), x)
                return y.as_subclass(torch.Tensor), out, gx

            step = graphs.GraphedStep(net, train_step, settle=1) if graphed else train_step
            trace = []
            for s in range(8):
                x = (torch.randn(6, 16, 8, 8, generator=gen(60 + s)) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16().to(DEV)
                y, out, gx = step(x)
                trace += [y.detach().clone(), out.detach().clone(), gx.clone()]
            if graphed:
                assert step.captured
                step.finish()
            outs.append((trace, net.site[0][1].mask.clone(), net.site[1].weight.clone()))
    finally:
        qs.set_qsparse_options(autocast_image=True, graph_safe=False)
    (ta, ma, sa), (tb, mb, sb) = outs
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert same(a.cpu(), b.cpu()), i
    assert torch.equal(ma, mb) and torch.equal(sa, sb)


# ----------------------------------------------------------------------------------------------------------------------
# Default since round 5.  What made it opt-in before -- something that observes the output's gradient AFTER the consumer took the
# image saw only the float32 consumers' share -- is closed: late `register_hook` / `retain_grad` and `torch.autograd.grad(...,
# inputs=[output])` see the whole gradient.  Every scenario below runs with the image (the default) and without it and compares
# what the USER observes bit for bit: outputs, hook arguments, retained gradients, input and weight gradients.
# ----------------------------------------------------------------------------------------------------------------------
def _pair():
    return fuse_prune_quantize_pairs(nn.Sequential(
        nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
        qs.quantize(bits=4, channelwise=-1, timeout=1)))


class Block(nn.Module):
    """a site in front of a convolution, with every kind of second consumer a network may have"""

    def __init__(self, mode):
        super().__init__()
        self.site = _pair()
        torch.manual_seed(7)
        self.conv = nn.Conv2d(16, 16, 1, bias=False)
        self.lin = nn.Linear(8, 6, bias=False)
        self.mode = mode
        self.observed = []
        self.kinds = []

    def forward(self, x):
        y = self.site(x)
        self.kinds.append(type(y))
        m = self.mode
        if m == "hook_before":
            y.register_hook(lambda g: self.observed.append(g.clone()))
        if m == "inplace_before" and len(self.kinds) > 1:      # an in-place consumer in front of the convolution: the image is stale
            y = F.hardtanh_(y, -1.0, 1.0)                        # (not on step 0: the idle site's output is nn.ReLU's, which autograd keeps)
        z = self.lin(y) if m == "linear" else self.conv(y)
        if m == "hook_after":
            y.register_hook(lambda g: self.observed.append(g.clone()))
        if m == "hook_after_replacing":
            y.register_hook(lambda g: g * 2.0)
        if m == "retain_after":
            y.retain_grad()
            self.retained = y
        if m == "grad_wrt_output":
            self.held = y
        out = z.float().mean((2, 3)) if z.dim() == 4 else z.float()
        if m in ("residual", "hook_after", "hook_after_replacing", "retain_after", "grad_wrt_output"):
            out = out + (y * 0.25).mean((2, 3))[:, :out.shape[1]] if out.dim() == 2 else out
        if m == "cat":
            out = out + torch.cat([y, y * 2.0], 1).mean((2, 3))[:, :16]
        if m == "view":
            out = out + y.flatten(1)[:, :16] + y[:, :, 0, 0]
        if m == "second_conv":
            out = out + self.conv(y).float().mean((2, 3))
        return out


MODES = ["plain", "linear", "residual", "cat", "view", "second_conv", "hook_before", "hook_after", "hook_after_replacing", "retain_after",
         "grad_wrt_output", "inplace_before"]


@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("mode", MODES)
def test_everything_observable_is_the_same_with_and_without_the_image(mode, cl):
    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        try:
            net = Block(mode).to(DEV).train()
            trace = []
            for s in range(5):
                x = (torch.randn(6, 16, 8, 8, generator=gen(40 + s)) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16().to(DEV)
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last)
                x.requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(x)
                loss = (out * torch.linspace(-1, 1, out.shape[1], device=DEV)).sum()
                if mode == "grad_wrt_output":
                    gy, gx = torch.autograd.grad(loss, [net.held, x])
                    trace += [gy.as_subclass(torch.Tensor).clone(), gx.clone()]
                    continue
                loss.backward()
                trace += [out.detach().clone(), x.grad.clone(), (net.lin if mode == "linear" else net.conv).weight.grad.clone()]
                if mode == "retain_after":
                    trace.append(net.retained.grad.as_subclass(torch.Tensor).clone())
                net.zero_grad()
            runs.append((trace, net.observed, net.kinds))
        finally:
            qs.set_qsparse_options(autocast_image=True)
    (ta, oa, ka), (tb, ob, kb) = runs
    assert len(ta) == len(tb) and len(oa) == len(ob)
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), ("trace", i)
    for i, (a, b) in enumerate(zip(oa, ob)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), ("gradient seen by the hook", i)      # (step 0: operators idle, bf16 passes through)
    if mode.startswith("hook") and mode != "hook_after_replacing":
        assert len(ob) == 5 and all(float(g.abs().sum()) > 0 for g in ob[1:])
    assert all(k is torch.Tensor for k in ka) and AutocastImageTensor in kb          # the image route really ran


def test_the_image_is_the_default_and_takes_the_steady_state_fast_path():
    """library defaults: under autocast a training site returns the subclass, also from `_FastPair` (the steady-state fast path);
    outside autocast and with the option off a plain tensor"""
    assert qs.get_qsparse_option("autocast_image") is True
    site = _pair().to(DEV).train()
    conv = nn.Conv2d(16, 16, 1, bias=False).to(DEV)
    kinds, fast = [], []
    for s in range(8):
        x = torch.randn(4, 16, 8, 8, generator=gen(s)).bfloat16().to(DEV).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = site(x)
            kinds.append(type(y))
            fast.append(site.__dict__.get("_qs_fast") is not None)
            conv(y).float().sum().backward()
    assert kinds[-1] is AutocastImageTensor and fast[-1] and fast[-2]
    assert type(site(torch.randn(4, 16, 8, 8, device=DEV).bfloat16())) is torch.Tensor      # no autocast: nothing to hand over


def test_copies_and_pickles_of_the_output_are_plain_tensors():
    import copy
    import pickle
    site = _pair().to(DEV).train()
    conv = nn.Conv2d(16, 4, 1).to(DEV)
    for s in range(4):
        x = torch.randn(4, 16, 8, 8, generator=gen(s)).bfloat16().to(DEV)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = site(x)
            if s < 3:
                conv(y)          # (a site whose images nobody takes stops making them)
    assert type(y) is AutocastImageTensor
    for z in (copy.deepcopy(y.detach()), pickle.loads(pickle.dumps(y.detach())), y.detach().clone(), y + 0):
        assert type(z) is torch.Tensor and torch.equal(z, y.as_subclass(torch.Tensor))
    net = nn.Sequential(site, nn.Conv2d(16, 4, 1)).to(DEV)
    twin = copy.deepcopy(net)                       # a network holding a site deep-copies and keeps working
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.equal(net(x), twin(x))


def test_under_distributed_data_parallel_the_gradients_are_those_of_the_plain_route():
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        grads = []
        for image in (False, True):
            qs.set_qsparse_options(autocast_image=image)
            torch.manual_seed(3)
            net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), _pair(), nn.Conv2d(16, 8, 3, padding=1), nn.Flatten(), nn.Linear(8 * 64, 5))
            net = nn.parallel.DistributedDataParallel(net.to(DEV).train(), device_ids=[0])
            for s in range(4):
                x = torch.randn(4, 3, 8, 8, generator=gen(s)).to(DEV)
                net.zero_grad()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    net(x).float().sum().backward()
            grads.append([p.grad.clone() for p in net.parameters() if p.grad is not None])
        for a, b in zip(*grads):
            assert torch.allclose(a, b, rtol=2e-2, atol=1e-3)      # (the convolutions' own algorithm choice may differ between runs)
    finally:
        qs.set_qsparse_options(autocast_image=True)
        dist.destroy_process_group()


def test_a_consumer_that_reads_the_output_before_the_convolution_cancels_the_image():
    """three gradient streams (a residual read BEFORE the convolution, the convolution, a `cat` after it): autograd sums them in
    reverse order of creation; only a FIRST consumer's share is the last term in both routes, so anything that touches the output
    before the image consumer cancels the image (found by tests/fuzz/fuzz_image.py: one float32 ulp in the summed gradient)"""
    site = _pair().to(DEV).train()
    conv = nn.Conv2d(16, 16, 1, bias=False).to(DEV)
    for s in range(4):
        x = torch.randn(4, 16, 8, 8, generator=gen(s)).bfloat16().to(DEV).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = site(x)
            if s == 3:
                assert type(y) is AutocastImageTensor and "_qs_image" in y.__dict__
                assert y.shape[1] == 16 and y.dim() == 4 and y.is_contiguous() and y.dtype == torch.float32       # metadata: no consumer
                assert "_qs_image" in y.__dict__
                r = y * 0.5                                     # a consumer
                assert "_qs_image" not in y.__dict__
            conv(y).float().sum().backward()


# ----------------------------------------------------------------------------------------------------------------------
# The weight side of the same mechanism (batch.py `_weight_images`): the multi-tensor weight path hands every layer its quantized
# weight together with the low-precision image its convolution / linear would cast it to -- ONE cast of the flat buffer for all
# layers -- and takes the low-precision weight gradients directly (one multi-tensor copy per hand-out group).
# ----------------------------------------------------------------------------------------------------------------------
class _Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(4)
        self.a, self.b, self.c = nn.Linear(24, 32), nn.Linear(32, 32, bias=False), nn.Linear(32, 5)
        self.skip_b = False
        self.observe = None          # a list: read c's weight by hand, use it, THEN hook it (a late observer)
        self.kinds = []

    def forward(self, x):
        h = F.relu(self.a(x))
        if not self.skip_b:
            h = F.relu(self.b(h))
        if self.observe is None:
            return self.c(h)
        w, b = self.c.weight, self.c.bias
        self.kinds.append(type(w))
        out = F.linear(h, w, b)
        w.register_hook(lambda g: self.observe.append(g.detach().clone()))
        return out


@pytest.mark.parametrize("quant", [dict(channelwise=-1), dict(channelwise=1), dict(channelwise=0, bias_bits=8, callback="decimal")])
@pytest.mark.parametrize("pruned", [False, True])
def test_weight_images_are_value_identical(quant, pruned):
    """linear layers (deterministic GEMMs) under bf16 autocast, with and without the images: outputs, input gradients, raw weight
    and bias gradients, quantizer and prune state -- bit for bit; through a skipped layer (roll-back), an evaluation step and a
    late hook on a handed-out weight"""
    import copy
    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        try:
            net = _Mlp()
            if pruned:
                net = qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=1), weight_layers=[nn.Linear], log=False)
            kw = dict(quant)
            cb = qs.DecimalQuantizer() if kw.pop("callback", None) == "decimal" else None
            net = qs.convert(net, qs.quantize(bits=4, timeout=1, callback=cb, **kw), weight_layers=[nn.Linear], log=False).to(DEV).train()
            seen, kinds, trace = [], [], []
            for s in range(8):
                net.skip_b = s == 3
                x = torch.randn(16, 24, generator=gen(70 + s)).to(DEV).requires_grad_(True)
                if s == 5:
                    net.eval()
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                        trace.append(net(x).clone())
                    net.train()
                    continue
                net.observe = seen if s == 6 else None
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(x)
                out.float().sum().backward()
                trace += [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None]
                net.zero_grad()
            runs.append((trace, seen, net.kinds, {k: v.detach().clone() for k, v in net.state_dict().items()}))
        finally:
            qs.set_qsparse_options(autocast_image=True)
    (ta, sa_, ka, sta), (tb, sb_, kb, stb) = runs
    assert len(ta) == len(tb) and len(sa_) == len(sb_) == 1
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), ("trace", i)
    assert sa_[0].dtype == sb_[0].dtype == torch.float32 and same(sa_[0].cpu(), sb_[0].cpu())      # the hook saw the WHOLE weight gradient
    assert ka == [torch.Tensor] and kb == [AutocastImageTensor]
    for k in sta:
        assert same(sta[k].cpu(), stb[k].cpu()), k


def test_weight_images_cost_one_cast_per_forward(monkeypatch):
    """the cast launches autocast puts in front of every layer are gone: no `aten::_to_copy` of a weight-sized float32 tensor
    inside the layers' forwards once the images are handed out"""
    net = qs.convert(_Mlp(), qs.quantize(bits=4, timeout=1, channelwise=-1), weight_layers=[nn.Linear], log=False).to(DEV).train()
    x = torch.randn(16, 24, device=DEV)
    for _ in range(3):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            net(x).float().sum().backward()
    from torch.profiler import profile, ProfilerActivity
    counts = {}
    for image in (True, False):
        qs.set_qsparse_options(autocast_image=image)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                net(x).float().sum().backward()          # (the option epoch changed: one full-path step)
            with profile(activities=[ProfilerActivity.CPU]) as prof:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    net(x).float().sum().backward()
            counts[image] = sum(e.count for e in prof.key_averages() if e.key == "aten::_to_copy")
        finally:
            qs.set_qsparse_options(autocast_image=True)
    assert counts[True] < counts[False], counts


@pytest.mark.parametrize("act", ["relu", "relu6"])
@pytest.mark.parametrize("cl", [False, True])
def test_quantize_only_sites_hand_out_images_too(act, cl):
    """`convert(model, quantize(...), activation_layers=[nn.ReLU])` -- quantization without pruning: the site is
    Sequential(act, QuantizeLayer), one qs_quantize_step call per forward, which writes the image in the same pass (ABI v21)"""
    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        try:
            torch.manual_seed(3)
            net = nn.Sequential(nn.ReLU() if act == "relu" else nn.ReLU6())
            site = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1), activation_layers=[type(net[0])], log=False).to(DEV).train()
            w = torch.randn(12, 8, device=DEV)
            trace, kinds = [], []
            for s in range(6):
                x = (torch.randn(6, 16, 5, 8, generator=gen(90 + s)) * 2).bfloat16().to(DEV)
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last)
                x.requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = site(x)
                    kinds.append(type(y))
                    out = F.linear(y, w).float().sum(-1) + ((y * 0.5).sum(-1) if s % 2 else 0)
                out.sum().backward()
                trace += [y.detach().as_subclass(torch.Tensor).clone(), out.detach().clone(), x.grad.clone()]
            runs.append((trace, kinds, {k: v.detach().clone() for k, v in site.state_dict().items()}))
        finally:
            qs.set_qsparse_options(autocast_image=True)
    (ta, ka, sa), (tb, kb, sb) = runs
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), ("trace", i)
    assert all(k is torch.Tensor for k in ka) and all(k is AutocastImageTensor for k in kb[2:]), kb
    for k in sa:
        assert same(sa[k].cpu(), sb[k].cpu()), k


def test_two_replacing_hooks_registered_after_the_consumer_chain_like_ordinary_hooks():
    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        try:
            site = _pair().to(DEV).train()
            torch.manual_seed(1)
            lin = nn.Linear(8, 6, bias=False).to(DEV)
            trace = []
            for s in range(4):
                x = torch.randn(4, 16, 8, 8, generator=gen(s)).bfloat16().to(DEV).requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = site(x)
                    out = lin(y)
                    y.register_hook(lambda g: g * 0.5)
                    y.register_hook(lambda g: g + 1.0)
                out.float().sum().backward()
                trace.append(x.grad.clone())
            runs.append(trace)
        finally:
            qs.set_qsparse_options(autocast_image=True)
    for a, b in zip(*runs):
        assert same(a.cpu(), b.cpu())


@pytest.mark.parametrize("site_kind", ["pair", "act_quantize"])
def test_inputs_in_other_layouts_come_back_in_their_layout_with_or_without_the_image(site_kind):
    """a transposed / permuted activation under autocast: the site runs on its contiguous copy, `_hip.keeps_layout` puts the
    result into the layout the reference returns (dense in the input's stride order) -- whether or not the site handed out an
    image on the way, the values, the layout and the gradients are those of the plain route"""
    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        try:
            torch.manual_seed(3)
            if site_kind == "pair":
                site = _pair().to(DEV).train()
            else:
                site = qs.convert(nn.Sequential(nn.ReLU()), qs.quantize(bits=4, channelwise=-1, timeout=1), activation_layers=[nn.ReLU],
                                  log=False).to(DEV).train()
            w = torch.randn(12, 24, device=DEV)
            trace = []
            for s in range(6):
                x = (torch.randn(6, 24, 16, generator=gen(40 + s)) * 2).bfloat16().to(DEV).transpose(1, 2)    # [B, C, T] view of [B, T, C]
                x.requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = site(x)
                    assert y.stride() == x.stride(), (y.stride(), x.stride())
                    out = F.linear(y, w).float().sum(-1) + ((y * 0.5).sum(-1) if s % 2 else 0)
                out.sum().backward()
                trace += [y.detach().as_subclass(torch.Tensor).clone(), out.detach().clone(), x.grad.clone()]
            runs.append((trace, {k: v.detach().clone() for k, v in site.state_dict().items()}))
        finally:
            qs.set_qsparse_options(autocast_image=True)
    (ta, sa), (tb, sb) = runs
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), ("trace", i)
    for k in sa:
        assert same(sa[k].cpu(), sb[k].cpu()), k


# ----------------------------------------------------------------------------------------------------------------------------------
# round 6: the second image (a second autocast consumer with a gradient slot of its own) and the promoting add in front of a site
# ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("g2dt", [torch.bfloat16, torch.float16])
def test_all_float32_backward_kernel_riders_equal_autograd_arithmetic(g2dt):
    """qs_site_bwd_v's riders on the all-float32 form: (g + f32(g3)) + f32(g2) with each term optional from the left, and the image
    RNE(gx) written next to gx -- against torch's own float32 adds, clamp, mask, gate and cast, bit for bit"""
    from qsparse_amd import fused
    for shape in ((6, 16, 8, 8), (3, 24, 7, 7), (2, 40, 3, 5), (5, 64)):
        for cl in ((False, True) if len(shape) == 4 else (False,)):
            for with_g32 in (True, False):
                C = shape[1]
                fmt = torch.channels_last if cl else torch.contiguous_format
                view = (1, -1) + (1,) * (len(shape) - 2)
                p = qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1).to(DEV).train()
                q = qs.quantize(bits=4, channelwise=-1, timeout=1).to(DEV).train()
                site = fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.Sequential(nn.ReLU(), p), q)))[0]
                for s in range(3):          # into the live regime: plan built, mask and scale set
                    site((torch.randn(shape, generator=gen(40 + s)) * torch.linspace(0.3, 3, C).view(view)).to(DEV).contiguous(memory_format=fmt))
                plan = q.__dict__["_qs_site_plan"]
                assert plan is not None and plan.xdt == torch.float32
                x = (torch.randn(shape, generator=gen(1)) * 2)
                x.view(-1)[:3] = torch.tensor([0.0, -0.0, float("nan")])
                x = x.to(DEV).contiguous(memory_format=fmt)
                _, _, gate = _hip.quant_fwd("scaler", x, q.weight.detach(), -1, torch.float32, chan_mask=p.mask.detach().view(-1),
                                            mask_channel_index=1, pre_relu=True, want_gate=True)
                def mk(seed, dt, nan_at=None):
                    t = (torch.randn(shape, generator=gen(seed)) * 3).to(dt)
                    if nan_at is not None:
                        t.view(-1)[nan_at] = float("nan")
                    return t.to(DEV).contiguous(memory_format=fmt)

                g32 = mk(3, torch.float32, 5) if with_g32 else None
                g2, g3 = mk(4, g2dt), mk(5, g2dt)
                gx = torch.empty_like(x)
                gimg = torch.empty_like(x, dtype=g2dt)
                _hip.site_bwd(plan.ref, g32, gate.bits, gx, 0, -8.0, 7.0, g2=g2, g3=g3, gx_image=gimg)
                total = g3.float() if g32 is None else g32 + g3.float()              # autograd's accumulation, in arrival order
                total = total + g2.float()
                scale = q.weight.detach().view(())
                lo_b, hi_b = (scale * -8.0).item(), (scale * 7.0).item()
                clamped = torch.clamp(total, lo_b, hi_b) * p.mask.detach().view(view)
                ref = torch.where(x > 0, clamped, torch.zeros_like(clamped))
                ref = torch.where(x != x, clamped, ref)
                a, b = gx.cpu(), ref.cpu()
                tag = (shape, cl, with_g32, g2dt)
                assert torch.equal(a.isnan(), b.isnan()), tag
                assert same(torch.where(a.isnan(), torch.zeros_like(a), a), torch.where(b.isnan(), torch.zeros_like(b), b)), tag
                ia, ib = gimg.cpu(), gx.to(g2dt).cpu()                                # the image is ATen's cast of gx
                assert torch.equal(ia.isnan(), ib.isnan()), tag
                assert same(torch.where(ia.isnan(), torch.zeros_like(ia), ia), torch.where(ib.isnan(), torch.zeros_like(ib), ib)), tag
                # without the third stream, and the image alone
                gx2 = torch.empty_like(x)
                _hip.site_bwd(plan.ref, g32, gate.bits, gx2, 0, -8.0, 7.0, g2=g2, gx_image=gimg)
                gx3 = torch.empty_like(x)
                _hip.site_bwd(plan.ref, g32, gate.bits, gx3, 0, -8.0, 7.0, g2=g2)
                assert same(gx2.cpu().nan_to_num(7.0), gx3.cpu().nan_to_num(7.0)), tag


class Bottleneckish(nn.Module):
    """a ResNet bottleneck's data flow around two sites: y0 = site0(x) feeds a first convolution, a down-sampling convolution (the
    SECOND autocast consumer) or the residual add (`conv(...) + y0`: bf16 + float32), whose sum is site1's input"""

    def __init__(self, C, down, kind="pair"):
        super().__init__()
        def site():
            if kind == "act_q":          # a quantize-only recipe: convert(model, quantize(...), activation_layers=[nn.ReLU])
                return fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=1))))[0]
            return fuse_prune_quantize_pairs(nn.Sequential(
                nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                qs.quantize(bits=4, channelwise=-1, timeout=1)))
        self.site0, self.site1 = site(), site()
        torch.manual_seed(11)
        self.conv1 = nn.Conv2d(C, C, 1, bias=False)
        self.bn = nn.BatchNorm2d(C)
        self.down = nn.Conv2d(C, C, 1, bias=False) if down else None
        self.head = nn.Conv2d(C, 8, 1, bias=False)

    def forward(self, x):
        y0 = self.site0(x)
        out = self.bn(self.conv1(y0))
        y1 = self.site1(out + (y0 if self.down is None else self.down(y0)))
        return self.head(y1).float().mean((2, 3))


@pytest.mark.parametrize("kind", ["pair", "act_q"])
@pytest.mark.parametrize("down", [False, True])
@pytest.mark.parametrize("cl", [False, True])
def test_bottleneck_data_flow_second_image_and_promoting_add_are_value_identical(down, cl, kind):
    from qsparse_amd.fused import ROUTES
    shape = (6, 16, 8, 8)
    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        before = dict(ROUTES)
        try:
            net = Bottleneckish(16, down, kind).to(DEV).train()
            if cl:
                net = net.to(memory_format=torch.channels_last)
            trace = []
            for s in range(6):
                x = (torch.randn(shape, generator=gen(60 + s)) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).to(DEV)
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last)
                x.requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(x)
                (out * torch.linspace(-1, 1, 8, device=DEV)).sum().backward()
                trace += [out.detach().clone(), x.grad.clone(), net.bn.weight.grad.clone(), net.bn.bias.grad.clone()]
                net.zero_grad()
            runs.append((trace, {k: ROUTES[k] - before.get(k, 0) for k in ROUTES}))
        finally:
            qs.set_qsparse_options(autocast_image=True)
    (ta, ra), (tb, rb) = runs
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert a.dtype == b.dtype and same(a.cpu(), b.cpu()), (down, cl, kind, i)
    assert not any(ra.get(k) for k in ("image", "second_image", "grad_image")), ra
    if down:
        assert rb.get("second_image", 0) >= 4 and not rb.get("grad_image"), rb      # conv1 took the first image, the down conv the second
    else:
        assert rb.get("grad_image", 0) >= 4 and not rb.get("grad_image_cast"), rb    # the add's bf16 operand got the kernel's image


def test_second_image_is_dropped_when_anything_touches_the_output_in_between():
    """conv(y); y * 0.5; conv2(y): the float32 consumer between the two convolutions would be added AFTER the second convolution's
    share in autograd's order -- the second convolution casts for itself, the values stay the reference's"""
    from qsparse_amd.fused import ROUTES

    class N(nn.Module):
        def __init__(self):
            super().__init__()
            self.site = fuse_prune_quantize_pairs(nn.Sequential(
                nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                qs.quantize(bits=4, channelwise=-1, timeout=1)))
            torch.manual_seed(3)
            self.a, self.b = nn.Conv2d(16, 8, 1, bias=False), nn.Conv2d(16, 8, 1, bias=False)

        def forward(self, x):
            y = self.site(x)
            o = self.a(y).float().mean((2, 3))
            o = o + (y * 0.5).mean((2, 3))[:, :8]
            return o + self.b(y).float().mean((2, 3))

    runs = []
    for image in (False, True):
        qs.set_qsparse_options(autocast_image=image)
        before = ROUTES["second_image"]
        try:
            net = N().to(DEV).train()
            trace = []
            for s in range(5):
                x = (torch.randn(4, 16, 8, 8, generator=gen(80 + s)) * 2).to(DEV).requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(x)
                out.sum().backward()
                trace += [out.detach().clone(), x.grad.clone()]
            runs.append(trace)
            assert ROUTES["second_image"] == before
        finally:
            qs.set_qsparse_options(autocast_image=True)
    for a, b in zip(*runs):
        assert same(a.cpu(), b.cpu())


def test_a_site_whose_image_nobody_took_offers_it_again(monkeypatch):
    """one forward whose consumer is not an autocast matmul (an evaluation pass into a pooling layer, a hook that touched
    the output first) used to switch the image off for the rest of the process; now the site offers it again after REARM_EVERY steps
    or as soon as an option changes"""
    from qsparse_amd import fused
    from qsparse_amd.fused import ROUTES
    monkeypatch.setattr(fused, "REARM_EVERY", 3)
    site = fuse_prune_quantize_pairs(nn.Sequential(
        nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
        qs.quantize(bits=4, channelwise=-1, timeout=1))).to(DEV).train()
    conv = nn.Conv2d(16, 8, 1, bias=False).to(DEV)

    def step(use_conv):
        x = (torch.randn(4, 16, 8, 8, generator=gen(1)) * 2).to(DEV).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = site(x)
            out = conv(y).float().sum() if use_conv else (y * 2.0).sum()
        out.backward()
        return type(y)

    for _ in range(4):
        assert step(True) in (torch.Tensor, AutocastImageTensor)
    assert step(True) is AutocastImageTensor
    before = dict(ROUTES)
    assert step(False) is AutocastImageTensor              # made, not taken ...
    assert step(True) is torch.Tensor                       # ... so the next step makes none
    assert ROUTES["image_disarmed"] == before.get("image_disarmed", 0) + 1
    kinds = [step(True) for _ in range(5)]
    assert AutocastImageTensor in kinds                    # offered again after REARM_EVERY steps -- and taken: it stays
    assert kinds[-1] is AutocastImageTensor and ROUTES["image_rearmed"] == before.get("image_rearmed", 0) + 1
    assert step(False) is AutocastImageTensor
    assert step(True) is torch.Tensor
    qs.set_qsparse_options(relu_gate=True)                 # any option change re-arms at once
    assert step(True) is AutocastImageTensor
