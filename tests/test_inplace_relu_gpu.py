"""nn.ReLU(inplace=True) in front of a site (torchvision-style networks).

The plain module writes relu(x) into x's storage and autograd runs a ReLU backward pass of its own.  The fused sites keep
the first (other holders of x must see relu(x)) and drop the second: `_OwnedRelu` / `_Tap` (qsparse_amd/fused.py) hand the
site an alias of the modified tensor, the site's backward gates with the bits it recorded, and the gradient passes through
the in-place node untouched -- unless the modified tensor has a further consumer, whose share is gated there.  Everything is
compared bit for bit with the module-by-module route (`fold_relu=False`: ATen's relu_ and its backward)."""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def _site(kind, inplace=True, quantizer="scaler"):
    cb = qs.DecimalQuantizer() if quantizer == "decimal" else None
    net = nn.Sequential(nn.ReLU(inplace=inplace))
    if kind in ("pair", "relu_p"):
        net = qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=1, dimensions={1}),
                         activation_layers=[nn.ReLU], log=False)
    if kind in ("pair", "relu_q"):
        net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1, callback=cb), activation_layers=[nn.ReLU], log=False)
    return net.cuda().train()


def _find(fn, name, seen=None):
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return None
    seen.add(fn)
    if type(fn).__name__ == name:
        return fn
    for nxt, _ in fn.next_functions:
        hit = _find(nxt, name, seen)
        if hit is not None:
            return hit
    return None


def test_the_owned_relu_keeps_nothing_for_its_backward():
    """ATen's in-place ReLU saves its result (2 B/elem until the backward); the owned one relies on the site's bitmap"""
    net = _site("pair")
    g = torch.Generator().manual_seed(0)
    for _ in range(4):
        x0 = torch.randn(6, 16, 5, 7, generator=g).cuda().requires_grad_()
        y = net(x0 * 1.0)
        node = _find(y.grad_fn, "_OwnedReluBackward")
        kept = None if node is None else len(node.saved_tensors)
        y.sum().backward()
    assert kept == 0


def _graph_names(fn, seen=None):
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return set()
    seen.add(fn)
    names = {type(fn).__name__}
    for nxt, _ in fn.next_functions:
        names |= _graph_names(nxt, seen)
    return names


def _run(kind, fold, inplace, dtype, channels_last, second_consumer=False, quantizer="scaler"):
    qs.set_qsparse_options(fold_relu=fold)
    try:
        net = _site(kind, inplace, quantizer)
        g = torch.Generator().manual_seed(0)
        out = []
        for step in range(5):
            x0 = torch.randn(6, 16, 5, 7, generator=g).cuda().to(dtype)
            if channels_last:
                x0 = x0.contiguous(memory_format=torch.channels_last)
            x0.requires_grad_()
            h = x0 * 1.0                                     # what a convolution / batch norm hands the in-place ReLU
            y = net(h)
            loss = (y.float() * torch.randn(y.shape, generator=g).cuda() * 3).sum()
            if second_consumer:                              # someone else still holds the (modified) tensor and uses it
                loss = loss + (h.float() * torch.randn(h.shape, generator=g).cuda()).sum()
            names = _graph_names(y.grad_fn)
            loss.backward()
            out.append((y.detach().clone(), h.detach().clone(), x0.grad.clone()))
        net.eval()
        with torch.no_grad():
            x0 = torch.randn(6, 16, 5, 7, generator=g).cuda().to(dtype)
            h = x0 * 1.0
            out.append((net(h).clone(), h.clone(), x0))
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        return out, state, names
    finally:
        qs.set_qsparse_options(fold_relu=True)


def _same(a, b):
    (oa, sa, _), (ob, sb, _) = a, b
    for i, (ta, tb) in enumerate(zip(oa, ob)):
        for j, (u, v) in enumerate(zip(ta, tb)):
            assert u.dtype == v.dtype and torch.equal(u, v), (i, j)
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["pair", "relu_q", "relu_p"])
def test_inplace_relu_site_equals_the_module_by_module_route(kind, dtype, channels_last):
    owned = _run(kind, True, True, dtype, channels_last)
    plain = _run(kind, False, True, dtype, channels_last)
    _same(owned, plain)
    # x's storage holds relu(x) afterwards, as with the plain module
    for y, h, _ in owned[0]:
        assert bool((h >= 0).all())
    # ... and the ReLU's backward is not a pass of its own any more
    assert "ReluBackward0" in plain[2] and "ReluBackward0" not in owned[2] and "_OwnedReluBackward" in owned[2]
    # same values as the out-of-place ReLU's fused site
    out_of_place = _run(kind, True, False, dtype, channels_last)
    for (ya, _, ga), (yb, _, gb) in zip(owned[0][:5], out_of_place[0][:5]):
        assert torch.equal(ya, yb) and torch.equal(ga, gb)


@pytest.mark.parametrize("kind", ["pair", "relu_q", "relu_p"])
def test_a_second_consumer_of_the_modified_tensor_is_gated_too(kind):
    _same(_run(kind, True, True, torch.float32, False, second_consumer=True),
          _run(kind, False, True, torch.float32, False, second_consumer=True))


def test_decimal_quantizer_behind_an_inplace_relu():
    _same(_run("pair", True, True, torch.bfloat16, True, quantizer="decimal"),
          _run("pair", False, True, torch.bfloat16, True, quantizer="decimal"))


def test_a_leaf_that_requires_grad_raises_as_with_the_plain_module():
    net = _site("relu_q")
    for _ in range(3):
        net(torch.randn(2, 4, 3, 3, device="cuda") * 1.0)
    x = torch.randn(2, 4, 3, 3, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="leaf Variable that requires grad"):
        net(x)


def test_a_view_keeps_atens_own_inplace_route():
    net = _site("relu_q")
    g = torch.Generator().manual_seed(1)
    for _ in range(4):
        base = (torch.randn(2, 8, 3, 3, generator=g).cuda().requires_grad_()) * 1.0
        y = net(base[:, :4])                                  # in-place ReLU on a view: ATen's view bookkeeping
        assert "_OwnedReluBackward" not in _graph_names(y.grad_fn)
        y.sum().backward()


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,quantizer", [("pair", "scaler"), ("relu_q", "scaler"), ("pair", "decimal"), ("relu_p", "scaler")])
def test_the_apply_kernel_writes_relu_back_itself(kind, quantizer, dtype, channels_last, monkeypatch):
    """the in-place ReLU's FORWARD is not a pass of its own either.  The site's apply kernel loads every
    element of x anyway and stores relu(x) back into x's storage (`xback_out`, include/qsparse_hip.h): ATen's `relu_` is not
    called for the quantizing sites, in training (gate recorded) and in evaluation / no-grad forwards alike; x's storage holds
    exactly what the reference's CPU `relu_` leaves there -- NaN and -0.0 pass, -inf becomes +0 -- and the site's output and
    gradient are those of the module-by-module route.  (The prune-only site has no such kernel: it keeps ATen's pass.)"""
    calls = []
    real = torch.Tensor.relu_
    monkeypatch.setattr(torch.Tensor, "relu_", lambda t: (calls.append(1), real(t))[1])
    net, plain = _site(kind, True, quantizer), _site(kind, True, quantizer)
    g = torch.Generator().manual_seed(3)
    special = torch.tensor([-0.0, float("nan"), float("-inf"), float("inf"), -1.5, 2.5, -1e-30, 0.0])
    for step in range(6):
        train = step < 4
        net.train(train), plain.train(train)
        x0 = torch.randn(6, 16, 6, 8, generator=g).to(dtype)
        if step == 2:         # training step: finite specials only (a NaN / Inf would poison the running statistics for good)
            x0.view(-1)[7:15] = torch.tensor([-0.0, -1.5, 2.5, -1e-30, 0.0, -3.0, 1e-30, 4.0]).to(dtype)
        if step == 5:         # evaluation: statistics are frozen
            x0.view(-1)[7:15] = special.to(dtype)
        xa = x0.clone().cuda()
        if channels_last:
            xa = xa.contiguous(memory_format=torch.channels_last)
        xb = xa.clone()
        xa.requires_grad_(train), xb.requires_grad_(train)
        del calls[:]
        qs.set_qsparse_options(fold_relu=True)
        with torch.set_grad_enabled(train):
            ha = xa * 1.0
            ya = net(ha)
        n_owned = len(calls)
        qs.set_qsparse_options(fold_relu=False)
        try:
            with torch.set_grad_enabled(train):
                hb = xb * 1.0
                yb = plain(hb)
        finally:
            qs.set_qsparse_options(fold_relu=True)
        active = step >= 1                                      # (step 0: the operators are not active yet, nothing is folded)
        # the composite route (prune -> quantize pairs, the lone Scaler quantizer) writes back in training and in no-grad forwards;
        # the fine-grained route where it records a gate, i.e. in training; the prune-only site never
        kernel_wrote = active and kind != "relu_p" and (quantizer == "scaler" or kind == "pair" or train)
        if active:
            assert n_owned == (0 if kernel_wrote else 1), (step, n_owned)
        got = ha.detach().cpu().contiguous()
        bits = torch.int16 if dtype == torch.bfloat16 else torch.int32
        if kernel_wrote:
            want = torch.relu_(x0.clone())                      # the reference's own in-place ReLU, on the CPU (-0.0 stays -0.0)
        else:
            want = hb.detach().cpu().contiguous()               # ATen's pass on the device (module by module: the same kernel)
        nan = want.isnan()                                      # (a NaN stays a NaN; its payload is ATen's business)
        assert torch.equal(got.isnan(), nan) and torch.equal(got.view(bits)[~nan], want.view(bits)[~nan]), step
        if step != 5:
            assert torch.equal(ya, yb), step
            if train:
                ga, = torch.autograd.grad(ya.sum(), xa)
                gb, = torch.autograd.grad(yb.sum(), xb)
                assert torch.equal(ga, gb), step
        else:       # NaN / Inf inputs: same values where the module-by-module route is finite, NaN where it is NaN
            a, b = ya.detach().float(), yb.detach().float()
            assert bool(((a == b) | (a.isnan() & b.isnan())).all()), step
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), plain.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
