"""Resuming a training run from a checkpoint.

``state_dict()`` keeps the reference's schema, and with it the reference's quirk B7: ``QuantizeLayer._quantized`` and the
quantizer's running-mean count ``t`` are not in it (reference quantize.py:466,505 / :307,348).  ``qs.extra_state_dict`` /
``qs.load_extra_state_dict`` carry the two in a separate dict; with them a resumed run continues bit for bit like the
uninterrupted one, without them it behaves like the reference's resumed run (eval passes through, running scale restarts)."""
import copy

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def _net(quantizer):
    torch.manual_seed(0)
    cb = {"scaler": None, "decimal": qs.DecimalQuantizer(), "adaptive": qs.AdaptiveQuantizer()}[quantizer]
    net = nn.Sequential(nn.Linear(12, 16), nn.ReLU(), nn.Linear(16, 16), nn.ReLU(), nn.Linear(16, 4))
    net = qs.convert(net, qs.prune(sparsity=0.5, start=2, interval=2, repetition=2, dimensions={1}), activation_layers=[nn.ReLU],
                     log=False)
    return qs.convert(net, qs.quantize(bits=6, channelwise=-1, timeout=2, callback=cb), activation_layers=[nn.ReLU],
                      weight_layers=[nn.Linear], log=False)


def _trainable(net):
    return [p for p in net.parameters() if p.requires_grad]


def _steps(net, opt, xs):
    outs = []
    for x in xs:
        opt.zero_grad()
        y = net(x)
        y.square().mean().backward()
        opt.step()
        outs.append(y.detach().clone())
    return outs


def _resume_case(dev, quantizer):
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(8, 12, generator=g).to(dev) * (1 + i) for i in range(10)]
    # uninterrupted
    ref = _net(quantizer).to(dev).train()
    opt = torch.optim.SGD(_trainable(ref), lr=0.05, momentum=0.9)
    _steps(ref, opt, xs[:6])
    ckpt = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    extra = qs.extra_state_dict(ref)
    opt_state = copy.deepcopy(opt.state_dict())
    want = _steps(ref, opt, xs[6:])
    want_state = ref.state_dict()
    assert len(extra) == 2 + 3                                # one record per QuantizeLayer: two activation sites, three weights
    assert all(set(v) == {"quantized", "t"} for v in extra.values()) and any(v["t"] > 0 for v in extra.values())

    def load():
        net = _net(quantizer).to(dev)
        qs.preload_qsparse_state_dict(net, ckpt)
        net.load_state_dict(ckpt)
        o = torch.optim.SGD(_trainable(net), lr=0.05, momentum=0.9)
        o.load_state_dict(copy.deepcopy(opt_state))
        return net, o

    # with the extra state: bit for bit the uninterrupted run, and evaluation quantizes right after loading
    net, o = load()
    qs.load_extra_state_dict(net, extra)
    ref_eval = _net(quantizer).to(dev)
    qs.preload_qsparse_state_dict(ref_eval, ckpt)
    ref_eval.load_state_dict(ckpt)
    with torch.no_grad():
        passthrough = ref_eval.eval()(xs[6])                  # quirk B7: nothing quantizes yet
        quantizing = net.eval()(xs[6])
    assert not torch.equal(passthrough, quantizing)
    got = _steps(net.train(), o, xs[6:])
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    for k, v in want_state.items():
        assert torch.equal(v, net.state_dict()[k]), k
    assert qs.extra_state_dict(net) == qs.extra_state_dict(ref)
    # without it: the reference's resumed run -- the running scale restarts (t = 0), so the trajectories part
    net, o = load()
    got = _steps(net.train(), o, xs[6:])
    assert not all(torch.equal(a, b) for a, b in zip(want, got))
    # a DataParallel / DDP style wrapper on either side changes nothing (paths are those of the unwrapped network)
    class Wrapper(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module
    assert qs.extra_state_dict(Wrapper(ref)) == qs.extra_state_dict(ref)
    qs.load_extra_state_dict(Wrapper(net), qs.extra_state_dict(ref))
    # strictness
    with pytest.raises(KeyError):
        qs.load_extra_state_dict(net, {"nope": {"quantized": 1, "t": 1}})
    qs.load_extra_state_dict(net, {}, strict=False)


@pytest.mark.parametrize("quantizer", ["scaler", "decimal", "adaptive"])
def test_resume_on_the_cpu(quantizer):
    _resume_case("cpu", quantizer)


@pytest.mark.gpu
@pytest.mark.parametrize("quantizer", ["scaler", "decimal", "adaptive"])
def test_resume_on_the_gpu(quantizer):
    _resume_case("cuda", quantizer)
