"""torch.inference_mode(): tensors created there carry no version counter (``Tensor._version`` raises), which the host mirrors
of the step counters, the weight batcher and the autocast image use to notice writes.  The reference has no such
machinery and simply works; so must this package -- whether the network's state was created outside inference mode (the
usual case: train, then serve) or inside it (first forward / preload under ``torch.inference_mode()``)."""
import copy

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd.common import HostMirror

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def _net():
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(), nn.Flatten(),
                        nn.Linear(8 * 6 * 6, 5))
    net = qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=1, dimensions={1}), activation_layers=[nn.ReLU],
                     log=False)
    return qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1), activation_layers=[nn.ReLU],
                      weight_layers=[nn.Conv2d, nn.Linear], log=False)


def _run(dev, autocast=False):
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, 3, 6, 6, generator=g).to(dev) for _ in range(6)]
    trained = _net().to(dev).train()
    for x in xs[:4]:
        trained(x).sum().backward()
    trained.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        want = [trained(x) for x in xs[4:]]
    # (1) state created outside, served inside inference mode
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        got = [trained(x) for x in xs[4:]]
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    # (2) state created INSIDE inference mode: a fresh network's first (training-mode) forwards run there
    fresh, twin = _net().to(dev).train(), _net().to(dev).train()
    with torch.inference_mode():
        a = [fresh(x) for x in xs[:4]]
    with torch.no_grad():
        b = [twin(x) for x in xs[:4]]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    sa, sb = fresh.state_dict(), twin.state_dict()
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    fresh.eval(), twin.eval()
    with torch.inference_mode():
        u = fresh(xs[5])
    with torch.no_grad():
        v = twin(xs[5])
    assert torch.equal(u, v)


def test_inference_mode_on_the_cpu_with_tracked_counters():
    HostMirror.track_cpu = True
    try:
        _run("cpu")
    finally:
        HostMirror.track_cpu = False


@pytest.mark.gpu
@pytest.mark.parametrize("image", [False, True])
def test_inference_mode_on_the_gpu(image):
    qs.set_qsparse_options(autocast_image=image)
    try:
        _run("cuda", autocast=image)
    finally:
        qs.set_qsparse_options(autocast_image=True)       # (the default)
