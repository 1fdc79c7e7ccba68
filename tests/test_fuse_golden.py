"""fuse_bn against fixtures recorded from the reference (qsparse/fuse.py:76-163): folded parameters, the
rewritten module tree and the fused network's output, bit for bit.  CPU only."""
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import Golden, same


def _nets():
    return {
        "conv": nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5)),
        "linear": nn.Sequential(nn.Linear(12, 7, bias=False), nn.BatchNorm1d(7)),
        "deconv": nn.Sequential(nn.ConvTranspose2d(3, 5, 3), nn.BatchNorm2d(5)),
        "nested": nn.Sequential(nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)), nn.ReLU(),
                                nn.Sequential(nn.Conv2d(4, 4, 3)), nn.BatchNorm2d(4), nn.ReLU(),
                                nn.Sequential(nn.BatchNorm2d(4), nn.ConvTranspose2d(4, 2, 3), nn.BatchNorm2d(2))),
    }


def test_f11_fuse_bn():
    g = Golden("f11_fuse_bn")
    nets = _nets()
    for c in g.cases:
        name, net = c["name"], nets[c["name"]]
        net.load_state_dict({k: torch.as_tensor(g.get(f"{name}_in_{k}")) for k in net.state_dict()})
        net.eval()
        x = g.get(name + "_x")
        before = net(x)
        fused = qs.fuse_bn(net, log=False)
        assert str(fused) == c["tree"], name
        sd = fused.state_dict()
        assert list(sd.keys()) == c["out_keys"], name
        for k, v in sd.items():
            assert same(v, torch.as_tensor(g.get(f"{name}_out_{k}"))), (name, k)
        y = fused(x)
        assert same(y.detach(), g.get(name + "_y")), name
        assert torch.allclose(y, before, atol=1e-5)   # the reference's own criterion (tests/test_fuse.py)


def test_fuse_bn_options():
    net = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Linear(4, 4), nn.BatchNorm1d(4))
    out = qs.fuse_bn(net, layers=["Conv2d"], log=False, inplace=False)
    assert "BatchNorm2d" in str(net) and "BatchNorm2d" not in str(out) and "BatchNorm1d" in str(out)
    dp = nn.DataParallel(nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)))
    assert "batchnorm" not in str(qs.fuse_bn(dp, log=False)).lower()

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.ReLU())
            self.head = nn.Linear(4, 2)

    assert "batchnorm" not in str(qs.fuse_bn(Net(), log=False)).lower()
