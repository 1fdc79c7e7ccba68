import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import qsparse_amd as qs
from qsparse_amd import fused
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
def site():
    net = nn.Sequential(nn.ReLU())
    net = qs.convert(net, qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1), activation_layers=[nn.ReLU], log=False)
    return qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1), activation_layers=[nn.ReLU], log=False)
pre, s1 = site().cuda().train(), site().cuda().train()
conv = nn.Conv2d(16, 16, 1, bias=False).cuda()
conv2 = nn.Conv2d(16, 8, 1, bias=False).cuda()
orig = fused._PromotingAdd.backward
def spy(ctx, g):
    c = ctx.cell
    print("add bwd: cell keys", list(c), "g is gx", c.get("gx") is g, "same ptr", c.get("gx") is not None and c["gx"].data_ptr() == g.data_ptr(), type(g).__name__)
    return orig(ctx, g)
fused._PromotingAdd.backward = staticmethod(spy)
for step in range(4):
    x = torch.randn(4, 16, 8, 8, device="cuda").bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = pre(x)
        h = conv(y0) + y0
        print(step, type(y0).__name__, y0.dtype, type(h).__name__, h.dtype, fused._GRAD_IMAGE_CELL in h.__dict__)
        y = s1(h)
        out = conv2(y).float().sum()
    out.backward()
print(dict(fused.ROUTES))
print("---- debug")
ob = fused._SiteStep.backward
def spy2(ctx, *gs):
    print("site bwd: has_gate", ctx.has_gate, "x_dtype", ctx.x_dtype, "cell", ctx.cell, "grads", [None if g is None else (g.dtype, g.is_contiguous()) for g in gs])
    return ob(ctx, *gs)
fused._SiteStep.backward = staticmethod(spy2)
of = fused._SiteStep.forward
x = torch.randn(4, 16, 8, 8, device="cuda").bfloat16().requires_grad_(True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    y0 = pre(x)
    h = conv(y0) + y0
    print("cell via fn", fused.grad_image_cell(h), type(s1[0]).__name__, type(s1[0]).__mro__[0])
    y = s1(h)
    print(type(y), y.grad_fn)
    out = conv2(y).float().sum()
out.backward()
