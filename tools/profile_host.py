#!/usr/bin/env python3
"""host-side (Python) cost of one weight prune+quantize step, by cProfile (development tool)"""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import qsparse_amd as qs
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
dev = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "weight"
if mode == "weight":
    conv = nn.Conv2d(256, 256, 3).to(dev)
    conv = qs.quantize(qs.prune(conv, sparsity=0.5, dimensions={0, 1, 2, 3}, start=0, interval=1, repetition=1,
                                callback=qs.MagnitudePruningCallback(running_average=False)), bits=4, timeout=1, channelwise=0)
    conv.train()
    def step():
        conv.weight.sum().backward()
elif mode == "conv":     # a weight layer as the --pq recipe converts it: tensor-wise 4-bit weight quantizer, nothing else
    conv = qs.quantize(nn.Conv2d(64, 64, 3, padding=1, bias=False), bits=4, channelwise=-1, timeout=1).to(dev).train()
    xin = torch.randn(4, 64, 8, 8, device=dev)
    def step():
        conv(xin).sum().backward()
elif mode == "plainconv":
    conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).train()
    xin = torch.randn(4, 64, 8, 8, device=dev)
    def step():
        conv(xin).sum().backward()
elif mode == "relupair":
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1)),
                         qs.quantize(bits=4, channelwise=-1, timeout=1)).to(dev).train()
    pair = fuse_prune_quantize_pairs(pair)
    x = torch.randn(8, 64, 16, 16, device=dev, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(8, 64, 16, 16, device=dev)
    def step():
        torch.autograd.grad(pair(x), x, g)
else:
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1)),
                         qs.quantize(bits=4, channelwise=-1, timeout=1)).to(dev).train()
    fuse_prune_quantize_pairs(pair)
    x = torch.randn(8, 64, 16, 16, device=dev, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(8, 64, 16, 16, device=dev)
    def step():
        torch.autograd.grad(pair(x), x, g)
for _ in range(10):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(500):
    step()
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per step (wall, no profiler)")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
print(s.getvalue()[:7000])
