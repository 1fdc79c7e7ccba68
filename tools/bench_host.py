#!/usr/bin/env python3
"""Host-side (Python + launch) cost per operator site and training step, measured as wall time per iteration on tensors so
small that the GPU is never the bottleneck (development tool).  This is what an eager training loop pays per site."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd.fused import fuse_prune_quantize_pairs

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
dev = "cuda"


def wall_us(fn, iters=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


x = torch.randn(4, 16, 8, 8, device=dev, dtype=torch.bfloat16, requires_grad=True)
g32 = torch.randn(4, 16, 8, 8, device=dev)
g16 = g32.bfloat16()
rows = []
pair = fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                                               qs.quantize(bits=4, channelwise=-1, timeout=1)).to(dev).train())
rows.append(("ReLU->prune->quantize pair, forward", wall_us(lambda: pair(x))))
rows.append(("ReLU->prune->quantize pair, forward + backward", wall_us(lambda: torch.autograd.grad(pair(x), x, g32))))
site = fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=1)).to(dev).train())
rows.append(("ReLU->quantize site, forward + backward", wall_us(lambda: torch.autograd.grad(site(x), x, g32))))
p = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1).to(dev).train()
rows.append(("PruneLayer alone, forward + backward", wall_us(lambda: torch.autograd.grad(p(x), x, g16))))
conv = qs.quantize(nn.Conv2d(16, 16, 3, padding=1), bits=4, channelwise=-1, timeout=1).to(dev).train()
xc = torch.randn(4, 16, 8, 8, device=dev)
plain = nn.Conv2d(16, 16, 3, padding=1).to(dev).train()
rows.append(("plain Conv2d forward + backward (baseline)", wall_us(lambda: plain(xc).sum().backward())))
rows.append(("quantize(Conv2d) forward + backward", wall_us(lambda: conv(xc).sum().backward())))
relu = nn.ReLU()
rows.append(("plain ReLU forward + backward (baseline)", wall_us(lambda: torch.autograd.grad(relu(x), x, g16))))
for name, us in rows:
    print(f"{name:52s} {us:7.1f} us/iter")
