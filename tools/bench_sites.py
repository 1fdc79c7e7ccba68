#!/usr/bin/env python3
"""Training-step time (fwd + bwd, live statistics) of single activation sites at the BASELINE shapes, through the
public module API (development tool):
  config 2   QuantizeLayer(bits=8, tensor-wise) alone on 256x64x56x56 bf16                 (14 B/elem)
  relu+q     convert-built Sequential(ReLU, QuantizeLayer), folded vs module by module    (16 vs 24 B/elem)
  pair       the headline prune->quantize pair (what bench.py reports)                      (14 B/elem)
Buffers rotate so that the 256 MiB Infinity Cache cannot hold the working set of the small shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd.fused import fuse_prune_quantize_pairs

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEV = "cuda"
GRAPH = "--graph" in sys.argv
if GRAPH:
    qs.set_qsparse_options(graph_safe=True)


def run(site, shape, nbuf, bytes_per_elem, label, steps=200):
    C = shape[1]
    xs = [(torch.randn(shape, device=DEV) * torch.linspace(0.25, 4, C, device=DEV).view(1, C, 1, 1)).bfloat16().requires_grad_(True)
          for _ in range(nbuf)]
    for i in range(50):          # reach the steady state first: an inactive quantizer still returns x's dtype
        site(xs[i % nbuf])
    g = torch.randn(shape, device=DEV, dtype=site(xs[0]).dtype)
    for i in range(50):
        torch.autograd.grad(site(xs[i % nbuf]), xs[i % nbuf], g)
    torch.cuda.synchronize()
    graph = None
    if GRAPH:      # ONE captured graph of nbuf steps (a hipGraphLaunch costs more than a 3-kernel step): what the kernels
        #            alone take, without the Python between them
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for k in range(nbuf):
                torch.autograd.grad(site(xs[k]), xs[k], g)
        graph.replay()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if graph:
        for i in range(steps // nbuf):
            graph.replay()
        steps = (steps // nbuf) * nbuf
    else:
        for i in range(steps):
            torch.autograd.grad(site(xs[i % nbuf]), xs[i % nbuf], g)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    n = xs[0].numel()
    print(f"{label:58s} {str(shape):20s} {ms:7.4f} ms/step {n / ms / 1e6:7.1f} Gelem/s  "
          f"{bytes_per_elem} B/elem -> {bytes_per_elem * n / ms / 1e6:6.0f} GB/s ({bytes_per_elem * n / ms / 1e6 / 8000:.2f} of 8 TB/s)", flush=True)


def main():
    M, H = (256, 64, 56, 56), (256, 256, 56, 56)
    for shape, nbuf in ((M, 4), (H, 1)):
        q = qs.quantize(bits=8, channelwise=-1, timeout=1).to(DEV).train()
        run(q, shape, nbuf, 14, "config 2: QuantizeLayer(bits=8) alone")
        for fold in (True, False):
            qs.set_qsparse_options(fold_relu=fold)
            site = fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.quantize(bits=8, channelwise=-1, timeout=1))).to(DEV).train()
            run(site, shape, nbuf, 16 if fold else 24, f"ReLU -> QuantizeLayer(bits=8), fold_relu={fold}")
        qs.set_qsparse_options(fold_relu=True)
        pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1, repetition=1)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
        fuse_prune_quantize_pairs(pair)
        run(pair, shape, nbuf, 14, "pair: prune(0.75,{1}) -> quantize(4)")
        p = qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1, repetition=1).to(DEV).train()
        run(p, shape, nbuf, 10, "PruneLayer(0.75,{1}) alone (bf16 in/out)")
        for fold in (True, False):
            qs.set_qsparse_options(fold_relu=fold)
            site = fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1,
                                                                               repetition=1))).to(DEV).train()
            run(site, shape, nbuf, 12 if fold else 20, f"ReLU -> PruneLayer(0.75,{{1}}), fold_relu={fold}")
        qs.set_qsparse_options(fold_relu=True)


if __name__ == "__main__":
    main()
