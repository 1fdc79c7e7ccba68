#!/usr/bin/env python3
"""Kernel timeline of one activation site's training step (development tool).
    rocprofv3 --kernel-trace -d out -- python3 tools/profile_site.py pair M|H|N,C,H,W [--graph] [--cl]      # run
    python3 tools/profile_site.py --timeline out/.../results.db [kernels-per-step]            # print the last steps
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeline(db, per_step):
    rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-3 * per_step:]
    prev = None
    for i, (n, s, e) in enumerate(rows):
        if i % per_step == 0:
            print("--- step")
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"gap {gap:7.1f} us  dur {(e - s) / 1e3:7.1f} us  {n[:100]}")
        prev = e
    print(f"span of the last step: {(rows[-1][2] - rows[-per_step][1]) / 1e3:.1f} us")


def main():
    if sys.argv[1] == "--timeline":
        return timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    import torch
    import torch.nn as nn

    import qsparse_amd as qs
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    kind = sys.argv[1]
    shape = {"M": (256, 64, 56, 56), "H": (256, 256, 56, 56)}.get(sys.argv[2]) or tuple(int(v) for v in sys.argv[2].split(","))
    channels_last = "--cl" in sys.argv
    graph = "--graph" in sys.argv
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, graph_safe=graph)
    if kind == "pair":
        site = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1, repetition=1)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1))
        fuse_prune_quantize_pairs(site)
    elif kind == "quantize":
        site = qs.quantize(bits=8, channelwise=-1, timeout=1)
    else:
        site = qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1, repetition=1)
    site = site.to("cuda").train()
    C, nbuf = shape[1], 4
    xs = [(torch.randn(shape, device="cuda") * torch.linspace(0.25, 4, C, device="cuda").view(1, C, 1, 1)).bfloat16() for _ in range(nbuf)]
    if channels_last:
        xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    xs = [x.requires_grad_(True) for x in xs]
    for i in range(20):
        site(xs[i % nbuf])
    g = torch.randn(shape, device="cuda", dtype=site(xs[0]).dtype)
    if channels_last:
        g = g.contiguous(memory_format=torch.channels_last)
    for i in range(20):
        torch.autograd.grad(site(xs[i % nbuf]), xs[i % nbuf], g)
    torch.cuda.synchronize()
    graphs = []
    if graph:
        for k in range(nbuf):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                torch.autograd.grad(site(xs[k]), xs[k], g)
            graphs.append(gr)
    for i in range(40):
        if graph:
            graphs[i % nbuf].replay()
        else:
            torch.autograd.grad(site(xs[i % nbuf]), xs[i % nbuf], g)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
