#!/usr/bin/env python3
"""Mask-aware traffic elision, measured (runs on the GPU box): the fused apply kernels through the C ABI with
elide_masked = 0 / 1 at several channel densities and mask patterns, NCHW and channels_last.

    python tools/bench_elide.py [--shape 256 256 56 56] [--json out.json]

Per row: kernel, layout, kept fraction, pattern, ms with and without elision, algorithmic bytes of each and the
resulting GB/s -- "algorithmic" for the elided kernel = kept fraction x read + full write.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from qsparse_amd import _hip


def time_ms(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=4, default=[256, 256, 56, 56])
    ap.add_argument("--json", default=None)
    ap.add_argument("--lib", default=None, help="variant library to time instead of the in-tree one")
    ap.add_argument("--only", default=None, help="layout filter: nchw | nhwc")
    args = ap.parse_args()
    lib = _hip.load(args.lib)
    dev = "cuda"
    N, C, H, W = args.shape
    numel = N * C * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.randn(args.shape, generator=g, device=dev).relu_() * torch.linspace(0.25, 4, C, device=dev).view(1, C, 1, 1)).bfloat16()
    gout = torch.randn(args.shape, generator=g, device=dev)
    y = torch.empty(args.shape, device=dev)
    gx = torch.empty(args.shape, device=dev, dtype=torch.bfloat16)
    scale = torch.tensor([0.37], device=dev)
    rows = []
    for layout in ("nchw", "nhwc"):
        if args.only and layout != args.only:
            continue
        outer, Cc, inner = (N, C, H * W) if layout == "nchw" else (N * H * W, C, 1)
        for keep, pattern in ((1.0, "all"), (0.5, "strided"), (0.25, "strided"), (0.25, "random"), (0.25, "block"), (0.0, "none")):
            if pattern == "all":
                mask = torch.ones(C, device=dev, dtype=torch.uint8)
            elif pattern == "none":
                mask = torch.zeros(C, device=dev, dtype=torch.uint8)
            elif pattern == "strided":
                mask = (torch.arange(C, device=dev) % round(1 / keep) == 0).to(torch.uint8)
            elif pattern == "block":
                mask = (torch.arange(C, device=dev) < int(C * keep)).to(torch.uint8)
            else:
                mask = (torch.rand(C, generator=g, device=dev) < keep).to(torch.uint8)
            kept = float(mask.float().mean())
            for kernel in ("fwd", "bwd"):
                res = {}
                for elide in (0, 1):
                    if kernel == "fwd":
                        def fn():
                            assert lib.qs_quant_scaler_fwd(x.data_ptr(), y.data_ptr(), None, scale.data_ptr(), 1, 0.0, mask.data_ptr(),
                                                           outer, Cc, inner, 1, 0, 0, 0, 0, 0, 0, elide, None, None, 0, None, None) == 0
                        rd, wr = 2, 4
                    else:
                        def fn():
                            assert lib.qs_quant_ste_bwd(gout.data_ptr(), gx.data_ptr(), scale.data_ptr(), 1, 0.0, 0, -8.0, 7.0, 0,
                                                        mask.data_ptr(), outer, Cc, inner, 0, 1, elide, None) == 0
                        rd, wr = 4, 2
                    ms = time_ms(fn)
                    bytes_ = numel * ((kept if elide else 1.0) * rd + wr)
                    res[elide] = dict(ms=round(ms, 4), GBps=round(bytes_ / ms / 1e6, 1), bytes=int(bytes_))
                row = dict(kernel=kernel, layout=layout, kept=round(kept, 4), pattern=pattern, dense=res[0], elided=res[1])
                rows.append(row)
                print(f"{kernel} {layout} kept={kept:.3f} {pattern:8s} dense {res[0]['ms']:.4f} ms {res[0]['GBps']:7.1f} GB/s | "
                      f"elided {res[1]['ms']:.4f} ms {res[1]['GBps']:7.1f} GB/s (alg.)  x{res[0]['ms'] / res[1]['ms']:.2f}", flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(shape=args.shape, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
