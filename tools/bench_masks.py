#!/usr/bin/env python3
"""x * mask for every broadcast pattern class of qs_mask_apply (development tool, runs on the GPU box): channel mask,
full-shape mask and general broadcast patterns, bf16 in/out (4 B/elem + the mask bytes), on the BASELINE shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from qsparse_amd import _hip


def t_ms(fn, iters=20):
    for _ in range(3):
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
    for shape in ((256, 256, 56, 56), (256, 64, 56, 56)):
        N, C, H, W = shape
        x = torch.randn(shape, device="cuda").bfloat16()
        for name, mshape in (("channel (1,C,1,1)", (1, C, 1, 1)), ("full (N,C,H,W)", shape), ("(1,C,H,W)", (1, C, H, W)),
                             ("general (N,1,H,1)", (N, 1, H, 1)), ("general (1,C,1,W)", (1, C, 1, W)),
                             ("general (N,1,1,W)", (N, 1, 1, W)), ("general (N,1,H,W)", (N, 1, H, W))):
            m = torch.rand(mshape, device="cuda") > 0.5
            ms = t_ms(lambda: _hip.mask_apply(x, m))
            nbytes = x.numel() * 4 + m.numel()
            print(f"{str(shape):20s} {name:20s} {ms:7.4f} ms  {nbytes / ms / 1e6:7.0f} GB/s  ({nbytes / ms / 1e6 / 8000:.2f} of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
