#!/usr/bin/env python3
"""Per-operator microbenchmarks through the public Python API (development tool): GB/s of algorithmic
traffic and fraction of the 8 TB/s roofline for every element-wise / statistics kernel family."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import qsparse_amd as qs
from qsparse_amd import _hip
from qsparse_amd.quantize import quantize_with_decimal, quantize_with_line, quantize_with_scaler
from qsparse_amd.sparse import apply_mask
from qsparse_amd.util import squeeze_tensor_to_shape

dev = "cuda"
PEAK = 8000.0


def t_ms(fn, iters=15, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def report(name, fn, nbytes):
    ms = t_ms(fn)
    print(f"{name:58s} {ms*1e3:9.1f} us {nbytes/ms/1e6:8.0f} GB/s  {nbytes/ms/1e6/PEAK:5.2f}", flush=True)


def main():
    shape = (256, 64, 56, 56)      # config 2 of BASELINE.json (51.4 M elements)
    big = (256, 256, 56, 56)
    for shp in (shape, big):
        n = 1
        for s in shp:
            n *= s
        C = shp[1]
        print(f"--- shape {shp} ({n/1e6:.1f} M elements)")
        xb = torch.randn(shp, device=dev).bfloat16()
        s1 = torch.tensor([[0.05]], device=dev)
        sc = torch.rand(C, 1, device=dev) * 0.1 + 0.01
        d1, dc = torch.tensor([[5.0]], device=dev), torch.randint(0, 8, (C, 1), device=dev).float()
        l1 = torch.tensor([[-1.0, 2.0]], device=dev)
        lc = torch.cat([-torch.rand(C, 1, device=dev), torch.rand(C, 1, device=dev) + 0.1], 1)
        g = torch.randn(shp, device=dev)
        mask_c = (torch.rand(1, C, 1, 1, device=dev) > 0.5)
        mask_full = torch.rand(shp, device=dev) > 0.5
        mask_odd = torch.rand(shp[0], 1, shp[2], 1, device=dev) > 0.5
        report("scaler fwd bf16->f32 tensor-wise", lambda: quantize_with_scaler(xb, 8, s1), 6 * n)
        report("scaler fwd bf16->f32 per-channel", lambda: quantize_with_scaler(xb, 8, sc, 1), 6 * n)
        report("decimal fwd bf16->f32 tensor-wise", lambda: quantize_with_decimal(xb, 8, d1), 6 * n)
        report("decimal fwd bf16->f32 per-channel", lambda: quantize_with_decimal(xb, 8, dc, 1), 6 * n)
        report("line fwd (train form) bf16->f32 tensor-wise", lambda: quantize_with_line(xb, 8, l1, -1, False, True), 6 * n)
        report("line fwd (train form) bf16->f32 per-channel", lambda: quantize_with_line(xb, 8, lc, 1, False, True), 6 * n)
        report("line fwd (eval form) bf16->f32 per-channel", lambda: quantize_with_line(xb, 8, lc, 1, False, False), 6 * n)
        report("ste bwd f32->bf16 tensor-wise", lambda: _hip.ste_bwd(g, s1, False, -1, -128.0, 127.0, False, torch.bfloat16), 6 * n)
        report("ste bwd f32->bf16 per-channel", lambda: _hip.ste_bwd(g, sc, False, 1, -128.0, 127.0, False, torch.bfloat16), 6 * n)
        xf = g
        report("scaler fwd f32->f32 tensor-wise", lambda: quantize_with_scaler(xf, 8, s1), 8 * n)
        report("scaler fwd f32->f32 per-channel", lambda: quantize_with_scaler(xf, 8, sc, 1), 8 * n)
        report("ste bwd f32->f32 per-channel", lambda: _hip.ste_bwd(g, sc, False, 1, -128.0, 127.0, False, torch.float32), 8 * n)
        report("mask apply bf16 channel mask", lambda: _hip.mask_apply(xb, mask_c), 4 * n)
        report("mask apply bf16 full-shape mask", lambda: _hip.mask_apply(xb, mask_full), 5 * n)
        report("mask apply bf16 general broadcast (N,1,H,1)", lambda: _hip.mask_apply(xb, mask_odd), 4 * n)
        report("abs-max tensor-wise bf16", lambda: _hip.absmax(xb, -1), 2 * n)
        report("abs-max per-channel bf16", lambda: _hip.absmax(xb, 1), 2 * n)
        report("min/max per-channel bf16 (Adaptive)", lambda: _hip.minmax(xb, 1), 2 * n)
        report("staged mean x -> (1,C,1,1) bf16 (squeeze_tensor_to_shape)", lambda: squeeze_tensor_to_shape(xb, (1, C, 1, 1)), 2 * n)
        del xb, g, mask_full
        torch.cuda.empty_cache()
    # ragged inner (7x7) and 2-d
    x7 = torch.randn(256, 2048, 7, 7, device=dev).bfloat16()
    n = x7.numel()
    m7 = torch.rand(1, 2048, 1, 1, device=dev) > 0.5
    sc7 = torch.rand(2048, 1, device=dev) * 0.1 + 0.01
    print(f"--- shape {tuple(x7.shape)} ({n/1e6:.1f} M elements, inner=49)")
    s7 = torch.tensor([[0.05]], device=dev)
    report("scaler fwd bf16->f32 tensor-wise", lambda: quantize_with_scaler(x7, 8, s7), 6 * n)
    report("scaler fwd bf16->f32 per-channel (element-wise channel walk)", lambda: quantize_with_scaler(x7, 8, sc7, 1), 6 * n)
    report("mask apply bf16 channel mask (element-wise channel walk)", lambda: _hip.mask_apply(x7, m7), 4 * n)
    report("abs-max per-channel bf16 (rows of 49)", lambda: _hip.absmax(x7, 1), 2 * n)
    report("staged mean |x| -> (1,C,1,1) bf16", lambda: squeeze_tensor_to_shape(x7, (1, 2048, 1, 1)), 2 * n)


if __name__ == "__main__":
    main()
