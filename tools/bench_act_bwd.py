#!/usr/bin/env python3
"""The site backward with the caller's GELU (ste_relu_bwd_kernel<..., DACT>) against the two passes it replaces: the gated site
backward + ATen's gelu_backward.  usage (GPU box): python3 tools/bench_act_bwd.py [B*T=25088] [C=3072]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qsparse_amd import _hip

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 196
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
dev = "cuda"
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(rows, C, device=dev).to(dt)
    g2 = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    g = torch.randn(rows, C, device=dev)
    mask = (torch.rand(C, device=dev) > 0.75)
    inf = float("inf")

    def t(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    n = rows * C
    for name, gg, g2g in (("g2 alone", None, g2), ("fp32 g", g, None)):
        fused = t(lambda: _hip.ste_act_bwd(gg, x, 0.25, False, -8.0, 7.0, chan_mask=mask, mask_channel_index=1, g2=g2g))
        gh = _hip.ste_bwd(gg if gg is not None else g2g.float(), 0.25, False, -1, -8.0, 7.0, False, dt, chan_mask=mask, mask_channel_index=1)
        aten = t(lambda: torch.ops.aten.gelu_backward(gh, x))
        site = t(lambda: _hip.ste_bwd(gg if gg is not None else g2g, 0.25, False, -1, -8.0, 7.0, False, dt, chan_mask=mask, mask_channel_index=1)) if not (gg is None and dt != torch.bfloat16) else float("nan")
        byt = n * ((4 if gg is not None else 2) + 2 * x.element_size())
        print(f"{dt} {rows} x {C}, {name}: fused {fused:.1f} us ({byt / fused / 1e6:.2f} TB/s), ATen gelu_backward {aten:.1f} us, site backward alone {site:.1f} us")
