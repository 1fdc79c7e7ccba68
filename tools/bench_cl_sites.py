#!/usr/bin/env python3
"""The activation sites of BASELINE config 4 one by one: ReLU -> prune(0.75, {1}) -> quantize(4) as `convert` builds and
fuses them, channels_last, batch 256, in the dtype autocast hands each site (bf16 behind a convolution's batch norm, fp32
behind a residual add), steady state.  Per library kernel: median time (HIP events), the bytes it must move, the share of
the 8 TB/s roofline.  Development tool; `--nchw` runs the same sites in NCHW order, `--no-gate` switches the ReLU gate
bitmap off (the backward then reads the ReLU's input again)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd import _hip
from qsparse_amd.fused import fuse_prune_quantize_pairs

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEV = "cuda"
CL = "--nchw" not in sys.argv
GATE = "--no-gate" not in sys.argv
qs.set_qsparse_options(relu_gate=GATE)
BATCH = 256

# (C, H, W, dtype, how many such sites a ResNet-50 step has)
SITES = [(64, 112, 112, torch.bfloat16, 1), (64, 56, 56, torch.bfloat16, 6), (256, 56, 56, torch.float32, 3),
         (128, 56, 56, torch.bfloat16, 1), (128, 28, 28, torch.bfloat16, 7), (512, 28, 28, torch.float32, 4),
         (256, 28, 28, torch.bfloat16, 1), (256, 14, 14, torch.bfloat16, 11), (1024, 14, 14, torch.float32, 6),
         (512, 14, 14, torch.bfloat16, 1), (512, 7, 7, torch.bfloat16, 5), (2048, 7, 7, torch.float32, 2)]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    total = {}
    ideal = 0.0
    for C, H, W, dt, count in SITES:
        shape = (BATCH, C, H, W)
        n = BATCH * C * H * W
        eb = 2 if dt == torch.bfloat16 else 4
        nbuf = max(1, min(4, int(8e8 // (n * eb))))
        xs = [(torch.randn(shape, device=DEV) * torch.linspace(0.25, 4, C, device=DEV).view(1, C, 1, 1)).to(dt) for _ in range(nbuf)]
        if CL:
            xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
        xs = [x.requires_grad_(True) for x in xs]
        site = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={1}, start=1, interval=1, repetition=1)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1)).to(DEV).train()
        site = fuse_prune_quantize_pairs(site)
        for i in range(8):
            y = site(xs[i % nbuf])
        g = torch.randn(shape, device=DEV, dtype=y.dtype)
        if CL:
            g = g.contiguous(memory_format=torch.channels_last)
        for i in range(6):
            torch.autograd.grad(site(xs[i % nbuf]), xs[i % nbuf], g)
        torch.cuda.synchronize()
        _hip.start_event_log()
        for i in range(24):
            torch.autograd.grad(site(xs[i % nbuf]), xs[i % nbuf], g)
        log = _hip.stop_event_log()
        # bytes: statistics read x; forward read x, write fp32; backward read g (fp32) and x, write gx in x's dtype
        # (with the gate bitmap: +1/8 B/elem written by the forward, read by the backward instead of x)
        need = {"mean_dim+absmax": eb, "quant_scaler_fwd+mask": eb + 4 + (0.125 if GATE else 0),
                "quant_ste_relu_bwd": 4 + eb + (0.125 if GATE else eb)}
        cells = []
        site_us = 0.0
        for name, times in log.items():
            us = med(times) * 1e3
            site_us += us
            total[name] = total.get(name, 0.0) + us * count
            if name in need:
                cells.append(f"{name} {us:7.1f} us {need[name] * n / us / 1e3 / 8000:5.2f}")
            else:
                cells.append(f"{name} {us:5.1f} us")
        site_ideal = sum(need.values()) * n / 6.29e6
        ideal += site_ideal * count
        print(f"{str(shape):22s} {str(dt)[6:]:8s} x{count:<2d} site {site_us:7.1f} us (copy-ceiling {site_ideal:7.1f})  " + " | ".join(cells),
              flush=True)
    print("\nper ResNet-50 step (sum over sites x count), ms:")
    for k, v in sorted(total.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v / 1e3:7.3f}")
    print(f"  {'total':28s} {sum(total.values()) / 1e3:7.3f}   (streaming bytes at the 6.29 TB/s copy ceiling: {ideal / 1e3:.3f})")


if __name__ == "__main__":
    main()
