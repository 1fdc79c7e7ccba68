#!/usr/bin/env python3
"""development probe: how does the kernel that ran before it change the statistics kernel's duration?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qsparse_amd import _hip

lib = _hip.load()
ELIDE = int(os.environ.get("QS_PROBE_ELIDE", "0"))
dev = "cuda"
N, C, H, W = 256, 256, 56, 56
numel = N * C * H * W
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn((N, C, H, W), generator=g, device=dev).relu_() * torch.linspace(0.25, 4, C, device=dev).view(1, C, 1, 1)).bfloat16()
gout = torch.randn((N, C, H, W), generator=g, device=dev)
y = torch.empty((N, C, H, W), device=dev)
gx = torch.empty((N, C, H, W), device=dev, dtype=torch.bfloat16)
mask = (torch.arange(C, device=dev) % 4 == 0).to(torch.uint8)
scale = torch.tensor([0.37], device=dev)
stage = torch.empty(C * H * W, device=dev, dtype=torch.bfloat16)
amax = torch.zeros(C, device=dev)
big = torch.empty(300 * 1024 * 1024, device=dev, dtype=torch.uint8)

def fwd(): lib.qs_quant_scaler_fwd(x.data_ptr(), y.data_ptr(), None, scale.data_ptr(), 1, 0.0, mask.data_ptr(), N, C, H * W, 1, 0, 0, 0, 0, 0, 0, ELIDE, None, None, 0, None, None)
def bwd(): lib.qs_quant_ste_bwd(gout.data_ptr(), gx.data_ptr(), scale.data_ptr(), 1, 0.0, 0, -8.0, 7.0, 0, mask.data_ptr(), N, C, H * W, 0, 1, 0, None)
def stats(): lib.qs_mean_dim(x.data_ptr(), stage.data_ptr(), 1, N, C * H * W, 1, 1, 1, None, amax.data_ptr(), 1, H * W, C, None)
def readg(): lib.qs_absmax(gout.data_ptr(), amax.data_ptr(), 0, 1, 1, numel, 0, 0, 0, 1, None, 0, None)
def fill(): big.zero_()

def timed(pre, fn, iters=15):
    ts = []
    for _ in range(iters):
        for p in pre: p()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

if os.environ.get("QS_PROBE_SHORT"):
    a = timed([bwd], stats); b = timed([stats], fwd); c = timed([fwd], bwd)
    print(f"{os.environ.get('QSPARSE_HIP_LIB','default')[-16:]}: stats|bwd {a*1e3:6.1f}  fwd|stats {b*1e3:6.1f}  bwd|fwd {c*1e3:6.1f}  sum {(a+b+c)*1e3:6.1f} us", flush=True)
    sys.exit(0)
for name, pre in (("stats after stats", [stats]), ("stats after fwd", [fwd]), ("stats after bwd", [bwd]),
                  ("stats after fwd,bwd", [fwd, bwd]), ("stats after read-only pass over g", [readg]),
                  ("stats after 300MB memset", [fill])):
    print(f"{name:40s} {timed(pre, stats)*1e3:7.1f} us", flush=True)
for name, pre in (("fwd after stats", [stats]), ("fwd after bwd", [bwd]), ("fwd after fwd", [fwd])):
    print(f"{name:40s} {timed(pre, fwd)*1e3:7.1f} us", flush=True)
for name, pre in (("bwd after fwd", [fwd]), ("bwd after bwd", [bwd]), ("bwd after stats", [stats])):
    print(f"{name:40s} {timed(pre, bwd)*1e3:7.1f} us", flush=True)
