#!/usr/bin/env python3
"""Statistics stages outside the headline shapes ("Contract of the staged mean's bits" ff.): wall time per call,
back to back, on an MI355X.  Three tables:

  1. a dense tensor in another dim order, reduced where it lies (qs_mean_strided / qs_mean_dim_split / the inner stage on the
     memory view) against the route it replaces -- a contiguous copy, then the NCHW kernels (the copy's summation order);
  2. the NCHW first stage [n, post] with and without the abs-max rider, float32 and bfloat16 (the dispatch of qs_mean_dim:
     one lane per output for float32 without the rider, row split with the rider only);
  3. a prune -> quantize site behind nn.ReLU / nn.ReLU6 / nn.LeakyReLU / nn.Hardtanh, kernel by kernel (the activation's kind is
     a compile-time mode of the statistics kernels).

usage (GPU box, repo root):  python3 tools/bench_stats_layouts.py  > gpurun_out/profiles/rNN_stats_layouts.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd import _hip
from qsparse_amd.fused import fuse_prune_quantize_pairs
from qsparse_amd.util import squeeze_tensor_to_shape

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def us(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def layouts():
    print("1. squeeze_tensor_to_shape of a permuted dense tensor, us per call (GB/s of the input)\n")
    for dtype in (torch.bfloat16, torch.float32):
        for shape, tr, target in [((256, 196, 768), (1, 2), (1, 768, 1)), ((64, 1024, 1024), (1, 2), (1, 1024, 1)), ((64, 1024, 1024), (1, 2), (64, 1024, 1)),
                                  ((4096, 4096), (0, 1), (4096, 1)), ((4096, 4096), (0, 1), (1, 4096))]:
            x = torch.randn(shape, device="cuda").to(dtype).transpose(*tr)
            a = us(lambda: squeeze_tensor_to_shape(x, list(target)))
            b = us(lambda: squeeze_tensor_to_shape(x.contiguous(), list(target)))
            nb = x.numel() * x.element_size()
            print(f"  {str(dtype)[6:]:9s} {tuple(x.shape)!s:20s} strides {tuple(x.stride())!s:22s} -> {target!s:14s} in place {a:8.1f} ({nb / a / 1e3:6.0f})"
                  f"   copy + NCHW kernels {b:8.1f}")
    print()


def first_stage():
    print("2. NCHW first stage, mean over n of |x| for x [n, C * hw], us per call (GB/s)\n")
    for dtype in (torch.bfloat16, torch.float32):
        for n, C, hw in [(256, 768, 196), (256, 1024, 196), (256, 2048, 196), (256, 4096, 196), (256, 512, 196), (256, 256, 196), (128, 256, 256), (64, 4096, 196),
                         (1024, 256, 256)]:
            post = C * hw
            x = torch.randn(n * post, device="cuda").to(dtype)
            acc = torch.zeros(C, 32, device="cuda")
            fl = _hip.mean_flags(True, False)
            a = us(lambda: _hip.mean_dim(x, 1, n, post, dtype, fl))
            b = us(lambda: _hip.mean_dim(x, 1, n, post, dtype, fl, absmax_out=acc, chan_div=hw, C=C))
            nb = x.numel() * x.element_size()
            print(f"  {str(dtype)[6:]:9s} [{n}, {C} x {hw}]".ljust(36) + f"plain {a:7.1f} ({nb / a / 1e3:5.0f})   with the abs-max rider {b:7.1f} ({nb / b / 1e3:5.0f})")
    print()


def activations():
    print("3. prune(0.75, {1}) -> quantize(4b) site on 256 x 256 x 56 x 56, training step, us per kernel (events around each launch)\n")
    for cl in (False, True):
        for dtype in (torch.bfloat16, torch.float32):
            x = torch.randn(256, 256, 56, 56, device="cuda").to(dtype)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
            x.requires_grad_(True)
            for name, act in (("relu", nn.ReLU()), ("relu6", nn.ReLU6()), ("leaky", nn.LeakyReLU(0.1)), ("hardtanh", nn.Hardtanh(-1.0, 1.0))):
                site = fuse_prune_quantize_pairs(nn.Sequential(
                    nn.Sequential(act, qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1)),
                    qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train())
                g = torch.randn(256, 256, 56, 56, device="cuda")
                if cl:
                    g = g.contiguous(memory_format=torch.channels_last)

                def step():
                    torch.autograd.grad(site(x), x, g)
                for _ in range(6):
                    step()
                _hip.start_event_log()
                for _ in range(5):
                    step()
                log = _hip.stop_event_log()
                per = {k: sum(v[-5:]) / 5 * 1e3 for k, v in log.items()}
                print(f"  {'channels_last' if cl else 'NCHW':13s} {str(dtype)[6:]:9s} {name:9s} " + "  ".join(f"{k} {v:6.1f}" for k, v in sorted(per.items())))
    print()


if __name__ == "__main__":
    print(f"{torch.cuda.get_device_name(0)}; tree {open('.tree_sha').read().strip() if os.path.exists('.tree_sha') else 'unknown'}\n")
    layouts()
    first_stage()
    activations()
