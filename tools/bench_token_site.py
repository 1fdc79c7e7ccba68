#!/usr/bin/env python3
"""A token-major prune(dimensions={2}) -> quantize(4b) site (qs_site_plan layout 3) on ViT-sized activations: us per kernel of a
training step (HIP events around each launch of the fine-grained route; the composite issues the same launches) and the fraction
of the 8 TB/s roofline for the streaming ones.  usage (GPU box): python3 tools/bench_token_site.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd import _hip
from qsparse_amd.fused import fuse_prune_quantize_pairs

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
SHAPES = (((256, 197, 3072), torch.bfloat16), ((256, 197, 768), torch.bfloat16), ((64, 1024, 4096), torch.bfloat16),
          ((256, 197, 3072), torch.float32))
if os.environ.get("QS_SHAPE"):               # one shape only, e.g. QS_SHAPE=64,1024,4096
    SHAPES = ((tuple(int(v) for v in os.environ["QS_SHAPE"].split(",")), torch.bfloat16),)
for shape, dtype in SHAPES:
    x = torch.randn(shape, device="cuda").to(dtype).requires_grad_(True)
    g = torch.randn(shape, device="cuda")
    site = fuse_prune_quantize_pairs(nn.Sequential(
        nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={2}, start=0, interval=1, repetition=1)),
        qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train())

    def step():
        torch.autograd.grad(site(x), x, g)

    for _ in range(6):
        step()
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e6
    if os.environ.get("QS_NO_EVENTS"):        # (under rocprofv3 --kernel-trace: the composite's own launches only)
        print(f"{shape} {str(dtype)[6:]}: composite step {wall:7.1f} us = {x.numel() / wall / 1e3:6.1f} Gelem/s", flush=True)
        del x, g, site
        torch.cuda.empty_cache()
        continue
    _hip.start_event_log()
    for _ in range(5):
        step()
    log = _hip.stop_event_log(with_bytes=True)
    n = x.numel()
    parts = []
    for k, v in sorted(log.items()):
        us = sum(ms for ms, _ in v[-5 * (len(v) // 5):]) / 5 * 1e3
        nb = sum(b for _, b in v[-5 * (len(v) // 5):]) / 5
        parts.append(f"{k} {us:7.1f} us" + (f" ({nb / us / 1e3 / 8000:.2f} of 8 TB/s)" if nb > 1e6 else ""))
    print(f"{shape} {str(dtype)[6:]}: composite step {wall:7.1f} us = {n / wall / 1e3:6.1f} Gelem/s;  " + ";  ".join(parts), flush=True)
    del x, g, site
    torch.cuda.empty_cache()
