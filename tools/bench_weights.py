#!/usr/bin/env python3
"""Weight-side path at scale (development tool): per-step cost of reading `conv.weight`
through prune (unstructured or per-input-channel) + quantize (per-output-channel) for ResNet-sized kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd import _hip

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
dev = "cuda"


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for shape in ((64, 64, 1, 1), (256, 256, 3, 3), (512, 512, 3, 3), (2048, 512, 1, 1)):
    res = {}
    for name, dims, ra in (("unstructured", {0, 1, 2, 3}, False), ("per-in-channel", {1}, False)):
        conv = nn.Conv2d(shape[1], shape[0], shape[2]).to(dev)
        conv = qs.quantize(qs.prune(conv, sparsity=0.5, dimensions=dims, start=0, interval=1, repetition=1,
                                    callback=qs.MagnitudePruningCallback(running_average=ra)),
                           bits=4, timeout=1, channelwise=0)
        conv.train()

        def step():
            w = conv.weight
            w.sum().backward()

        res[name] = round(timeit(step), 1)
    imp = torch.rand(shape[0] * shape[1] * shape[2] * shape[3], device=dev)
    n = imp.numel()
    res["kth+mask only"] = round(timeit(lambda: qs.calculate_mask_given_importance(imp, 0.5)), 1)
    res["torch sort ref"] = round(timeit(lambda: imp >= imp.sort()[0][n // 2]), 1)
    print(shape, "us/step", res, flush=True)
