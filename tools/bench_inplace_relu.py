#!/usr/bin/env python3
"""ResNet training step of a network built with nn.ReLU(inplace=True) modules (torchvision style): the in-place ReLU owned by
the fused sites (default) against the module-by-module route (`fold_relu=False`) and against the same network with
out-of-place ReLU modules (development tool):
    python tools/bench_inplace_relu.py [resnet50|resnet18] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
qs.set_qsparse_options(log_on_created=False, log_during_train=False, autocast_image=bool(int(os.environ.get("QS_IMAGE", "0"))))


def run(inplace, fold, pq=True, steps=10):
    qs.set_qsparse_options(fold_relu=fold)
    torch.manual_seed(0)
    if arch == "resnet18":
        model, shape, classes, sp = resnet18(10, True), (batch, 3, 32, 32), 10, 0.5
    else:
        model, shape, classes, sp = resnet50(1000, False), (batch, 3, 224, 224), 1000, 0.75
    for m in model.modules():
        if type(m) is nn.ReLU:
            m.inplace = inplace
    if pq:
        model = convert_pq(model, sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(shape, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), generator=g, device=dev)
    losses = []

    def step():
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(x).float(), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(8):
        losses.append(step().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    del model, opt
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    qs.set_qsparse_options(fold_relu=True)
    return ms, losses, peak


if __name__ == "__main__":
    plain, _, _ = run(True, True, pq=False)
    print(f"{arch} b{batch} plain network (in-place ReLU): {plain:.2f} ms/step")
    for name, inplace, fold in (("in-place ReLU, module by module (fold_relu=False)", True, False),
                                ("in-place ReLU owned by the sites (default)", True, True),
                                ("out-of-place ReLU folded (default)", False, True)):
        ms, losses, peak = run(inplace, fold)
        print(f"{name:55s} {ms:7.2f} ms/step = {ms / plain:.3f} x plain   peak {peak:.1f} GiB   losses {[round(v, 4) for v in losses[:3]]} .. {losses[-1]:.4f}")
