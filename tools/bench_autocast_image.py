#!/usr/bin/env python3
"""ResNet training step with and without the `autocast_image` extension (development tool):
    python tools/bench_autocast_image.py [resnet50|resnet18] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def run(image, pq=True, steps=10):
    qs.set_qsparse_options(autocast_image=image)
    torch.manual_seed(0)
    if arch == "resnet18":
        model, shape, classes, sp = resnet18(10, True), (batch, 3, 32, 32), 10, 0.5
    else:
        model, shape, classes, sp = resnet50(1000, False), (batch, 3, 224, 224), 1000, 0.75
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
    del model, opt
    torch.cuda.empty_cache()
    return ms, losses


def run_eval(image, pq=True, steps=20):
    """inference: evaluation mode, no_grad, autocast -- after a few training steps so that masks and scales exist"""
    qs.set_qsparse_options(autocast_image=image)
    torch.manual_seed(0)
    model, shape = (resnet18(10, True), (batch, 3, 32, 32)) if arch == "resnet18" else (resnet50(1000, False), (batch, 3, 224, 224))
    if pq:
        model = convert_pq(model, sparsity=0.5 if arch == "resnet18" else 0.75, bits=4, prune_start=1, prune_interval=1, repetition=1,
                           quant_timeout=1)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            model(x).float().sum().backward()
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(5):
            out = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    res = out.float().clone()
    del model
    torch.cuda.empty_cache()
    return ms, res


if "--eval" in sys.argv:
    p_ms, _ = run_eval(False, pq=False)
    a_ms, a = run_eval(False)
    b_ms, b = run_eval(True)
    print(f"{arch} b{batch} inference: plain {p_ms:.2f} ms, pq {a_ms:.2f} ms ({a_ms / p_ms:.3f}x), pq + autocast_image {b_ms:.2f} ms "
          f"({b_ms / p_ms:.3f}x); outputs equal: {torch.equal(a, b)}")
    sys.exit(0)

plain, _ = run(False, pq=False)
off, la = run(False)
off2, la2 = run(False)
on, lb = run(True)
print("losses off (again):", [round(v, 4) for v in la2], "-- run-to-run noise of the convolutions, for scale")
print(f"{arch} b{batch}: plain {plain:.2f} ms, pq {off:.2f} ms ({off / plain:.3f}x), pq + autocast_image {on:.2f} ms ({on / plain:.3f}x)")
print("losses off:", [round(v, 4) for v in la])
print("losses on: ", [round(v, 4) for v in lb])
