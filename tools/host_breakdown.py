#!/usr/bin/env python3
"""Wall-clock host time spent inside the library's Python entry points during an eager ResNet step (monkeypatched timers; the
autograd thread's backward functions included, which cProfile does not see).  Development tool:
    python tools/host_breakdown.py [resnet18|resnet50] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50
from qsparse_amd import batch, fused

quantize = sys.modules["qsparse_amd.quantize"]      # (the package re-exports the FUNCTION under the module's name)

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
acc = {}


def timed(obj, name, label):
    fn = getattr(obj, name)
    raw = fn.__func__ if isinstance(fn, staticmethod) else fn

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            e = acc.setdefault(label, [0.0, 0])
            e[0] += time.perf_counter() - t0
            e[1] += 1

    setattr(obj, name, staticmethod(wrapper) if isinstance(obj.__dict__.get(name), staticmethod) else wrapper)


timed(fused, "fused_prune_quantize", "site forward (fused_prune_quantize, incl. apply)")
timed(fused._SiteStep, "backward", "site backward (_SiteStep.backward)")
timed(fused._SiteStep, "forward", "  of which _SiteStep.forward")
timed(batch.WeightBatcher, "_precompute", "weights: precompute (3 launches)")
timed(batch._GroupSte, "backward", "weights: grouped STE backward")
timed(quantize._QuantStep, "forward", "lone quantizer forward")
timed(quantize._QuantStep, "backward", "lone quantizer backward")

dev = torch.device("cuda", 0)
torch.manual_seed(0)
if arch == "resnet18":
    model, shape, classes, sp = resnet18(10, True), (bs, 3, 32, 32), 10, 0.5
else:
    model, shape, classes, sp = resnet50(1000, False), (bs, 3, 224, 224), 1000, 0.75
model = convert_pq(model, sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
model = model.to(dev).to(memory_format=torch.channels_last).train()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, classes, (bs,), device=dev)


def step():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.cross_entropy(model(x).float(), y)
    loss.backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize()
acc.clear()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n * 1e3
print(f"{arch} b{bs}: {total:.3f} ms per eager step; host time inside the library per step:")
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:52s} {t / n * 1e3:7.3f} ms  ({c / n:5.1f} calls, {t / c * 1e6:6.1f} us each)")
