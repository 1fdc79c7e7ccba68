#!/usr/bin/env python3
"""cProfile of the HOST side of one converted-network training step (eager), development tool:
    python tools/profile_step_host.py [resnet18|resnet50] [batch]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
dev = torch.device("cuda", 0)
if arch == "resnet18":
    model, shape, classes, sp = resnet18(10, True), (batch, 3, 32, 32), 10, 0.5
else:
    model, shape, classes, sp = resnet50(1000, False), (batch, 3, 224, 224), 1000, 0.75
torch.manual_seed(0)
model = convert_pq(model, sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
model = model.to(dev).to(memory_format=torch.channels_last).train()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, classes, (batch,), device=dev)


def step():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.cross_entropy(model(x).float(), y)
    loss.backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
print(f"{arch} b{batch}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step (wall, eager)")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
