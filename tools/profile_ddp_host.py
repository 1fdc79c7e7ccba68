#!/usr/bin/env python3
"""where the host time of a DistributedDataParallel rank's step goes (development tool): the config-3 network in a one-rank RCCL
group -- bare module, DDP without the statistics exchange, DDP with it -- wall time per step and a cProfile of each
    python tools/profile_ddp_host.py [resnet18|resnet50] [batch]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def build(pq, ddp):
    torch.manual_seed(0)
    if arch == "resnet18":
        model, shape, classes, sp = resnet18(10, True), (batch, 3, 32, 32), 10, 0.5
    else:
        model, shape, classes, sp = resnet50(1000, False), (batch, 3, 224, 224), 1000, 0.75
    if pq:
        model = convert_pq(model, sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    net = nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=os.environ.get("QS_BCAST", "1") == "1") if ddp else model
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(shape, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), generator=g, device=dev)

    def step():
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(net(x).float(), y)
        loss.backward()
        opt.step()

    return step


def measure(name, step, n=30, top=18):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:7.2f} ms/step")
    if os.environ.get("QS_NO_PROFILE"):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n * 1e3)
        print(f"{name:40s} best of 5 x {n}: {min(ts):7.3f} ms/step")
        return
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("tottime")
    st.print_stats(top)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:4 + top + 6]))
    if os.environ.get("QS_CALLERS"):          # who calls a function by that name (regular expression)
        s2 = io.StringIO()
        st.stream = s2
        st.print_callers(os.environ["QS_CALLERS"])
        print("\n".join(l[:300] for l in s2.getvalue().splitlines()[:40]))


for pq in (False, True):
    qs.set_qsparse_options(sync_statistics=False)
    measure(f"{'pq' if pq else 'plain'} bare module", build(pq, False))
    measure(f"{'pq' if pq else 'plain'} DDP, no exchange", build(pq, True))
    if pq:
        qs.set_qsparse_options(sync_statistics="always")
        measure("pq DDP, exchange live", build(pq, True))
torch.cuda.synchronize()
dist.destroy_process_group()
