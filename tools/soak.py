#!/usr/bin/env python3
"""Soak: a converted network trained for many steps -- device memory, host memory and the loss must stay put (no leak through the
per-site caches, the image bookkeeping, saved tensors of the fused autograd nodes).  usage (GPU box): python3 tools/soak.py [steps=400]"""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
import qsparse_amd as qs
from examples.models import TokenNet, convert_pq_tokens, convert_pq, resnet18
from qsparse_amd.fused import ROUTES

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
dev = torch.device("cuda", 0)
for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("tokennet_gelu", "resnet18")):
    torch.manual_seed(0)
    if name == "tokennet_gelu":
        net = convert_pq_tokens(TokenNet(num_classes=100, dim=192, hidden=768, depth=4, patch=16, act=nn.GELU), act=nn.GELU, sparsity=0.75, bits=4,
                                prune_start=1, prune_interval=1, repetition=1, quant_timeout=1).to(dev).train()
        x = torch.randn(32, 3, 224, 224, device=dev); y = torch.randint(0, 100, (32,), device=dev)
    elif name == "resnet50":          # BASELINE config 4's network and recipe at batch 64
        from examples.models import resnet50
        net = convert_pq(resnet50(1000, False), sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1).to(dev).to(memory_format=torch.channels_last).train()
        x = torch.randn(64, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last); y = torch.randint(0, 1000, (64,), device=dev)
    else:
        net = convert_pq(resnet18(10, True), sparsity=0.5, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1).to(dev).to(memory_format=torch.channels_last).train()
        x = torch.randn(64, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last); y = torch.randint(0, 10, (64,), device=dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    marks = []
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(net(x).float(), y)
        loss.backward()
        opt.step()
        if s in (steps // 4, steps // 2, steps - 1):
            torch.cuda.synchronize()
            marks.append((s, torch.cuda.memory_allocated() >> 10, torch.cuda.max_memory_allocated() >> 10, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, round(loss.item(), 4)))
    print(name, "step / device KiB / peak KiB / host MiB / loss:", marks, "routes", dict(ROUTES))
    a, b = marks[0], marks[-1]
    assert b[1] <= a[1] + 1024, "device memory grew"
    assert b[3] <= a[3] + 64, "host memory grew"
    assert b[4] == b[4], "loss is NaN"
print("soak ok")
