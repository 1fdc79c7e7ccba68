#!/usr/bin/env python3
"""kernel-time breakdown of one converted ResNet-50 training step (development tool; run under rocprofv3)"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import qsparse_amd as qs
from examples.models import convert_pq, resnet50
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
mode = sys.argv[1] if len(sys.argv) > 1 else "pq"
base = resnet50(1000, False)
m = base if mode == "plain" else convert_pq(base, sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
m = m.cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(64, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (64,), device="cuda")
def step():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.cross_entropy(m(x).float(), y)
    loss.backward(); opt.step()
for _ in range(12): step()
torch.cuda.synchronize()
torch.cuda.cudart().cudaProfilerStart() if hasattr(torch.cuda, "cudart") else None
for _ in range(10): step()
torch.cuda.synchronize()
