#!/usr/bin/env python3
"""kernel-time breakdown of one converted ResNet training step (development tool; run under rocprofv3, then
tools/summarize_profile.py):  profile_resnet.py [pq|plain] [resnet50|resnet18] [batch] [channels_last]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
mode = sys.argv[1] if len(sys.argv) > 1 else "pq"
arch = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
if arch == "resnet18":
    base, shape, classes, sp = resnet18(10, True), (int(sys.argv[3]) if len(sys.argv) > 3 else 128, 3, 32, 32), 10, 0.5
else:
    base, shape, classes, sp = resnet50(1000, False), (int(sys.argv[3]) if len(sys.argv) > 3 else 64, 3, 224, 224), 1000, 0.75
m = base if mode == "plain" else convert_pq(base, sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
m = m.cuda().train()
x = torch.randn(shape, device="cuda"); y = torch.randint(0, classes, (shape[0],), device="cuda")
if "channels_last" in sys.argv:
    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
if "frozen" in sys.argv and mode != "plain":
    for mod in m.modules():
        if isinstance(mod, qs.MagnitudePruningCallback):
            mod.stop_mask_refresh = 3
if "batchw" in sys.argv and mode != "plain":
    qs.WeightBatcher(m)
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
def step():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.cross_entropy(m(x).float(), y)
    loss.backward(); opt.step()
for _ in range(12): step()
torch.cuda.synchronize()
for _ in range(10): step()
torch.cuda.synchronize()
