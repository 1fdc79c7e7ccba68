#!/usr/bin/env python3
"""A transformer-style MLP encoder on token-major activations (examples/models.py::TokenNet: patch embedding, `depth` blocks of
LayerNorm -> Linear -> act -> Linear + residual), plain against the --pq recipe on its hidden activations (prune(0.75,
dimensions={2}) + 4-bit quantization of activations and Linear / Conv2d weights): ms per training step, bf16 autocast, SGD.
usage (GPU box): python3 tools/bench_token_net.py [batch=128] [act=gelu|relu]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import TokenNet, convert_pq_tokens
from qsparse_amd.fused import ROUTES

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
act = {"gelu": nn.GELU, "relu": nn.ReLU}[sys.argv[2] if len(sys.argv) > 2 else "gelu"]
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
dev = torch.device("cuda", 0)
x = torch.randn(batch, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (batch,), device=dev)


def measure(pq):
    torch.manual_seed(0)
    net = TokenNet(num_classes=1000, dim=768, hidden=3072, depth=12, patch=16, act=act)
    if pq:
        net = convert_pq_tokens(net, act=act, sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
    net = net.to(dev).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(net(x).float(), y)
        loss.backward()
        opt.step()

    for _ in range(8):
        step()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
    return best


if os.environ.get("QS_ONLY") == "pq":          # (for rocprofv3: the converted network only)
    print(f"--pq {measure(True):.2f} ms; routes {dict(ROUTES)}")
    sys.exit(0)
plain = measure(False)
before = dict(ROUTES)
pq = measure(True)
print(f"TokenNet dim 768 / hidden 3072 / depth 12 / 196 tokens, batch {batch}, act {act.__name__}: plain {plain:.2f} ms, --pq {pq:.2f} ms "
      f"({pq / plain:.3f} x); routes {dict((k, ROUTES[k] - before.get(k, 0)) for k in ROUTES)}")
