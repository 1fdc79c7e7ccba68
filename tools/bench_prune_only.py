#!/usr/bin/env python3
"""A network whose ACTIVATIONS are pruned and nothing is quantized -- convert(model, prune(sparsity, dimensions={1}),
activation_layers=[nn.ReLU]) alone, the reference's structured activation pruning (qsparse/convert.py:199-229, sparse.py:215-273)
-- against the plain network: training step time, eager.  Development tool:  python3 tools/bench_prune_only.py [arch] [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import resnet18, resnet50

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    size, classes = (224, 1000) if arch == "resnet50" else (32, 10)
    x = torch.randn(batch, 3, size, size, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), device="cuda")
    row = {}
    for mode in ("plain", "pruned activations"):
        torch.manual_seed(0)
        net = (resnet50 if arch == "resnet50" else resnet18)(num_classes=classes)
        if mode != "plain":
            net = qs.convert(net, qs.prune(sparsity=0.75 if arch == "resnet50" else 0.5, dimensions={1}, start=2, interval=2, repetition=2),
                             activation_layers=[nn.ReLU], excluded_activation_layer_indexes=[(nn.ReLU, [-1])], log=False)
        net = net.cuda().to(memory_format=torch.channels_last).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(net(x), y)
            loss.backward()
            opt.step()

        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        row[mode] = round((time.perf_counter() - t0) / steps * 1e3, 2)
        del net, opt
        torch.cuda.empty_cache()
    print(f"{arch} batch {batch}, channels_last, bf16 autocast, ms/step: {row}, ratio {row['pruned activations'] / row['plain']:.3f}", flush=True)


if __name__ == "__main__":
    main()
