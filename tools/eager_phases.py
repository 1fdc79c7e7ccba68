#!/usr/bin/env python3
"""Where an eager training step of a converted network loses time against the plain one: wall time of the forward and of the
backward on their own (a synchronisation after each), plain vs converted (`--pq` recipe), channels_last, bf16 autocast.
Development tool:  python3 tools/eager_phases.py [resnet18|resnet50] [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50

qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    size, classes = (224, 1000) if arch == "resnet50" else (32, 10)
    x = torch.randn(batch, 3, size, size, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), device="cuda")
    for mode in ("plain", "converted"):
        torch.manual_seed(0)
        net = (resnet50 if arch == "resnet50" else resnet18)(num_classes=classes)
        if mode == "converted":
            net = convert_pq(net, sparsity=0.75 if arch == "resnet50" else 0.5, bits=4, prune_start=2, prune_interval=2, repetition=2,
                             quant_timeout=1)
        net = net.cuda().to(memory_format=torch.channels_last).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
        tf = tb = tw = 0.0
        for i in range(steps + 10):
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(net(x), y)
            t1 = time.perf_counter()          # host done with the forward
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            loss.backward()
            t3 = time.perf_counter()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            opt.step()
            if i >= 10:
                tf += t2 - t0
                tb += t4 - t2
                tw += (t1 - t0, t3 - t2)[0]
        print(f"{arch} b{batch} {mode:10s} forward {tf / steps * 1e3:6.2f} ms (host alone {tw / steps * 1e3:5.2f})   backward {tb / steps * 1e3:6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
