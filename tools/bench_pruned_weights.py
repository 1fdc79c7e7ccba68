#!/usr/bin/env python3
"""A ResNet whose WEIGHTS are pruned and quantized -- convert(prune(...), weight_layers) then convert(quantize(...), weight_layers),
the reference's way of stacking operators on a layer (qsparse/convert.py:199-229, imitation.py:61-68) -- in the steady state of a
frozen-mask recipe (`stop_mask_refresh` passed) and with the stock callback (every read averages the magnitude and rebuilds the
mask): step time with the multi-tensor weight path taking the pruned layers
(`batch_weights=True`, default) and layer by layer.  Development tool:

    python3 tools/bench_pruned_weights.py [resnet50|resnet18] [batch] [steps]"""
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


def build(arch, batched, dims, stock=False, quantized=True):
    qs.set_qsparse_options(batch_weights=batched)
    torch.manual_seed(0)
    net = (resnet50 if arch == "resnet50" else resnet18)(num_classes=1000 if arch == "resnet50" else 10)
    net = qs.convert(net, qs.prune(sparsity=0.5, dimensions=dims, start=1, interval=1, repetition=1,
                                   callback=qs.MagnitudePruningCallback() if stock else
                                   qs.MagnitudePruningCallback(mask_refresh_interval=1, stop_mask_refresh=2)),
                     weight_layers=[nn.Conv2d, nn.Linear], log=False)
    if quantized:
        net = qs.convert(net, qs.quantize(bits=8, timeout=1), weight_layers=[nn.Conv2d, nn.Linear], log=False)
    return net.cuda().to(memory_format=torch.channels_last).train()


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    size = 224 if arch == "resnet50" else 32
    x = torch.randn(batch, 3, size, size, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 10, (batch,), device="cuda")
    for dims, name, stock, quantized in (
            ({0, 1, 2, 3}, "unstructured masks, frozen", False, True), ({1}, "per-input-channel masks, frozen", False, True),
            ({0, 1, 2, 3}, "unstructured masks, the stock callback: magnitude averaged and mask rebuilt on every read", True, True),
            ({1}, "per-input-channel masks, the stock callback -- prune()'s defaults", True, True),
            ({1}, "per-input-channel masks, the stock callback, NO quantizer: convert(model, prune(0.5), weight_layers=[...]) alone", True, False)):
        row = {}
        for batched in (True, False):
            net = build(arch, batched, dims, stock, quantized)
            opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

            def step():
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = F.cross_entropy(net(x), y)
                loss.backward()
                opt.step()

            for _ in range(8):          # through the schedule: masks built, then frozen
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            row["multi-tensor" if batched else "layer by layer"] = round((time.perf_counter() - t0) / steps * 1e3, 2)
            if batched:
                wb = net.__dict__.get("_qs_weight_batcher")
                row["layers taken"] = 0 if wb is None else len(wb.layers)
        qs.set_qsparse_options(batch_weights=True)
        print(f"{arch} batch {batch}, weights pruned 50 % ({name})" + (" + quantized 8-bit per channel" if quantized else "") + f", ms/step: {row}", flush=True)


if __name__ == "__main__":
    main()
