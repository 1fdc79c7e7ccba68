#!/usr/bin/env python3
"""One BASELINE network config in steady state, nothing else -- the command the in-model kernel profiles are taken with:

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o cfg -- python3 tools/profile_config.py resnet50 256 5

builds the converted network (the --pq recipe, channels_last, bf16 autocast, SGD momentum, default library options), runs 8
untimed steps, then K steps, and prints one JSON line: launches of the library per step (counted with the event log on one
extra step BEFORE the K steps) and the algorithmic bytes per kernel family of a step.  tools/family_table.py turns the
kernel trace of the last K steps into the per-family table under profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import bench
import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50
from qsparse_amd import _hip


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda", 0)
    image = os.environ.get("QS_PROFILE_IMAGE", "1") == "1"          # the autocast image (fused.py): the default since round 5
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, autocast_image=image,
                           batch_weights=os.environ.get("QS_PROFILE_NO_BATCHER", "0") != "1",
                           fold_relu=os.environ.get("QS_PROFILE_NO_FOLD", "0") != "1")
    if arch == "resnet18":
        model, shape, classes, sparsity = resnet18(10, True), (batch, 3, 32, 32), 10, 0.5
    else:
        model, shape, classes, sparsity = resnet50(1000, False), (batch, 3, 224, 224), 1000, 0.75
    if os.environ.get("QS_PROFILE_INPLACE", "0") == "1":             # the network written torchvision-style
        for m in model.modules():
            if type(m) is torch.nn.ReLU:
                m.inplace = True
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(shape, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), generator=g, device=dev)
    torch.manual_seed(0)
    model = convert_pq(model, sparsity=sparsity, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)

    def step():
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(x).float(), y)
        loss.backward()
        opt.step()

    for _ in range(8):
        step()
    acct = bench.library_kernel_accounting(step, reps=1, add_image_bytes=image)
    torch.cuda.synchronize()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if image:
        # the event log keeps to the fine-grained entry points (one gradient stream): count the launches of a real step instead
        from qsparse_amd import fused
        acct["launches"] = int(os.environ.get("QS_PROFILE_LAUNCHES", acct["launches"]))
    print(json.dumps({"arch": arch, "batch": batch, "steps": steps, "launches_per_step": acct["launches"], "autocast_image": image,
                      "families": {k: {"GB": v["GB"], "launches": v["launches"]} for k, v in acct["families"].items()},
                      "algorithmic_GB_per_step": acct["algorithmic_GB_per_step"]}))


if __name__ == "__main__":
    main()
