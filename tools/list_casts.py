#!/usr/bin/env python3
"""Which ATen dtype casts does a converted network's training step still issue, and on whose behalf?  One steady-state step of
BASELINE config 3 / 4 under a TorchDispatchMode that logs every `aten._to_copy` between float32 and bf16 with its shape, direction
(forward / backward) and the innermost Python frame outside torch -- the table DESIGN section 9 item 1 and VERDICT r05 item 3 are
about.  Usage: python3 tools/list_casts.py [resnet50|resnet18] [batch]"""
import os
import sys
import traceback
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from torch.utils._python_dispatch import TorchDispatchMode

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50


class CastLog(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = defaultdict(lambda: [0, 0])
        self.phase = "forward"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.ops.aten._to_copy.default and isinstance(args[0], torch.Tensor):
            src, dst = args[0].dtype, kwargs.get("dtype")
            if {src, dst} == {torch.float32, torch.bfloat16} and args[0].numel() > 1 << 16:
                where = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "/torch/" not in fr.filename and "list_casts" not in fr.filename:
                        where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                        break
                key = (self.phase, str(src)[6:] + "->" + str(dst)[6:], tuple(args[0].shape), where)
                self.rows[key][0] += 1
                self.rows[key][1] += args[0].numel() * 6
        return func(*args, **kwargs)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if arch == "resnet50" else 128)
    dev = torch.device("cuda", 0)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, autocast_image=os.environ.get("QS_IMAGE", "1") == "1")
    torch.manual_seed(0)
    if os.environ.get("QS_RECIPE") == "q":      # the quantize-ONLY recipe: every activation site is Sequential(act, QuantizeLayer)
        import torch.nn as nn
        base, shape, classes = (resnet50(1000, False), (batch, 3, 224, 224), 1000) if arch == "resnet50" else (resnet18(10, True), (batch, 3, 32, 32), 10)
        net = qs.convert(base, qs.quantize(bits=4, channelwise=-1, timeout=1), activation_layers=[nn.ReLU], weight_layers=[nn.Conv2d, nn.Linear],
                         input=True, log=False)
    elif arch == "resnet50":
        net, shape, classes = convert_pq(resnet50(1000, False), sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1), (batch, 3, 224, 224), 1000
    else:
        net, shape, classes = convert_pq(resnet18(10, True), sparsity=0.5, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1), (batch, 3, 32, 32), 10
    net = net.to(dev).to(memory_format=torch.channels_last).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), device=dev)

    def step(log=None):
        opt.zero_grad(set_to_none=True)
        if log is not None:
            log.phase = "forward"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(net(x), y)
        if log is not None:
            log.phase = "backward"
        loss.backward()
        opt.step()

    for _ in range(6):
        step()
    if os.environ.get("QS_TIME"):            # step time (best of 3 x 10 steps) next to the table
        import time
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        print(f"{arch} batch {batch} recipe {os.environ.get('QS_RECIPE', 'pq')} autocast_image={qs.get_qsparse_option('autocast_image')}: {best:.2f} ms per step")
    log = CastLog()
    with log:
        step(log)
    total = defaultdict(lambda: [0, 0])
    print(f"{arch} batch {batch}: ATen float32 <-> bf16 casts of one steady-state step (tensors above 64 K elements)")
    print(f"{'phase':9s} {'cast':18s} {'shape':24s} {'n':>3s} {'MB':>8s}  issued from")
    for (phase, cast, shp, where), (n, nbytes) in sorted(log.rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{phase:9s} {cast:18s} {str(shp):24s} {n:3d} {nbytes / 1e6:8.1f}  {where}")
        total[(phase, cast)][0] += n
        total[(phase, cast)][1] += nbytes
    for (phase, cast), (n, nbytes) in sorted(total.items()):
        print(f"total {phase:9s} {cast:18s}: {n} launches, {nbytes / 1e9:.3f} GB (read + write) ~ {nbytes / 6.0e12 * 1e3:.3f} ms at 6 TB/s")


if __name__ == "__main__":
    main()
