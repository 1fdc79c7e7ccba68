#!/usr/bin/env python3
"""End-to-end step time of the BASELINE.json model configurations on one GPU (development tool, not the
headline): ResNet-18 (CIFAR shape) / ResNet-50 (ImageNet shape), synthetic data, --pq conversion.
Reports ms/step for: plain model, converted with the fused pair path, converted unfused."""
import argparse
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50


CHANNELS_LAST = False


def run(model, shape, classes, steps, warmup, dtype, graph=False, batch_weights=False):
    model = model.cuda().train()   # fp32 master weights; bf16 compute through autocast (activations are bf16)
    if batch_weights:
        qs.WeightBatcher(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(shape, device="cuda")
    if CHANNELS_LAST:              # NHWC activations and weights: MIOpen's native layout, no transposes around the convolutions
        model = model.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (shape[0],), device="cuda")

    def step():
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            loss = F.cross_entropy(model(x).float(), y)
        loss.backward()
        opt.step()

    for i in range(warmup):
        step()
    if graph:
        from qsparse_amd import graphs
        assert graphs.steady_state(model) or not any(True for _ in model.modules() if hasattr(_, "_kwargs") or type(_).__name__ in ("PruneLayer", "QuantizeLayer"))
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            step()
        step = g.replay
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run_eval(model, shape, steps, dtype, batch_weights=False):
    """inference: forward only, eval mode, no_grad"""
    model = model.cuda().eval()
    if batch_weights:
        qs.WeightBatcher(model)
    x = torch.randn(shape, device="cuda")
    if CHANNELS_LAST:
        model = model.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            return model(x)

    for _ in range(8):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--eval", action="store_true", help="also time inference (forward only) of the trained network")
    ap.add_argument("--frozen-masks", action="store_true",
                    help="stop the mask refresh after 3 steps (the reference recipe's steady state: scales keep following the data)")
    args = ap.parse_args()
    global CHANNELS_LAST
    CHANNELS_LAST = args.channels_last
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    dtype = getattr(torch, args.dtype)
    if args.arch == "resnet18":
        base, shape, classes, sp = resnet18(10, True), (args.batch, 3, 32, 32), 10, 0.5
    else:
        base, shape, classes, sp = resnet50(1000, False), (args.batch, 3, 224, 224), 1000, 0.75
    res = {"plain": run(copy.deepcopy(base), shape, classes, args.steps, 10, dtype),
           "plain_graph": run(copy.deepcopy(base), shape, classes, args.steps, 10, dtype, graph=True)}
    qs.set_qsparse_options(graph_safe=True)
    for name, fuse, graph in (("pq_fused", True, False), ("pq_unfused", False, False), ("pq_fused_batchw", True, False),
                              ("pq_fused_graph", True, True), ("pq_fused_graph_batchw", True, True),
                              ("pq_fused_graph_batchw_preserve_dtype", True, True)):
        qs.set_qsparse_options(preserve_dtype=name.endswith("preserve_dtype"))
        m = convert_pq(copy.deepcopy(base), sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1,
                       quant_timeout=1, fuse=fuse)
        if args.frozen_masks:
            for mod in m.modules():
                if isinstance(mod, qs.MagnitudePruningCallback):
                    mod.stop_mask_refresh = 3
        res[name] = run(m, shape, classes, args.steps, 10, dtype, graph=graph, batch_weights="batchw" in name)
    if args.eval:
        ev = {"plain": run_eval(copy.deepcopy(base), shape, args.steps, dtype)}
        for name in ("pq_fused", "pq_fused_batchw"):
            m = convert_pq(copy.deepcopy(base), sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
            run(m, shape, classes, 6, 6, dtype)            # a few training steps: masks and scales exist
            ev[name] = run_eval(m, shape, args.steps, dtype, batch_weights="batchw" in name)
        print(args.arch, shape, args.dtype, "inference", {k: round(v, 2) for k, v in ev.items()}, flush=True)
    print(args.arch, shape, args.dtype, ("channels_last" if CHANNELS_LAST else "nchw") + (" frozen-masks" if args.frozen_masks else ""), {k: round(v, 2) for k, v in res.items()}, flush=True)


if __name__ == "__main__":
    main()
