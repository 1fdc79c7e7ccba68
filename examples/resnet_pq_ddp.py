#!/usr/bin/env python3
"""BASELINE.json configs 3-5 as a runnable recipe: ResNet-18 (CIFAR shape) / ResNet-50 (ImageNet shape) with 4-bit
weights and activations + channel pruning of every activation, synthetic data, bf16 autocast, one process per GPU.

    python examples/resnet_pq_ddp.py --arch resnet50 --batch 64 --channels-last [--no-graph] [--no-autocast-image]   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/resnet_pq_ddp.py --arch resnet50 --batch 64 --channels-last                                   # 8 GPUs

Under torchrun the model is wrapped in DistributedDataParallel (backend nccl = RCCL over xGMI): gradients follow
DDP's bucketed all-reduce, masks and scales the per-layer statistics exchange (qsparse_amd/distributed.py), so every
rank holds the same network.  A single process replays whole training steps from a hipGraph once the schedules have finished
(`graphs.GraphedStep`: eager until the steady state, then capture + replay -- the recommended way to run a converted network
whose eager step is host-bound, e.g. ResNet-18 at batch 128: 6.9-7.6 ms eager, 6.6 replayed; --no-graph stays eager; under
torchrun the steps stay eager, a captured DDP step would need its collectives captured too).  Prints images/s, whole job.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50
from qsparse_amd import graphs


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet50", choices=["resnet18", "resnet50"])
    ap.add_argument("--batch", type=int, default=64, help="per GPU")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--sparsity", type=float, default=None, help="default 0.5 (resnet18) / 0.75 (resnet50)")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--no-pq", action="store_true", help="the unconverted network, for comparison")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--graph", action="store_true", help="(the default for a single process; kept for older command lines)")
    ap.add_argument("--no-graph", action="store_true", help="stay eager")
    ap.add_argument("--no-batch-weights", action="store_true", help="layer-by-layer weight quantizers (the multi-tensor path is the default)")
    ap.add_argument("--preserve-dtype", action="store_true")
    ap.add_argument("--no-autocast-image", action="store_true",
                    help="opt out of the (value-identical, default) autocast image: sites and the weight path then hand out plain "
                         "float32 tensors and autocast casts them in front of every convolution")
    ap.add_argument("--inplace-relu", action="store_true", help="build the network with nn.ReLU(inplace=True) modules (torchvision style)")
    args = ap.parse_args(argv)

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, preserve_dtype=args.preserve_dtype,
                           batch_weights=not args.no_batch_weights, autocast_image=not args.no_autocast_image)

    torch.manual_seed(0)                      # identical initial weights on every rank
    if args.arch == "resnet18":
        model, shape, classes, sparsity = resnet18(10, True), (args.batch, 3, 32, 32), 10, 0.5
    else:
        model, shape, classes, sparsity = resnet50(1000, False), (args.batch, 3, 224, 224), 1000, 0.75
    if args.inplace_relu:
        for m in model.modules():
            if type(m) is torch.nn.ReLU:
                m.inplace = True
    if not args.no_pq:
        model = convert_pq(model, sparsity=args.sparsity or sparsity, bits=args.bits, prune_start=2, prune_interval=2,
                           repetition=3, quant_timeout=4)
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(100 + rank)       # every rank its own shard of synthetic data
    x = torch.randn(shape, generator=g).to(dev)
    y = torch.randint(0, classes, (shape[0],), generator=g).to(dev)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if world > 1 else model
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

    def train_step(xb, yb):
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(net(xb).float(), yb)
        loss.backward()
        opt.step()
        return loss.detach()

    step = graphs.GraphedStep(model, train_step) if (not args.no_graph and world == 1 and not args.no_pq) else train_step
    for _ in range(args.warmup):
        loss = step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(x, y)
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        ms = elapsed.item() / args.steps * 1e3
        print(f"{args.arch} batch {args.batch}/GPU x {world} GPU(s){' channels_last' if args.channels_last else ''}"
              f"{' graphed' if getattr(step, 'captured', False) else ''}: {ms:.2f} ms/step, "
              f"{world * args.batch / ms * 1e3:.0f} images/s, loss {float(loss):.3f}")
    if world > 1:
        # the point of the statistics exchange: every rank ends with the same masks and scales
        sd = model.state_dict()
        digest = torch.stack([v.double().sum() for k, v in sd.items() if k.endswith(("mask", "quantize.weight", "1.weight"))]).sum()
        lo, hi = digest.clone(), digest.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            print("operator state identical across ranks:", bool(lo == hi))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
