#!/usr/bin/env python3
"""Headline benchmark: Gelem/s of the prune(0.75, dims={1}) -> quantize(4-bit, tensor-wise) training
step (forward + backward, live statistics every step) on a 256x256x56x56 bf16 activation per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input, everything resident in HBM:
    stats   (read x)            staged per-channel mean|x| + per-channel max|x|
    select  (C-sized)           running magnitude, k-th value, mask, scale
    apply   (read x, write y)   y = dequant(round(x*mask / s))         bf16 -> fp32
    bwd     (read g, write gx)  gx = clamp(g) * mask                    fp32 -> bf16
Algorithmic bytes: 14 B/elem for the step (2 + 6 + 6), 6 B/elem for each apply kernel.
Rank 0 prints ONE JSON line.  With N > 1 every rank processes its own batch shard (weak scaling) and
the C-sized statistics are all-reduced over RCCL each step (qsparse_amd/distributed.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.nn as nn

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
SHAPE = (256, 256, 56, 56)


def make_input(shape, device, seed=0):
    """x = relu(randn) * linspace(0.25, 4, C) per channel, bf16; grad_out = randn fp32."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(shape, generator=g, device=device).relu_()
    x *= torch.linspace(0.25, 4.0, shape[1], device=device).view(1, -1, 1, 1)
    x = x.to(torch.bfloat16)
    gout = torch.randn(shape, generator=g, device=device)
    return x, gout


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a headline kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic.json: FETCH_SIZE*1024*2 + WRITE_SIZE*1024, see the file for the correction).
    PMC counters need their own profiler passes, so they are collected with the same command line
    under rocprofv3 and read back here; null if no such profile is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"][kernel_key]
        return k["hbm_bytes"], os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError, OSError):
        return None, None


def make_pair(device):
    import qsparse_amd as qs
    from qsparse_amd.fused import fuse_prune_quantize_pairs

    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    pair = nn.Sequential(
        nn.Sequential(nn.Identity(), qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1)),
        qs.quantize(bits=4, channelwise=-1, timeout=1)).to(device).train()
    return fuse_prune_quantize_pairs(pair)


def _cpu_steps(batch, reps, threads):
    from oracle import qs_oracle as O

    torch.set_num_threads(threads)
    shape = (batch,) + SHAPE[1:]
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(shape, generator=g).relu_() * torch.linspace(0.25, 4.0, shape[1]).view(1, -1, 1, 1)).bfloat16()
    gout = torch.randn(shape, generator=g)
    ps, qsim = O.PruneSim(0.75, [1], 0, 1, 1, False), O.QuantizeSim("scaler", 4, -1, 1)
    best = float("inf")
    for i in range(reps + 2):   # 2 warm-up steps bring both operators live (t > 0)
        t0 = time.perf_counter()
        y = qsim.step(ps.step(x, True), True)
        gx = ps.grad(qsim.grad(gout.clone(), torch.bfloat16), True)
        dt = time.perf_counter() - t0
        if i >= 2:
            best = min(best, dt)
    del y, gx
    return x.numel() / best / 1e9


def cpu_baseline(batch=64, reps=8, threads=None):
    """the oracle (a port of the reference's ATen op chain) timed on this host's cores on a bounded sample of the
    same workload: the same tensor restricted to `batch` samples (about 10 s of CPU work in total), with the
    best thread count found on the GPU box's EPYC (32; 8/16/64/128 threads were slower) and with one thread."""
    cores = threads or int(os.environ.get("QS_CPU_THREADS", "0")) or min(os.cpu_count() or 1, 32)
    multi = _cpu_steps(batch, reps, cores)
    single = _cpu_steps(max(batch // 2, 1), 2, 1)
    return {"value": round(multi, 4), "unit": "Gelem/s", "cores": cores, "kind": "port",
            "value_1thread": round(single, 4),
            "sample": f"oracle/qs_oracle.py PruneSim->QuantizeSim fwd+bwd (train mode, live stats) on "
                      f"{batch}x256x56x56 bf16, best of {reps} ({cores} threads); {max(batch // 2, 1)}x256x56x56, best of 2 "
                      f"(1 thread); torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="THREADS",
                    help="only time the CPU oracle with this many threads (no GPU needed)")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(threads=args.cpu_baseline_only)))
        return

    # stdout must carry exactly ONE line, the JSON record of rank 0.  RCCL / MIOpen / the HIP runtime print banners
    # through C stdio (flushed at exit, i.e. after anything Python printed), so from here on file descriptor 1 is
    # pointed at stderr for every rank and rank 0 writes its record to a private duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development aid for 1-GPU boxes: QS_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo, which
    # exercises the whole N>1 code path (rendezvous, statistics exchange, max-over-ranks timing) without RCCL
    share_gpu = os.environ.get("QS_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # development aid: QS_BENCH_FORCE_EXCHANGE=1 runs the N=1 bench inside a one-rank RCCL group with the statistics
    # exchange live, i.e. with the collective's launch + kernel latency on the step's critical path (what every
    # rank pays at N>1, minus the xGMI hop); the JSON line then says so in config.exchange
    force_exchange = world == 1 and os.environ.get("QS_BENCH_FORCE_EXCHANGE", "0") == "1"
    if world > 1 or force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        if force_exchange:
            import qsparse_amd as qs
            qs.set_qsparse_options(sync_statistics="always")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from qsparse_amd import _hip

    _hip.load()
    x, gout = make_input(SHAPE, device, seed=rank)
    x.requires_grad_(True)
    pair = make_pair(device)

    def step():
        y = pair(x)
        (gx,) = torch.autograd.grad(y, x, gout)
        return gx

    # setup: bring the operators to their steady state (schedule finished, mask refresh and running scale live,
    # allocator pools and clocks settled); not part of the W warm-up steps or the K timed steps
    for _ in range(20):
        step()
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events (torch.cuda.Event on the stream the kernels are launched on) bracket single launches inside the
    # timed region: the two apply kernels on every 4th step, every kernel on every 16th.  Each event pair costs
    # ~5 us of stream time, so bracketing every launch of every step would inflate the step by ~5 %.
    dom = ("quant_scaler_fwd+mask", "quant_ste_bwd+mask")
    fence()
    events = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % 16 == 15 or i == args.steps - 1:     # (the last step too, so that short runs carry a roofline as well)
            _hip.start_event_log(only=None)
        elif i % 4 == 3:
            _hip.start_event_log(only=dom)
        step()
        for kname, pairs in _hip.take_event_pairs().items():
            events.setdefault(kname, []).extend(pairs)
    fence()
    elapsed = time.perf_counter() - t0
    events = {kname: [a.elapsed_time(b) for a, b in pairs] for kname, pairs in events.items()}
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    numel = x.numel()
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * numel * args.steps / elapsed / 1e9
        avg = {k: sum(v) / len(v) for k, v in events.items() if v}
        fwd_ms = avg.get("quant_scaler_fwd+mask")
        bwd_ms = avg.get("quant_ste_bwd+mask")
        stats_ms = avg.get("mean_dim+absmax")
        # dominant kernel: the fused apply forward (bf16 in, fp32 out: 6 B/elem); backward is its mirror image
        dom_name, dom_ms = ("quant_scaler_fwd+mask", fwd_ms) if (fwd_ms or 0) >= (bwd_ms or 0) else ("quant_ste_bwd+mask", bwd_ms)
        achieved = 6 * numel / (dom_ms * 1e-3) / 1e9 if dom_ms else None
        traffic, traffic_src = pmc_traffic("apply_fwd" if dom_name.startswith("quant_scaler_fwd") else "apply_bwd")
        kern = {}
        for name, ms, bpe in (("apply_fwd", fwd_ms, 6), ("apply_bwd", bwd_ms, 6), ("stats", stats_ms, 2)):
            if ms:
                kern[name] = {"ms": round(ms, 4), "GB/s": round(bpe * numel / ms / 1e6, 1),
                              "frac": round(bpe * numel / ms / 1e6 / HBM_PEAK_GBS, 4)}
        out = {
            "metric": "Gelem/s quantize+prune fwd+bwd, 256\u00d7256\u00d756\u00d756 bf16; % HBM roofline",   # BASELINE.json's string
            "value": round(value, 3), "unit": "Gelem/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "prune(0.75,dims={1})->quantize(4-bit,tensor-wise) train fwd+bwd, live mask+scale "
                                   "statistics every step, 256x256x56x56 bf16 in / fp32 out / fp32 grad in / bf16 grad out "
                                   "per GPU (SURVEY 8d scope ii, 14 B/elem)",
                       "shape_per_gpu": list(SHAPE), "fused": True,
                       "step_frac_of_hbm_peak": round(14 * numel / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": 6 * numel, "kernels": kern},
        }
        if world > 1 or force_exchange:
            out["config"]["exchange"] = ("one all-gather of a 2C-float record per step over " +
                                         ("gloo (shared GPU, development)" if share_gpu else "RCCL") +
                                         (" in a ONE-rank group (QS_BENCH_FORCE_EXCHANGE)" if force_exchange else ""))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if world > 1 or force_exchange:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
