#!/usr/bin/env python3
"""Headline benchmark: Gelem/s of the prune(0.75, dims={1}) -> quantize(4-bit, tensor-wise) training
step (forward + backward, live statistics every step) on a 256x256x56x56 bf16 activation per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input, everything resident in HBM:
    stats   (read x)            staged per-channel mean|x| + per-channel max|x|
    select  (C-sized)           running magnitude, k-th value, mask, scale
    apply   (read x, write y)   y = dequant(round(x*mask / s))         bf16 -> fp32
    bwd     (read g, write gx)  gx = clamp(g) * mask                    fp32 -> bf16
Algorithmic bytes: 14 B/elem for the dense step (2 + 6 + 6), 6 B/elem for each apply kernel.  The
library's default (`elide_pruned="forward"`, bit-identical for every input: the select marks the pruned channels that
hold a NaN / Inf and those are loaded) does not load the x of pruned channels in the apply forward, so its forward moves (2 * kept + 4) B/elem: every byte count below is the mask-aware one for the
mode that ran, and `config.variants` carries the dense step ("off") and the fully elided one ("all") next to it.

Rank 0 prints ONE JSON line.  With N > 1 every rank processes its own batch shard (weak scaling) and the C-sized
statistics are exchanged over RCCL each step (qsparse_amd/distributed.py); after the timed region the ranks also run
BASELINE config 5 -- ResNet-50, the --pq recipe, DistributedDataParallel -- and the line carries its whole-job images/s
in `configs.config5_resnet50_ddp` (under a watchdog: it can never cost the headline record).  At N == 1 the line also carries
`configs`: BASELINE.json's configs 2-4 (the 8-bit quantizer alone on 256x64x56x56; ResNet-18 CIFAR shape and ResNet-50
ImageNet shape, plain vs converted, eager and hipGraph replay) measured in the same process.
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
import torch.nn.functional as F

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
SHAPE = (256, 256, 56, 56)
SHAPE2 = (256, 64, 56, 56)   # BASELINE config 2


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


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (test infrastructure) timed on the host cores -- reported next to the GPU number, never
# part of it
# ---------------------------------------------------------------------------------------------------
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
        gx = ps.grad(qsim.grad(gout, torch.bfloat16), True)      # (O.ste_bwd works on its own copy of gout)
        dt = time.perf_counter() - t0
        if i >= 2:
            best = min(best, dt)
    del y, gx
    return x.numel() / best / 1e9


def cpu_baseline(batch=None, reps=3, threads=None):
    """the oracle (a port of the reference's ATen op chain) timed on this host's cores as BASELINE.md section 3 plans it:
    the SAME tensor as the GPU run (all 256 samples; QS_CPU_BATCH restricts it to a slice on small hosts), training mode
    in steady state, min of 3 after the warm-up steps -- about 10 s of CPU work -- with the best thread count found on
    the GPU box's EPYC (32; 8/16/64/128 threads were slower) and, on a 32-sample slice, with one thread."""
    cores = threads or int(os.environ.get("QS_CPU_THREADS", "0")) or min(os.cpu_count() or 1, 32)
    batch = batch or int(os.environ.get("QS_CPU_BATCH", "0")) or SHAPE[0]
    multi = _cpu_steps(batch, reps, cores)
    single = _cpu_steps(min(32, batch), 2, 1)
    host = os.cpu_count() or 1
    allcores = multi if host == cores else _cpu_steps(batch, 2, host)      # BASELINE.md section 3's all-cores figure
    what = "the headline tensor itself" if batch == SHAPE[0] else f"a {batch}-sample slice of the headline tensor"
    return {"value": round(multi, 4), "unit": "Gelem/s", "cores": cores, "threads": cores, "host_cores": os.cpu_count(), "kind": "port",
            "value_1thread": round(single, 4), "value_allcores": round(allcores, 4),
            "sample": f"{batch}x256x56x56 bf16 ({what}), oracle PruneSim->QuantizeSim train fwd+bwd, min of {reps}, {cores} thr",
            "sample_detail": f"oracle/qs_oracle.py PruneSim->QuantizeSim fwd+bwd (train mode, live stats) on {what} "
                             f"({batch}x256x56x56 bf16), min of {reps} after 2 warm-up steps ({cores} threads of {os.cpu_count()} "
                             f"host cores); {min(32, batch)}x256x56x56, best of 2 (1 thread); torch {torch.__version__} CPU"}


# ---------------------------------------------------------------------------------------------------
# BASELINE configs 2-4 (N == 1 only)
# ---------------------------------------------------------------------------------------------------
def _timed_loop(fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _steady_ms(fn, steps, rounds=3):
    """eager step time of a network in steady state: the best of `rounds` timed loops.  (A single loop right after the warm-up
    steps reads 5-12 % high on host-bound networks -- allocator pools, MIOpen's per-shape caches and the sites' fast paths are
    still settling: round 5's first record had the default ResNet-18 at 7.51 ms and the SAME kernels, measured a minute later as
    the `elide_off` variant, at 6.71.)  Plain and converted networks are measured the same way."""
    return min(_timed_loop(fn, steps) for _ in range(rounds))


def config2(device, steps=200, nbuf=4):
    """BASELINE config 2: QuantizeLayer(bits=8, tensor-wise) alone, training step (abs-max + running scale + apply
    forward + STE backward = 14 B/elem) on 256x64x56x56 bf16, `nbuf` rotating buffer sets (4 x 103 MB of inputs + 4 x 206 MB
    of gradients: the 256 MiB Infinity Cache cannot hold a working set across steps), eager and as hipGraph replay (what the kernels alone take)."""
    import qsparse_amd as qs

    q = qs.quantize(bits=8, channelwise=-1, timeout=1).to(device).train()
    xs, gs = [], []
    for k in range(nbuf):
        x, g = make_input(SHAPE2, device, seed=100 + k)
        xs.append(x.requires_grad_(True))
        gs.append(g)

    def step(i):
        k = i % nbuf
        return torch.autograd.grad(q(xs[k]), xs[k], gs[k])

    for i in range(12):
        step(i)
    eager = _timed_loop(step, steps)
    qs.set_qsparse_options(graph_safe=True)
    try:
        for i in range(4):
            step(i)
        gr = torch.cuda.CUDAGraph()      # ONE graph of nbuf steps: a hipGraphLaunch costs more than a 3-kernel step
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for k in range(nbuf):
                step(k)
        gr.replay()
        graphed = _timed_loop(lambda i: gr.replay(), max(steps // nbuf, 1)) / nbuf
    finally:
        qs.set_qsparse_options(graph_safe=False)
        qs.resync_host_state(q)
    n = xs[0].numel()

    def rec(ms):
        return {"ms_per_step": round(ms, 4), "Gelem/s": round(n / ms / 1e6, 1),
                "frac_of_hbm_peak": round(14 * n / ms / 1e6 / HBM_PEAK_GBS, 4)}

    return {"workload": "QuantizeLayer(bits=8, tensor-wise) train fwd+bwd, 256x64x56x56 bf16 in / fp32 out, 14 B/elem",
            "rotating_buffers": nbuf, "steps": steps, "eager": rec(eager), "graph_replay": rec(graphed)}


# kernel families of the in-model accounting: the streaming launches carry the bytes the reference's dtype contract makes
# them move (every data operand once, in the dtype and layout the site really saw -- qsparse_amd/_hip.py `_timed`); the
# C-sized launches (running means, selects, scale updates) are latency and carry none
FAMILIES = (("apply_fwd", ("quant_scaler_fwd", "quant_decimal_fwd", "quant_line_fwd", "multi_quant_fwd")),
            ("apply_bwd", ("quant_ste_bwd", "quant_ste_relu_bwd", "multi_ste_bwd")),
            ("mask_apply", ("mask_apply",)),
            ("statistics", ("mean_dim", "mean_last2", "absmax", "minmax", "l0_flag", "multi_absmax", "kth_value", "mask_ge")))


def library_kernel_accounting(step, reps=3, add_image_bytes=False):
    """`reps` training steps with a HIP event pair around every launch of the library: per kernel family the time per
    step, the algorithmic bytes per step and the fraction of the 8 TB/s roofline they amount to; overall the same with
    the C-sized launches' time in the denominator as well (they move nothing, so they only cost)."""
    from qsparse_amd import _hip

    _hip.start_event_log(only=None)
    for _ in range(reps):
        step()
    per = _hip.stop_event_log(with_bytes=True)
    # the event log takes the fine-grained entry points, which make no autocast image: what the image route (the default under
    # autocast) moves differently -- the image written by the forward, the 2-byte gradient read by the backward instead of
    # (or besides) the float32 one -- is counted by the composite site step itself over the same number of ordinary steps
    _hip.image_byte_delta = {"apply_fwd": 0, "apply_bwd": 0}
    try:
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        image_delta = {k: v // reps for k, v in _hip.image_byte_delta.items()}
    finally:
        _hip.image_byte_delta = None
    # what an event pair costs by itself: the pair brackets the launch's dispatch latency as well as its execution, which
    # back-to-back launches overlap; an EMPTY pair on the same (busy) stream measures that floor
    torch.cuda.synchronize()
    empty = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        empty.append((a, b))
    torch.cuda.synchronize()
    pair_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]

    def family_of(kernel):
        base = kernel.split("+")[0]
        for fam, names in FAMILIES:
            if base in names:
                return fam
        return "c_sized"

    fams, by_kernel = {}, {}
    for k, launches in per.items():
        f = fams.setdefault(family_of(k), {"ms": 0.0, "net": 0.0, "bytes": 0, "launches": 0})
        f["ms"] += sum(ms for ms, _ in launches) / reps
        f["net"] += sum(max(ms - pair_ms, 0.0) for ms, _ in launches) / reps
        f["bytes"] += sum(nb for _, nb in launches) // reps
        f["launches"] += len(launches) // reps
        by_kernel[k] = round(sum(ms for ms, _ in launches) / reps, 3)
    # `add_image_bytes`: for callers that set these bytes against durations of ORDINARY steps (tools/profile_config.py: rocprofv3
    # times the image route's kernels).  The event pairs below time the fine-grained route, whose bytes are counted as they are.
    if add_image_bytes:
        for fam, delta in image_delta.items():
            if fam in fams:
                fams[fam]["bytes"] += delta
    total_ms = sum(f["ms"] for f in fams.values())
    total_net = sum(f["net"] for f in fams.values())
    total_bytes = sum(f["bytes"] for f in fams.values())

    def frac(nbytes, ms):
        return round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (nbytes and ms) else None

    rec = {"ms_per_step": round(total_ms, 3), "launches": sum(f["launches"] for f in fams.values()),
           "algorithmic_GB_per_step": round(total_bytes / 1e9, 3), "frac_of_hbm_peak": frac(total_bytes, total_ms),
           "event_pair_overhead_us": round(pair_ms * 1e3, 2), "ms_per_step_net_of_event_overhead": round(total_net, 3),
           "frac_of_hbm_peak_net_of_event_overhead": frac(total_bytes, total_net),
           "families": {}, "by_kernel_ms": dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]))}
    for name, f in sorted(fams.items(), key=lambda kv: -kv[1]["ms"]):
        rec["families"][name] = {"ms": round(f["ms"], 3), "ms_net": round(f["net"], 3), "launches": f["launches"],
                                 "GB": round(f["bytes"] / 1e9, 3), "frac_of_hbm_peak": frac(f["bytes"], f["ms"]),
                                 "frac_net": frac(f["bytes"], f["net"])}
    rec["autocast_image_bytes_per_step"] = dict(image_delta, included=bool(add_image_bytes),
                                                note="what an ordinary step (autocast image, the default) moves on top of the fine-grained "
                                                     "route timed here: the image written by the forward, the 2-byte gradient read by the backward")
    rec["note"] = ("HIP event pairs around every library launch; `ms` is the raw sum (what `frac_of_hbm_peak` uses), `*_net` subtracts "
                   "the cost of an empty event pair from every launch (the pair also brackets dispatch latency that back-to-back "
                   "launches overlap; rocprofv3 kernel durations, profiles/r04_config*_family_table.txt, are the reference). "
                   "bytes = every data operand of a launch once, dense, in the dtype / layout the site saw; c_sized launches "
                   "carry no bytes and count in the overall fractions' time only")
    return rec


def kernel_trace_reference(tag):
    """the committed rocprofv3 per-family table of this config (tools/profile_model.sh -> profiles/rNN_<tag>_family_table.txt):
    kernel-trace durations are the reference for what the library's launches cost on the GPU -- the live event pairs above
    bracket dispatch latency as well.  Returned as recorded (another box, another day): a cross-check, not a live number."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_family_table.txt")))
    if not files:
        return None
    try:
        fams = {}
        for line in open(files[-1]):
            m = re.match(r"^(apply_fwd|apply_bwd|statistics|c_sized|mask_apply|all)\s+([0-9.]+)\s+(\d+)\s+([0-9.]+)\s+\d+\s+([0-9.]+)", line)
            if m:
                fams[m.group(1)] = {"ms": float(m.group(2)), "launches": int(m.group(3)), "GB": float(m.group(4)),
                                    "frac_of_hbm_peak": float(m.group(5)) or None}
        return {"source": os.path.relpath(files[-1], ROOT), "method": "rocprofv3 --kernel-trace, last 5 steps", **fams} if "all" in fams else None
    except (OSError, ValueError):
        return None


def resnet_config(arch, batch, device, steps):
    """BASELINE configs 3 / 4: full-width ResNet-18 (CIFAR shape, 50 % channel pruning) / ResNet-50 (ImageNet shape,
    75 %), 4-bit weights and activations, bf16 autocast, channels_last (MIOpen's native layout), SGD with momentum,
    synthetic data: ms per training step of the plain network and of the network converted with the reference's --pq
    recipe, eager and as whole-step hipGraph replay, plus the time the library's own kernels take inside a step."""
    import qsparse_amd as qs
    from examples.models import convert_pq, resnet18, resnet50
    from qsparse_amd import _hip, graphs

    if arch == "resnet18":
        make, shape, classes, sparsity = (lambda: resnet18(10, True)), (batch, 3, 32, 32), 10, 0.5
    else:
        make, shape, classes, sparsity = (lambda: resnet50(1000, False)), (batch, 3, 224, 224), 1000, 0.75
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(shape, generator=g, device=device).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), generator=g, device=device)
    out = {"model": arch, "input_shape": list(shape), "dtype": "bf16 autocast, fp32 master weights", "layout": "channels_last",
           "optimizer": "SGD momentum 0.9", "steps": steps}

    def build(pq):
        torch.manual_seed(0)
        model = make()
        if pq:
            model = convert_pq(model, sparsity=sparsity, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
        model = model.to(device).to(memory_format=torch.channels_last).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)

        def step(_=0):
            opt.zero_grad(set_to_none=False)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(model(x).float(), y)
            loss.backward()
            opt.step()

        return model, step

    def capture(step):
        gr = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            step()
        for _ in range(3):      # the first replays of a large graph carry one-time work (observed: 64.9 vs 59.0 ms/step)
            gr.replay()
        torch.cuda.synchronize()
        return gr

    # plain network
    model, step = build(False)
    for _ in range(6):
        step()
    out["plain_ms"] = round(_steady_ms(step, steps), 3)
    gr = capture(step)
    out["plain_graph_ms"] = round(_timed_loop(lambda i: gr.replay(), steps), 3)
    del model, step, gr
    torch.cuda.empty_cache()

    # converted network (the --pq recipe), schedules finished after 3 steps
    qs.set_qsparse_options(graph_safe=True)
    try:
        model, step = build(True)
        for _ in range(8):
            step()
        out["pq_ms"] = round(_steady_ms(step, steps), 3)
        out["library_kernels"] = library_kernel_accounting(step)
        lib_ms = out["library_kernels"]["ms_per_step"]
        ref = kernel_trace_reference("config3" if arch == "resnet18" else "config4")
        if ref:
            out["library_kernels"]["kernel_trace_reference"] = ref
        assert graphs.steady_state(model), "converted network did not reach its steady state"
        gr = capture(step)
        out["pq_graph_ms"] = round(_timed_loop(lambda i: gr.replay(), steps), 3)
        out["library_share_of_pq_graph_step"] = round(lib_ms / out["pq_graph_ms"], 4)
        out["pq_over_plain"] = {"eager": round(out["pq_ms"] / out["plain_ms"], 4),
                                "graph": round(out["pq_graph_ms"] / out["plain_graph_ms"], 4)}
        masks = [m.mask.float().mean().item() for m in model.modules() if isinstance(m, qs.sparse.PruneLayer)]
        out["mean_kept_channel_fraction"] = round(sum(masks) / max(len(masks), 1), 4)
        del model, step, gr
        torch.cuda.empty_cache()

        # the same network with every numerics-changing opt-in extension of the library switched on (none of them is the
        # default, none of them enters the numbers above): bf16 outputs instead of the reference's fp32 promotion
        # (preserve_dtype), backward / mask-apply elision.  (The multi-tensor weight path is part of the default since round 3.)
        def opt_in_run(key, label, graph, **options):
            """an opt-in configuration measured like the default one; its failure is recorded under its own key only"""
            try:
                qs.set_qsparse_options(**options)
                model, step = build(True)
                for _ in range(8):
                    step()
                rec = {"options": label, "pq_ms": round(_steady_ms(step, steps), 3)}
                if graph and graphs.steady_state(model):
                    gr = capture(step)
                    rec["pq_graph_ms"] = round(_timed_loop(lambda i: gr.replay(), steps), 3)
                    del gr
                rec["best_over_plain"] = round(min(v for k, v in rec.items() if k.endswith("_ms")) /
                                               min(out["plain_ms"], out["plain_graph_ms"]), 4)
                del model, step
            except Exception as e:      # noqa: BLE001  (recorded verbatim in the JSON line)
                rec = {"options": label, "error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                qs.set_qsparse_options(preserve_dtype=False, elide_pruned="forward", autocast_image=True)
                torch.cuda.empty_cache()
            out[key] = rec

        opt_in_run("opt_in_extensions", "preserve_dtype=True, elide_pruned='all'", True, preserve_dtype=True, elide_pruned="all")
        # the default WITHOUT the autocast image (round 4's default; `set_qsparse_options(autocast_image=False)`): every fused
        # site returns a plain float32 tensor and its first convolution casts it (fp32 -> bf16 forward, bf16 -> fp32 backward:
        # two 6 B/elem passes per site).  Since round 5 the site hands that convolution the bf16 image itself and takes its
        # bf16 gradient as it is (fused.py "Autocast image"): same values, and everything that can observe the output's
        # gradient sees the whole one
        opt_in_run("autocast_image_off", "autocast_image=False (plain float32 tensor out of every site: round 4's default)", True,
                   autocast_image=False)
        # every element loaded.  The default elides only where a pruned channel is a skippable row (NCHW, no gate recording; exact
        # for non-finite inputs as well since ABI v19), so in this channels_last training step "off" and the default run the
        # same kernels: the figure is the evidence of that
        opt_in_run("elide_off", "elide_pruned='off' (every element loaded; the channels_last default already does)", False,
                   elide_pruned="off")
    finally:
        qs.set_qsparse_options(graph_safe=False, preserve_dtype=False, elide_pruned="forward", autocast_image=True)
        torch.cuda.empty_cache()
    return out


def exchange_live_config(arch, device, steps=None):
    """the path a DistributedDataParallel rank really runs, measured on ONE GPU: the config-3 / config-4 network and recipe
    inside a one-rank `nccl` (RCCL) process group with `sync_statistics="always"`, wrapped in DDP -- every operator site then
    issues its statistics launches, the 2C-float all-gather and the combining select + apply (qs_site_stats / qs_site_fwd
    with QS_SITE_STATS_DONE: two calls around one collective), the quantize-only sites an all-reduce of their abs-max lines;
    hipGraph capture is off (an RCCL collective inside a capture crashes on this stack), so this is an eager figure.
    Runs in a CHILD process of the N = 1 bench (a crash inside RCCL must not cost the headline record)."""
    import qsparse_amd as qs
    from examples.models import convert_pq, resnet18, resnet50

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    batch = 128 if arch == "resnet18" else 256
    steps = steps or (10 if arch == "resnet18" else 5)
    if arch == "resnet18":
        make, shape, classes, sparsity = (lambda: resnet18(10, True)), (batch, 3, 32, 32), 10, 0.5
    else:
        make, shape, classes, sparsity = (lambda: resnet50(1000, False)), (batch, 3, 224, 224), 1000, 0.75
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(shape, generator=g, device=device).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, classes, (batch,), generator=g, device=device)

    def build(pq):
        torch.manual_seed(0)
        model = make()
        if pq:
            model = convert_pq(model, sparsity=sparsity, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
        model = model.to(device).to(memory_format=torch.channels_last).train()
        net = nn.parallel.DistributedDataParallel(model, device_ids=[device.index])
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

        def step(_=0):
            opt.zero_grad(set_to_none=False)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(net(x).float(), y)
            loss.backward()
            opt.step()

        return model, step

    out = {"model": arch, "input_shape": list(shape), "steps": steps,
           "group": "one-rank nccl (RCCL) process group, DistributedDataParallel, eager"}
    model, step = build(False)
    for _ in range(6):
        step()
    out["plain_ddp_ms"] = round(_timed_loop(step, steps), 3)
    del model, step
    torch.cuda.empty_cache()
    qs.set_qsparse_options(sync_statistics="always")
    model, step = build(True)
    for _ in range(8):
        step()
    from qsparse_amd import distributed as qdist
    assert qdist.exchange_active()
    out["pq_ddp_exchange_live_ms"] = round(_timed_loop(step, steps), 3)
    acc = library_kernel_accounting(step)
    out["library_kernels"] = {k: acc[k] for k in ("ms_per_step", "launches", "algorithmic_GB_per_step", "frac_of_hbm_peak")}
    out["library_kernels"]["families"] = {k: {"ms": v["ms"], "launches": v["launches"]} for k, v in acc["families"].items()}
    qs.set_qsparse_options(sync_statistics=False)          # the same DDP network without the exchange (ranks would drift)
    for _ in range(3):
        step()
    out["pq_ddp_no_exchange_ms"] = round(_timed_loop(step, steps), 3)
    qs.set_qsparse_options(sync_statistics="always")
    for _ in range(3):
        step()
    out["pq_ddp_exchange_live_ms_2"] = round(_timed_loop(step, steps), 3)
    live = min(out["pq_ddp_exchange_live_ms"], out["pq_ddp_exchange_live_ms_2"])
    out["exchange_live_over_no_exchange"] = round(live / out["pq_ddp_no_exchange_ms"], 4)
    out["pq_over_plain"] = round(live / out["plain_ddp_ms"], 4)
    sites = sum(1 for m in model.modules() if isinstance(m, qs.sparse.PruneLayer))
    out["collectives_per_step"] = f"{sites} record all-gathers (2C floats each) + all-reduces of the quantize-only sites"
    # the prototype without a host collective for the pair sites: records published into peer-mapped mailboxes (one rank here: its
    # own), two launches per site instead of the all-gather (include/qsparse_hip.h, "peer-mapped mailboxes"; off by default)
    try:
        qs.set_qsparse_options(sync_statistics="mailbox")
        for _ in range(4):
            step()
        out["pq_ddp_mailbox_ms"] = round(_timed_loop(step, steps), 3)
        out["mailbox_over_no_exchange"] = round(out["pq_ddp_mailbox_ms"] / out["pq_ddp_no_exchange_ms"], 4)
        torch.cuda.synchronize()
        qdist.close_mailboxes()
    except Exception as e:      # noqa: BLE001 -- a prototype must not cost the record
        out["mailbox_error"] = f"{type(e).__name__}: {e}"[:200]
    qs.set_qsparse_options(sync_statistics="always")
    torch.cuda.synchronize()
    dist.destroy_process_group()
    return out


def exchange_live_in_child(arch, timeout=900):
    """run `exchange_live_config` in a child process and return its record (or what went wrong)"""
    import subprocess
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--exchange-live", arch], capture_output=True, text=True,
                           timeout=timeout, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rec = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": f"child exit code {r.returncode}: {r.stderr[-300:]}"}
    except Exception as e:      # noqa: BLE001  (recorded verbatim in the JSON line)
        rec = {"error": f"{type(e).__name__}: {e}"[:300]}
    rec["bench_seconds"] = round(time.perf_counter() - t0, 1)
    return rec


def host_overhead(device, steps=300):
    """host time per operator-site training step (forward + backward): the same sites on tensors so small (8x64x16x16) that the
    GPU side is a few microseconds, so wall time per step IS the host's share -- Python state machines, the autograd Function,
    the FFI call(s).  What an eager (not graph-replayed) network pays per site and step; `plain_relu` is the floor a bare
    nn.ReLU sets on the same loop."""
    import qsparse_amd as qs
    from qsparse_amd.fused import fuse_prune_quantize_pairs

    def pair(act, dim=1):
        return fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(act, qs.prune(sparsity=0.75, dimensions={dim}, start=0, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1)).to(device).train())

    shape4, shape2, shape3 = (8, 64, 16, 16), (2048, 64), (8, 256, 64)       # the same element count, 64 channels
    sites = {"plain_relu": (nn.ReLU().to(device), shape4),
             "relu_prune_quantize_pair": (pair(nn.ReLU()), shape4),
             "prune_quantize_pair": (pair(nn.Identity()), shape4),
             "relu_quantize": (fuse_prune_quantize_pairs(nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=1)).to(device).train()), shape4),
             "quantize_alone": (qs.quantize(bits=8, channelwise=-1, timeout=1).to(device).train(), shape4),
             # a 2-d [N, C] site (behind an nn.Linear) and a token-major [B, T, C] one with the mask on the last dim (qs_site_plan
             # layouts 2 and 3): the composite serves both
             "relu_prune_quantize_pair_2d": (pair(nn.ReLU()), shape2),
             "relu_prune_quantize_pair_token_major": (pair(nn.ReLU(), 2), shape3)}
    out = {"shape": list(shape4), "shape_2d": list(shape2), "shape_token_major": list(shape3), "steps": steps,
           "unit": "us of host time per site step (forward + backward)"}
    for name, (site, shape) in sites.items():
        x = torch.randn(shape, device=device, dtype=torch.bfloat16, requires_grad=True)
        gs = {torch.float32: torch.randn(shape, device=device), torch.bfloat16: torch.randn(shape, device=device).bfloat16()}

        def step(_=0):
            y = site(x)
            torch.autograd.grad(y, x, gs[y.dtype])
        for _ in range(20):
            step()
        out[name] = round(_timed_loop(step, steps) * 1e3, 1)
    return out


def weights_pruned_config(device):
    """the weight side on its own: every Conv2d / Linear weight of a ResNet read through prune() with its defaults
    (one mask entry per input channel, the stock MagnitudePruningCallback: running magnitude averaged and mask rebuilt on every
    read) and through quantize() with its defaults (8 bits, per channel) -- `convert(model, prune(0.5), weight_layers=...)` then
    `convert(model, quantize(), weight_layers=...)`, reference convert.py:199-229 / imitation.py:61-68 -- training step time with
    the multi-tensor weight path (default) and layer by layer, next to the plain network"""
    import qsparse_amd as qs
    from examples.models import resnet18

    out = {"recipe": "convert(prune(sparsity=0.5), weight_layers=[Conv2d, Linear]); convert(quantize(bits=8), weight_layers=[...]); "
                     "channels_last, bf16 autocast, SGD momentum 0.9; ms per training step in steady state"}
    # (ResNet-18 only: the ResNet-50 figures -- 17.4 / 22.6 / 16.6 ms at batch 64 -- come from tools/bench_pruned_weights.py; new
    #  convolution shapes in the middle of this process have cost MIOpen seconds per step on some boxes)
    for arch, ctor, batch, size, classes, steps in (("resnet18", resnet18, 128, 32, 10, 20),):
        x = torch.randn(batch, 3, size, size, device=device).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, classes, (batch,), device=device)
        row = {}
        for mode in ("plain", "multi_tensor", "layer_by_layer"):
            qs.set_qsparse_options(batch_weights=mode != "layer_by_layer")
            try:
                torch.manual_seed(0)
                net = ctor(num_classes=classes)
                if mode != "plain":
                    net = qs.convert(net, qs.prune(sparsity=0.5, start=1, interval=1, repetition=1), weight_layers=[nn.Conv2d, nn.Linear], log=False)
                    net = qs.convert(net, qs.quantize(bits=8, timeout=1), weight_layers=[nn.Conv2d, nn.Linear], log=False)
                net = net.to(device).to(memory_format=torch.channels_last).train()
                opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

                def step(_=0):
                    opt.zero_grad(set_to_none=True)
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        loss = F.cross_entropy(net(x), y)
                    loss.backward()
                    opt.step()

                for _ in range(6):
                    step()
                row[mode + "_ms"] = round(_timed_loop(step, steps), 3)
                del net, opt
            finally:
                qs.set_qsparse_options(batch_weights=True)
            torch.cuda.empty_cache()
        row["multi_tensor_over_plain"] = round(row["multi_tensor_ms"] / row["plain_ms"], 4)
        row["layer_by_layer_over_plain"] = round(row["layer_by_layer_ms"] / row["plain_ms"], 4)
        out[f"{arch}_b{batch}"] = row
    return out


def token_major_site(device, steps=40):
    """SURVEY 8f widened (round 6): the pair on a TOKEN-MAJOR activation -- (B, T, C) = 256 x 197 x 3072 bf16, the hidden activation of
    a ViT-B MLP block, prune(0.75, dimensions={2}) -> quantize(4b) behind a folded nn.ReLU, training step with live statistics
    through the composite site calls (qs_site_plan layout 3: qs_token_stats, select, apply).  Algorithmic bytes per element:
    statistics 2 + forward 2 + 4 + 1/8 (x, y, gate bitmap) + backward 4 + 1/8 + 2 (g, gate, gx) = 14.25."""
    import qsparse_amd as qs
    from qsparse_amd.fused import fuse_prune_quantize_pairs

    shape = (256, 197, 3072)
    g = torch.Generator(device=device).manual_seed(3)
    x = (torch.randn(shape, generator=g, device=device) * torch.linspace(0.25, 4.0, shape[2], device=device)).to(torch.bfloat16).requires_grad_(True)
    gout = torch.randn(shape, generator=g, device=device)
    site = fuse_prune_quantize_pairs(nn.Sequential(
        nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={2}, start=0, interval=1, repetition=1)),
        qs.quantize(bits=4, channelwise=-1, timeout=1)).to(device).train())

    def step(_=0):
        torch.autograd.grad(site(x), x, gout)

    for _ in range(10):
        step()
    ms = _steady_ms(step, steps)
    n = x.numel()
    return {"workload": "256x197x3072 bf16 (B, T, C): relu -> prune(0.75, {2}) -> quantize(4b) train fwd+bwd, live mask+scale",
            "ms_per_step": round(ms, 4), "Gelem/s": round(n / ms / 1e6, 1), "bytes_per_elem": 14.25,
            "frac_of_hbm_peak": round(14.25 * n / ms / 1e6 / HBM_PEAK_GBS, 4), "kept_channel_fraction": round(site[0][1].mask.float().mean().item(), 4)}


def token_net(device, batch=128, steps=10):
    """SURVEY 8f widened (late round 6): a whole token-major network -- examples/models.py::TokenNet (patch embedding, 12 blocks of
    LayerNorm -> Linear(768, 3072) -> nn.GELU -> Linear + residual; 196 tokens) plain against the --pq recipe on its hidden activations
    (prune(0.75, dimensions={2}) + 4-bit quantization behind the nn.GELU, 4-bit Linear / Conv2d weights), bf16 autocast, SGD.  The
    sites sit behind an activation the kernels do not fold: identity fold for the autocast image, GELU backward in the site's
    backward kernel (`act_backward`, ABI v26)."""
    import qsparse_amd as qs
    from examples.models import TokenNet, convert_pq_tokens
    from qsparse_amd.fused import ROUTES

    x = torch.randn(batch, 3, 224, 224, device=device)
    y = torch.randint(0, 1000, (batch,), device=device)

    def measure(pq):
        torch.manual_seed(0)
        net = TokenNet(num_classes=1000, dim=768, hidden=3072, depth=12, patch=16, act=nn.GELU)
        if pq:
            net = convert_pq_tokens(net, act=nn.GELU, sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
        net = net.to(device).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

        def step(_=0):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(net(x).float(), y)
            loss.backward()
            opt.step()

        for _ in range(8):
            step()
        return _steady_ms(step, steps)

    plain = measure(False)
    before = dict(ROUTES)
    pq = measure(True)
    routes = {k: ROUTES[k] - before.get(k, 0) for k in ROUTES if ROUTES[k] - before.get(k, 0)}
    torch.cuda.empty_cache()
    variants = {}
    for key, label, opts, restore in (
            ("opt_in_extensions", "preserve_dtype=True, elide_pruned='all'", dict(preserve_dtype=True, elide_pruned="all"),
             dict(preserve_dtype=False, elide_pruned="forward")),
            ("without_identity_fold_and_act_backward", "round 6 before the GELU work: no image for sites behind nn.GELU, ATen's gelu_backward pass",
             dict(act_backward=False), dict(act_backward=True))):
        try:
            qs.set_qsparse_options(**opts)
            if key.startswith("without"):
                import qsparse_amd.fused as _f
                _f._IDENTITY_FOLD = False
            v = measure(True)
            variants[key] = {"options": label, "pq_ms": round(v, 3), "pq_over_plain": round(v / plain, 4)}
        except Exception as e:      # noqa: BLE001  (recorded verbatim in the JSON line)
            variants[key] = {"options": label, "error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            qs.set_qsparse_options(**restore)
            import qsparse_amd.fused as _f
            _f._IDENTITY_FOLD = os.environ.get("QS_NO_IDENTITY_FOLD", "0") != "1"
            torch.cuda.empty_cache()
    return {"variants": variants, "model": "TokenNet dim 768 / hidden 3072 / depth 12 / patch 16 (196 tokens), nn.GELU sites", "input_shape": [batch, 3, 224, 224],
            "dtype": "bf16 autocast, fp32 master weights", "optimizer": "SGD momentum 0.9", "steps": steps, "plain_ms": round(plain, 3),
            "pq_ms": round(pq, 3), "pq_over_plain": round(pq / plain, 4),
            "routes": routes}


def extra_configs(device, only=None):
    """configs 2-4 of BASELINE.json; a failure in one of them is recorded, it never costs the headline line"""
    out = {}
    for name, fn in (("host_overhead_per_site", lambda: host_overhead(device)),
                     ("weights_pruned_quantized", lambda: weights_pruned_config(device)),
                     ("token_major_site_256x197x3072", lambda: token_major_site(device)),
                     ("token_net_gelu_b128", lambda: token_net(device)),
                     ("config2_quantize8_256x64x56x56", lambda: config2(device)),
                     ("config3_resnet18_cifar_b128", lambda: resnet_config("resnet18", 128, device, 10)),
                     ("config4_resnet50_imagenet_b256", lambda: resnet_config("resnet50", 256, device, 5))):
        if only and not any(name.startswith(o) for o in only):
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001  (recorded verbatim in the JSON line)
            out[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            torch.cuda.empty_cache()
        out[name]["bench_seconds"] = round(time.perf_counter() - t0, 1)
        # the same network as a DDP rank runs it (statistics exchange live), in a child process
        arch = {"config3": "resnet18", "config4": "resnet50"}.get(name.split("_")[0])
        if arch and "error" not in out[name] and os.environ.get("QS_BENCH_NO_EXCHANGE_LIVE", "0") != "1":
            torch.cuda.empty_cache()
            out[name]["exchange_live"] = exchange_live_in_child(arch)
            ex = out[name]["exchange_live"]
            if "pq_ddp_exchange_live_ms" in ex:
                ex["over_no_group_eager"] = round(min(ex["pq_ddp_exchange_live_ms"], ex["pq_ddp_exchange_live_ms_2"]) / out[name]["pq_ms"], 4)
    return out


def config5_warm(device, rank):
    """rank-local, collective-free warm-up of config 5: one training step of the plain and of the converted ResNet-50 at the
    config's batch, so that MIOpen's algorithm search and kernel compilation (69.6 s on a fresh box at N = 1, more with eight
    ranks tuning side by side) happen BEFORE the deadline of the collective phase is armed.  The statistics exchange is
    switched off here -- nothing in this function may wait for another rank."""
    import qsparse_amd as qs
    from examples.models import convert_pq, resnet50

    batch = int(os.environ.get("QS_BENCH_DDP_BATCH", "256"))
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    x = torch.randn((batch, 3, 224, 224), generator=g, device=device).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (batch,), generator=g, device=device)
    before = qs.get_qsparse_option("sync_statistics")
    qs.set_qsparse_options(sync_statistics=False)
    try:
        for pq in (False, True):
            torch.manual_seed(0)
            model = resnet50(1000, False)
            if pq:
                model = convert_pq(model, sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
            model = model.to(device).to(memory_format=torch.channels_last).train()
            for _ in range(3 if pq else 2):        # (inactive -> live -> steady: every kernel variant the timed steps use)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = F.cross_entropy(model(x).float(), y)
                loss.backward()
            torch.cuda.synchronize()
            del model, loss
            torch.cuda.empty_cache()
    finally:
        from qsparse_amd import util as _u
        _u._options_["sync_statistics"] = before      # (None = auto is the default; set_qsparse_options(None) would leave False)
        _u._options_epoch[0] += 1


def config5(device, world, rank, steps=5):
    """BASELINE config 5 (N > 1 only): full-width ResNet-50 on synthetic ImageNet-shape data, 4-bit weights and
    activations + 75 % channel pruning (the --pq recipe), bf16 autocast, channels_last, one process per GPU under
    `DistributedDataParallel` -- gradients follow DDP's bucketed all-reduce, masks and scales the library's C-sized
    statistics exchange.  Batch per GPU fixed (weak scaling); images/s is the WHOLE job (all ranks), time = max over
    ranks.  Every rank calls this (it is collective); rank 0's return value goes into the JSON line."""
    import qsparse_amd as qs
    from examples.models import convert_pq, resnet50

    batch = int(os.environ.get("QS_BENCH_DDP_BATCH", "256"))
    shape = (batch, 3, 224, 224)
    g = torch.Generator(device=device).manual_seed(1000 + rank)        # every rank its own shard
    x = torch.randn(shape, generator=g, device=device).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (batch,), generator=g, device=device)
    out = {"model": "resnet50", "input_shape_per_gpu": list(shape), "world": world, "dtype": "bf16 autocast, fp32 master weights",
           "layout": "channels_last", "optimizer": "SGD momentum 0.9", "steps": steps,
           "parallelism": f"dp{world} (DistributedDataParallel, bucketed gradient all-reduce; one 2C-float statistics "
                          f"all-gather per operator site and step)"}

    def run(pq):
        torch.manual_seed(0)                                            # identical initial weights on every rank
        model = resnet50(1000, False)
        if pq:
            model = convert_pq(model, sparsity=0.75, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)
        model = model.to(device).to(memory_format=torch.channels_last).train()
        net = nn.parallel.DistributedDataParallel(model, device_ids=[device.index])
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)

        def step():
            opt.zero_grad(set_to_none=False)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = F.cross_entropy(net(x).float(), y)
            loss.backward()
            opt.step()

        for _ in range(8):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = t.item() / steps * 1e3
        same = None
        if pq:      # the point of the statistics exchange: every rank ends with the same masks and scales
            sd = model.state_dict()
            digest = torch.stack([v.double().sum() for k, v in sd.items()
                                  if k.endswith(("mask", "quantize.weight", "1.weight", "magnitude"))]).sum()
            lo, hi = digest.clone(), digest.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = bool((lo == hi).item())
        del net, model, opt
        torch.cuda.empty_cache()
        return ms, same

    plain_ms, _ = run(False)
    pq_ms, same = run(True)
    out.update({"plain_ms": round(plain_ms, 3), "plain_images_per_s": round(world * batch / plain_ms * 1e3, 1),
                "pq_ms": round(pq_ms, 3), "pq_images_per_s": round(world * batch / pq_ms * 1e3, 1),
                "pq_over_plain": round(pq_ms / plain_ms, 4), "operator_state_identical_across_ranks": same})
    return out


# ---------------------------------------------------------------------------------------------------
def self_launch_command(n, argv, port=None):
    """the command `python bench.py --gpus N ...` re-issues itself as when no launcher set WORLD_SIZE: one rank per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), a free port unless given"""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """run the N ranks as a child process group and relay rank 0's ONE JSON line; returns the launcher's exit code.
    The launcher's and the ranks' other output (torchrun banners, warnings) goes to stderr, so stdout stays one line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    child = subprocess.Popen(self_launch_command(n, argv), stdout=subprocess.PIPE, env=env, cwd=ROOT)
    try:
        for raw in child.stdout:
            line = raw.decode(errors="replace")
            if line.lstrip().startswith("{"):
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
        return child.wait()
    except BaseException:
        child.terminate()          # the exact child this process started (torchrun forwards the signal to its ranks)
        child.wait()
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2-4 (N == 1) / config 5 (N > 1)")
    ap.add_argument("--no-variants", action="store_true", help="skip the other elision modes after the timed region")
    ap.add_argument("--configs-only", default=None, metavar="NAMES",
                    help="run only these BASELINE configs (comma-separated prefixes, e.g. config2,config4) and print "
                         "{'configs': ...}: the command the per-config profiles under profiles/ are taken with")
    ap.add_argument("--elide", default=None, choices=["off", "forward", "all"],
                    help="headline mode (default: the library's default, 'forward')")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="THREADS",
                    help="only time the CPU oracle with this many threads (no GPU needed)")
    ap.add_argument("--exchange-live", default=None, choices=["resnet18", "resnet50"],
                    help="(child mode of the N = 1 bench) the config-3 / config-4 network in a one-rank RCCL group with the "
                         "statistics exchange live, under DistributedDataParallel; prints its record")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(threads=args.cpu_baseline_only)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  Nothing above touched the
        # GPU (importing torch does not), and nothing below does in THIS process: the ranks are children.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.exchange_live:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)                  # RCCL / MIOpen banners go to stderr; the record is the only line on stdout
        torch.cuda.set_device(0)
        import qsparse_amd as qs
        qs.set_qsparse_options(log_on_created=False, log_during_train=False)
        rec = exchange_live_config(args.exchange_live, torch.device("cuda", 0))
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        os.close(real_stdout)
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
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}: pass --gpus {world}, or run "
                         f"`python bench.py --gpus {args.gpus}` without a launcher (it starts its own ranks)")

    import qsparse_amd as qs
    from qsparse_amd import _hip

    _hip.load()
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    if args.configs_only:
        assert world == 1, "--configs-only is a single-GPU mode"
        rec = {"configs": extra_configs(device, only=args.configs_only.split(","))}
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        os.close(real_stdout)
        return
    if args.elide:
        qs.set_qsparse_options(elide_pruned=args.elide)
    mode = qs.get_qsparse_option("elide_pruned")
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
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if world > 1 or force_exchange:       # "did the collective library see N ranks": an all-reduce of ones, in the record
        ones = torch.ones(1, device=device, dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())

    # the other two elision modes, outside the timed region, on rank 0's clock (continuity with round 1: "off" is the
    # dense 14 B/elem step; "all" also elides the backward, which then writes +0.0 for the reference's -0.0)
    variants = {}
    if world == 1 and not args.no_variants:
        for other in ("off", "forward", "all"):
            if other == mode:
                continue
            qs.set_qsparse_options(elide_pruned=other)
            for _ in range(5):
                step()
            k = max(min(args.steps // 2, 60), 1)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            variants[other] = (time.perf_counter() - tv) / k * 1e3
        qs.set_qsparse_options(elide_pruned=mode)
        # the steady state of the reference's own recipe (devise_layerwise_pruning_schedule, sparse.py:343-359): the mask
        # froze when the callback's t passed stop_mask_refresh, the scale still follows the data every step
        cb = pair[0][1].callback
        stop = cb.stop_mask_refresh
        cb.stop_mask_refresh = 0
        try:
            for _ in range(5):
                step()
            k = max(min(args.steps // 2, 60), 1)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            variants["frozen_mask"] = (time.perf_counter() - tv) / k * 1e3
        finally:
            cb.stop_mask_refresh = stop

    numel = x.numel()
    if rank == 0:
        kept = pair[0][1].mask.float().mean().item()
        fwd_bpe = {"off": 6.0}.get(mode, 2.0 * kept + 4.0)        # apply forward: bf16 in (kept rows only), fp32 out
        bwd_bpe = 4.0 * kept + 2.0 if mode == "all" else 6.0       # apply backward: fp32 in, bf16 out
        step_bpe = 2.0 + fwd_bpe + bwd_bpe
        ms_step = elapsed / args.steps * 1e3
        value = world * numel * args.steps / elapsed / 1e9
        avg = {k: sum(v) / len(v) for k, v in events.items() if v}
        kern = {}
        for name, key, bpe in (("apply_fwd", "quant_scaler_fwd+mask", fwd_bpe), ("apply_bwd", "quant_ste_bwd+mask", bwd_bpe),
                               ("stats", "mean_dim+absmax", 2.0)):
            ms = avg.get(key)
            if ms:
                kern[name] = {"ms": round(ms, 4), "bytes_per_elem": round(bpe, 4), "GB/s": round(bpe * numel / ms / 1e6, 1),
                              "frac": round(bpe * numel / ms / 1e6 / HBM_PEAK_GBS, 4)}
        # dominant kernel: the longest-running one
        dom_name = max((k for k in ("apply_fwd", "apply_bwd") if k in kern), key=lambda k: kern[k]["ms"], default=None)
        roof = {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None}
        if dom_name:
            traffic, traffic_src = pmc_traffic(dom_name)
            roof.update({"kernel": {"apply_fwd": "quant_scaler_fwd+mask", "apply_bwd": "quant_ste_bwd+mask"}[dom_name],
                         "achieved": kern[dom_name]["GB/s"], "frac": kern[dom_name]["frac"], "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(round(kern[dom_name]["bytes_per_elem"] * numel)),
                         "kernels": kern})
        out = {
            "metric": "Gelem/s quantize+prune fwd+bwd, 256×256×56×56 bf16; % HBM roofline",   # BASELINE.json's string
            "value": round(value, 3), "unit": "Gelem/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # (the driver's parser keeps flat scalar keys and the first 128 characters of a string: shape first, short)
            "config": {"workload": "256x256x56x56 bf16 per GPU: prune(0.75,{1})->quantize(4b) train fwd+bwd, live mask+scale",
                       "shape": "x".join(str(d) for d in SHAPE), "io_dtypes": "bf16 x, f32 y, f32 grad_y, bf16 grad_x",
                       "shape_per_gpu": list(SHAPE), "fused": True, "elide_pruned": mode,
                       "kept_channel_fraction": round(kept, 4),
                       "step_bytes_per_elem": round(step_bpe, 4), "dense_step_bytes_per_elem": 14.0,
                       "algorithmic_bytes_per_elem": {"step": round(step_bpe, 4), "stats": 2.0, "apply_fwd": round(fwd_bpe, 4),
                                                      "apply_bwd": round(bwd_bpe, 4), "dense_step": 14.0},
                       "step_frac_of_hbm_peak": round(step_bpe * numel / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": roof,
        }
        for kname, rec in kern.items():            # flat copies of the per-kernel figures (nested objects do not survive the parser)
            roof[kname + "_ms"], roof[kname + "_frac"], roof[kname + "_bytes_per_elem"] = rec["ms"], rec["frac"], rec["bytes_per_elem"]
        if variants:
            cfg = out["config"]
            for m, key in (("off", "dense"), ("all", "elide_all"), ("forward", "elide_forward"), ("frozen_mask", "frozen_mask")):
                if m in variants:
                    cfg[key + "_ms_per_step"] = round(variants[m], 4)
                    cfg[key + "_gelem_s"] = round(numel / variants[m] / 1e6, 1)
            if "off" in variants:       # the dense step, 14 B/elem: the figure comparable with BASELINE.md's byte accounting
                cfg["dense_frac_of_hbm_peak"] = round(14.0 * numel / (variants["off"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            out["config"]["variants"] = {
                m: {"ms_per_step": round(ms, 4), "Gelem/s": round(numel / ms / 1e6, 1),
                    "note": {"off": "dense: every element loaded, 14 B/elem (round-1 record)",
                             "forward": "apply forward skips pruned channels whose values are all finite this step; bit-identical for every input (library default)",
                             "all": "backward elided as well: +0.0 where the reference has -0.0 (opt-in)",
                             "frozen_mask": "default mode after stop_mask_refresh (the layerwise recipe's steady state): mask fixed, "
                                            "scale live -- statistics are the per-channel abs-max alone"}[m]}
                for m, ms in variants.items()}
        if world > 1 or force_exchange:
            out["config"]["ranks_seen"] = ranks_seen
            out["config"]["backend"] = dist.get_backend()
            out["config"]["exchange"] = ("one all-gather of a 2C-float record per step over " +
                                         ("gloo (shared GPU, development)" if share_gpu else "RCCL") +
                                         (" in a ONE-rank group (QS_BENCH_FORCE_EXCHANGE)" if force_exchange else ""))
    import threading
    emit_lock, emitted = threading.Lock(), [False]

    def emit():
        """rank 0 writes its ONE line exactly once, whoever gets here first (main thread or the watchdog below)"""
        with emit_lock:
            if rank == 0 and not emitted[0]:
                emitted[0] = True
                os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if world > 1 and not args.no_configs and os.environ.get("QS_BENCH_NO_DDP_CONFIG", "0") != "1":
        # BASELINE config 5 after the headline's timed region.  It is collective, so a rank that fails alone would leave
        # the others waiting in RCCL: a watchdog on every rank ends the process cleanly at the deadline -- rank 0 after
        # writing the headline record with the failure noted -- so config 5 can never cost the headline line.
        del pair, x, gout
        torch.cuda.empty_cache()
        limit = float(os.environ.get("QS_BENCH_DDP_TIMEOUT", "480"))
        warm_limit = float(os.environ.get("QS_BENCH_DDP_WARM_TIMEOUT", "600"))
        t5 = time.perf_counter()
        phase = ["warm-up (MIOpen search, rank-local)", warm_limit]

        def bark():
            # a rank failed or hung inside a collective: the headline (measured before config 5 started) is still written,
            # marked `degraded`, and EVERY rank leaves with a non-zero code so that torchrun and the harness see the failure
            with emit_lock:
                if rank == 0 and "configs" not in out:
                    out["degraded"] = True
                    out["configs"] = {"config5_resnet50_ddp": {"error": f"{phase[0]} did not finish within {phase[1]:.0f} s (a rank "
                                                                        f"failed or hung); the headline record is unaffected"}}
            emit()
            os._exit(3)

        watchdog = threading.Timer(warm_limit, bark)
        watchdog.daemon = True
        watchdog.start()
        warm_s = None
        try:
            # phase 1, no collectives: every rank tunes and compiles on its own; the deadline of the collective phase is armed
            # only when all ranks are through (the barrier is inside phase 1's deadline)
            config5_warm(device, rank)
            torch.cuda.synchronize()
            dist.barrier()
            warm_s = round(time.perf_counter() - t5, 1)
            watchdog.cancel()
            phase[:] = ["the DDP phase", limit]
            watchdog = threading.Timer(limit, bark)
            watchdog.daemon = True
            watchdog.start()
            rec5 = config5(device, world, rank)
        except Exception as e:      # noqa: BLE001  (recorded verbatim in the JSON line)
            rec5 = {"error": f"{type(e).__name__}: {e}"[:400]}
        rec5["warmup_seconds"] = warm_s
        rec5["bench_seconds"] = round(time.perf_counter() - t5, 1)
        if rank == 0:
            with emit_lock:
                out["configs"] = {"config5_resnet50_ddp": rec5}
                if "error" in rec5:
                    out["degraded"] = True
        emit()                      # before the group is torn down: a hang in there ends at the watchdog, line already out
    if world > 1 or force_exchange:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not force_exchange and not args.no_configs:
            del pair, x, gout
            torch.cuda.empty_cache()
            out["configs"] = extra_configs(device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    emit()
    os.close(real_stdout)


if __name__ == "__main__":
    main()
