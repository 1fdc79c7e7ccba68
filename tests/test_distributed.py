"""world_size-2 gloo test of the data-parallel statistics exchange: with the exchange the
mask/scale state of both ranks is identical and equals the single-process state on the concatenated
batch statistics; without it the ranks drift (what the reference does)."""
import os
import socket

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sync, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qsparse_amd as qs
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, sync_statistics=sync)
    torch.manual_seed(0)
    p = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2)
    q = qs.quantize(bits=4, channelwise=-1, timeout=1)
    a = qs.quantize(bits=8, channelwise=1, timeout=1, callback=qs.AdaptiveQuantizer())
    for step in range(6):
        g = torch.Generator().manual_seed(100 + step)
        full = torch.randn(8, 12, 5, 5, generator=g) * torch.linspace(0.3, 3, 12).view(1, -1, 1, 1)
        shard = full[rank * 4:(rank + 1) * 4] * (1.0 + 0.5 * rank)     # ranks see different data
        q(p(shard))
        a(shard)
    # by value (numpy): a tensor travels as a file descriptor the parent fetches from THIS process, which may have exited by then
    out.put((rank,) + tuple(t.detach().numpy().copy() for t in (p.mask, p.callback.magnitude, q.weight, a.weight)))
    dist.barrier()
    dist.destroy_process_group()


def _run(sync):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sync, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([out.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return [(r[0],) + tuple(torch.from_numpy(a) for a in r[1:]) for r in res]


def test_statistics_exchange_keeps_ranks_identical():
    r0, r1 = _run(True)
    for a, b in zip(r0[1:], r1[1:]):
        assert torch.equal(a, b)
    assert r0[1].sum().item() == 6          # 50 % of 12 channels kept


def test_without_exchange_ranks_drift_like_the_reference():
    r0, r1 = _run(False)
    assert not torch.equal(r0[2], r1[2]) and not torch.equal(r0[3], r1[3])


def _gpu_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share cuda:0 on a 1-GPU box
    import torch.nn as nn
    import qsparse_amd as qs
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    res = {}
    # nchw / channels_last: the last statistics launch writes the exchange record itself; 2-d and token-major: qs_stats_pack does
    per = 4 if world <= 2 else 2          # samples per rank (the composite serves batches of at least two)
    for layout in ("nchw", "channels_last", "2d", "token"):
        for fused in (False, True):
            pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.5, dimensions={2 if layout == "token" else 1}, start=1,
                                                                        interval=1, repetition=2)),
                                 qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train()
            if fused:
                fuse_prune_quantize_pairs(pair)
            for step in range(6):
                g = torch.Generator().manual_seed(100 + step)
                full = (torch.randn(world * per, 16, 8, 8, generator=g) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16()
                shard = (full[rank * per:(rank + 1) * per].float() * (1.0 + 0.5 * (rank % 3))).bfloat16().cuda()
                if layout == "channels_last":
                    shard = shard.contiguous(memory_format=torch.channels_last)
                elif layout == "2d":
                    shard = shard[:, :, 0, 0].contiguous()
                elif layout == "token":       # (B, T, C) with the mask on the last dim: qs_site_plan layout 3
                    shard = shard.flatten(2).transpose(1, 2).contiguous()
                shard.requires_grad_(True)
                y = pair(shard)
                y.backward(torch.ones_like(y))
            res[(layout, fused)] = tuple(t.detach().cpu().numpy().copy()     # numpy: no fd hand-over after the worker's exit
                                         for t in (pair[0][1].mask, pair[0][1].callback.magnitude, pair[1].weight))
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_fused_pair_exchange_on_gpu_two_ranks():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(out.get(timeout=300) for _ in procs)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    _check_exchange_results(res, 2)


def _check_exchange_results(res, world):
    for layout in ("nchw", "channels_last", "2d", "token"):
        for fused in (False, True):
            for r in range(1, world):
                for a, b in zip(res[0][(layout, fused)], res[r][(layout, fused)]):
                    assert np.array_equal(a, b), (layout, fused, r)      # ranks agree
            assert 0 < int(res[0][(layout, fused)][0].sum()) < 16        # a real mask
        if layout == "channels_last":
            continue    # (the unfused layers sum a channels_last shard in NCHW order after a copy: 1 ulp apart in bf16)
        for a, b in zip(res[0][(layout, False)], res[0][(layout, True)]):
            assert np.array_equal(a, b), layout                   # fused exchange == unfused exchange


@pytest.mark.gpu
def test_fused_pair_exchange_on_gpu_eight_ranks():
    """the world size the driver's scaling run ends at: eight processes (here sharing the box's one GPU over gloo) exchange their
    records every step; the select combines eight records in rank order -- all eight ranks end with the same mask, magnitude and
    scale, on every layout incl. token-major, through the composite calls and through the fine-grained route"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 8, port, out)) for r in range(8)]
    for pr in procs:
        pr.start()
    res = dict(out.get(timeout=600) for _ in procs)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    _check_exchange_results(res, 8)


def _ddp_worker(rank, world, port, out):
    """BASELINE config 5 in miniature: a converted network under DistributedDataParallel (gloo, CPU)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn.functional as F
    from torch.nn.parallel import DistributedDataParallel as DDP
    import qsparse_amd as qs
    from examples.models import MnistNet, convert_pq
    from qsparse_amd.quantize import QuantizeLayer
    from qsparse_amd.sparse import PruneLayer
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    torch.manual_seed(0)
    net = convert_pq(MnistNet(), sparsity=0.5, bits=4, prune_start=2, prune_interval=1, repetition=2, quant_timeout=2, log=False)
    ddp = DDP(net)
    opt = torch.optim.SGD(ddp.parameters(), lr=0.05)
    for step in range(7):
        g = torch.Generator().manual_seed(500 + step)
        x, y = torch.randn(8, 1, 28, 28, generator=g), torch.randint(0, 10, (8,), generator=g)
        xs, ys = x[rank * 4:(rank + 1) * 4] * (1.0 + 0.3 * rank), y[rank * 4:(rank + 1) * 4]
        opt.zero_grad()
        F.nll_loss(ddp(xs), ys).backward()
        opt.step()
    state = {}
    for name, m in qs.util.nn_module(ddp).named_modules():
        if isinstance(m, PruneLayer):
            state[name + ".mask"] = m.mask.detach().clone()
        elif isinstance(m, QuantizeLayer) and m.initted:
            state[name + ".weight"] = m.weight.detach().clone()
    state.update({"param." + k: v.detach().clone() for k, v in net.named_parameters() if v.requires_grad})
    out.put((rank, {k: v.numpy().copy() for k, v in state.items()}))    # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_converted_network_under_ddp_keeps_ranks_identical():
    """weights follow DDP's gradient all-reduce; masks and scales (requires_grad=False parameters DDP neither
    reduces nor re-broadcasts) follow the statistics exchange: after training on different shards every rank
    holds the same network."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(out.get(timeout=300) for _ in procs)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert res[0].keys() == res[1].keys() and any(k.endswith(".mask") for k in res[0]) and any(k.endswith(".weight") for k in res[0])
    import numpy as np
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
    assert any((~v).any() for k, v in res[0].items() if k.endswith(".mask"))     # something was pruned


def _rccl_worker(port, out):
    """one rank, backend nccl (= RCCL): the collectives really run, on the stream the kernels run on."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.nn as nn
    import qsparse_amd as qs
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    torch.cuda.set_device(0)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)

    def run(tag):
        pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train()
        fuse_prune_quantize_pairs(pair)
        lone_q = qs.quantize(bits=8, channelwise=-1, timeout=1).cuda().train()
        lone_a = qs.quantize(bits=8, channelwise=1, timeout=1, callback=qs.AdaptiveQuantizer()).cuda().train()
        lone_p = qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1).cuda().train()
        outs = []
        for step in range(5):
            g = torch.Generator().manual_seed(900 + step)
            x = (torch.randn(8, 16, 8, 8, generator=g) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16().cuda().requires_grad_(True)
            y = pair(x)
            y.backward(torch.ones_like(y))
            outs += [y.detach().float().cpu().numpy(), x.grad.float().cpu().numpy(), lone_q(x.detach()).float().cpu().numpy(),
                     lone_a(x.detach()).float().cpu().numpy(), lone_p(x.detach()).float().cpu().numpy()]
        outs += [pair[0][1].mask.cpu().numpy(), pair[1].weight.detach().cpu().numpy(), lone_a.weight.detach().cpu().numpy()]
        return outs

    plain = run("no process group")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    qs.set_qsparse_options(sync_statistics="always")
    with_rccl = run("rccl, one rank")
    torch.cuda.synchronize()
    dist.destroy_process_group()
    out.put((plain, with_rccl))


@pytest.mark.gpu
def test_statistics_exchange_through_rccl_with_one_rank():
    """the exchange with backend nccl (RCCL) in a one-rank group -- all-gather of the packed record, all-reduces of
    the stand-alone layers -- leaves every output and state bit-identical to the run without a process group."""
    import numpy as np
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    pr = ctx.Process(target=_rccl_worker, args=(_free_port(), out))
    pr.start()
    plain, with_rccl = out.get(timeout=600)
    pr.join(timeout=120)
    assert pr.exitcode == 0
    assert len(plain) == len(with_rccl) > 0
    for a, b in zip(plain, with_rccl):
        assert np.array_equal(a, b)


def _ddp_resnet_worker(rank, world, port, out):
    """BASELINE config 5 on four ranks: a converted ResNet-18 (narrow, CIFAR shape) under DistributedDataParallel (gloo),
    every rank with its OWN data and its own batch size."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn.functional as F
    from torch.nn.parallel import DistributedDataParallel as DDP
    import qsparse_amd as qs
    from examples.models import convert_pq, resnet18
    from qsparse_amd.quantize import QuantizeLayer
    from qsparse_amd.sparse import PruneLayer
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    torch.manual_seed(0)
    net = convert_pq(resnet18(10, True, width=8), sparsity=0.5, bits=4, prune_start=2, prune_interval=2, repetition=2,
                     quant_timeout=2, log=False)
    ddp = DDP(net)
    opt = torch.optim.SGD(ddp.parameters(), lr=0.05, momentum=0.9)
    batch = 2 + rank                                             # unequal shards: 2, 3, 4, 5 samples
    for step in range(8):                                        # the schedule ends at step 4 (start 2, interval 2, 2 ramps)
        g = torch.Generator().manual_seed(1000 * rank + step)   # a different stream per rank
        x = torch.randn(batch, 3, 32, 32, generator=g) * (1.0 + 0.4 * rank)
        y = torch.randint(0, 10, (batch,), generator=g)
        opt.zero_grad()
        F.cross_entropy(ddp(x), y).backward()
        opt.step()
    state, counters = {}, {}
    for name, m in qs.util.nn_module(ddp).named_modules():
        if isinstance(m, PruneLayer):
            state[name + ".mask"] = m.mask.detach().clone()
            state[name + ".magnitude"] = m.callback.magnitude.detach().clone()
            counters[name] = (m._n_updates.item(), round(m._cur_sparsity.item(), 6), m.callback.t.item())
        elif isinstance(m, QuantizeLayer) and m.initted:
            state[name + ".weight"] = m.weight.detach().clone()
            counters[name] = (m._n_updates.item(),)
    state.update({"param." + k: v.detach().clone() for k, v in net.named_parameters() if v.requires_grad})
    out.put((rank, {k: v.numpy().copy() for k, v in state.items()}, counters))
    dist.barrier()
    dist.destroy_process_group()


def test_converted_resnet18_under_ddp_on_four_ranks_with_unequal_shards():
    """after the sparsity schedule has completed on shards of different content and size, all four ranks hold the
    same network: weights through DDP's all-reduce, masks / magnitudes / scales through the statistics exchange
    (qsparse_amd/distributed.py), counters through the common step count."""
    import numpy as np
    world = 4
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_resnet_worker, args=(r, world, port, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [out.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    res = {r: (s, c) for r, s, c in got}
    ref_state, ref_counters = res[0]
    masks = [k for k in ref_state if k.endswith(".mask")]
    assert len(masks) >= 8 and any(k.endswith(".weight") for k in ref_state)
    for r in range(1, world):
        assert res[r][1] == ref_counters, r
        assert res[r][0].keys() == ref_state.keys()
        for k in ref_state:
            assert np.array_equal(res[r][0][k], ref_state[k]), (r, k)
    for k in masks:                                              # the schedule completed: ~50 % of the channels pruned
        kept = ref_state[k].mean()
        assert 0.4 <= kept <= 0.75, (k, kept)
    assert all(c[0] == 8 for c in ref_counters.values())


# ---- one-rank RCCL: what can be pinned about the N > 1 path without a multi-GPU node ---------------------------------
_RCCL_GRAPH_AND_COUNTER = r'''
import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
import torch, torch.nn as nn, torch.distributed as dist
import qsparse_amd as qs
from qsparse_amd import graphs
from qsparse_amd.fused import fuse_prune_quantize_pairs
torch.cuda.set_device(0)
qs.set_qsparse_options(log_on_created=False, log_during_train=False, graph_safe=True)

def make():
    pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1)),
                         qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train()
    return fuse_prune_quantize_pairs(pair)

def data(i):
    g = torch.Generator().manual_seed(700 + i)
    return ((torch.randn(8, 16, 8, 8, generator=g) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16().cuda(),
            torch.randn(8, 16, 8, 8, generator=g).cuda())

captured = []

def run(replay):
    pair = make()
    sx, sg = data(0)
    sx.requires_grad_(True)
    outs = []
    def step():
        y = pair(sx)
        (gx,) = torch.autograd.grad(y, sx, sg)
        return y, gx
    for i in range(4):
        x, g = data(i)
        sx.data.copy_(x); sg.copy_(g)
        y, gx = step()
        outs.append((y.detach().clone(), gx.clone()))
    def step_fn(x, g):
        x = x.detach().requires_grad_(True)
        y = pair(x)
        (gx,) = torch.autograd.grad(y, x, g)
        return y, gx
    gstep = graphs.GraphedStep(pair, step_fn, settle=0) if replay else None
    for i in range(4, 8):
        x, g = data(i)
        if replay:
            y, gx = gstep(x, g)
        else:
            sx.data.copy_(x); sg.copy_(g)
            y, gx = step()
        outs.append((y.detach().clone(), gx.clone()))
    if replay:
        captured.append(gstep.captured)
    torch.cuda.synchronize()
    graphs.resync_host_state(pair)
    state = [pair[0][1].mask.clone(), pair[0][1].callback.magnitude.clone(), pair[1].weight.clone(),
             pair[0][1]._n_updates.clone(), pair[1]._n_updates.clone(), pair[0][1].callback.t.clone()]
    return outs, state

def mark(m):
    print("MARK", m, file=sys.stderr, flush=True)

plain = run(False)                                   # no process group, eager: the single-process truth
plain_graph = run(True)                              # without an exchange GraphedStep does capture -- and equals eager
assert captured[-1] is True
for (ya, ga), (yb, gb) in zip(plain_graph[0], plain[0]):
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
mark("plain done")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
qs.set_qsparse_options(sync_statistics="always")     # the exchange really runs: pack -> all_gather_into_tensor (RCCL) -> combining select
eager_x = run(False)
mark("eager with exchange done")
graph_x = run(True)                                  # GraphedStep with the exchange live: must NOT capture (see graphs.steady_state)
mark("graphed-step with exchange done")
res = {"captured_with_live_exchange": captured[-1]}
for (oa, sa), tag in ((eager_x, "eager"), (graph_x, "graph")):
    for (ya, ga), (yb, gb) in zip(oa, plain[0]):
        assert torch.equal(ya, yb) and torch.equal(ga, gb), tag
    for a, b in zip(sa, plain[1]):
        assert torch.equal(a, b), tag

# a counter written by a COLLECTIVE (raw-pointer write into the parameter's storage)
qs.set_qsparse_options(graph_safe=False)
layer = qs.quantize(bits=8, channelwise=-1, timeout=5).cuda().train()
x = torch.randn(4, 8, 6, 6).cuda()
for _ in range(2):
    layer(x)                                          # n_updates = 2 < timeout: identity
mark("comparisons done")
v0 = layer._n_updates._version
dist.all_gather_into_tensor(layer._n_updates.data, torch.full((1,), 9, dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
res["collective_bumped_version"] = bool(layer._n_updates._version != v0)
y = layer(x)
res["seen_without_resync"] = bool(not torch.equal(y, x))          # quantizes iff the host mirror noticed n_updates = 9 >= 5
qs.resync_host_state(layer)
y = layer(x)
res["seen_after_resync"] = bool(not torch.equal(y, x))
res["n_updates_after"] = int(layer._n_updates.item())
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
'''


@pytest.mark.gpu
def test_rccl_one_rank_graph_capture_with_live_exchange_and_collective_counter_writes():
    """(i) a graph-safe fused step with the statistics exchange LIVE -- pack, RCCL all_gather_into_tensor, combining
    select -- equals the run without a process group, bit for bit, eagerly AND through `GraphedStep`, which refuses to
    capture while an exchange is live (an RCCL collective inside a hipGraph capture is a SIGSEGV on this stack: found by
    this very test; `graphs.steady_state` is False for data-parallel steps); (ii) a step counter written by a collective (`dist.all_gather_into_tensor(layer._n_updates.data, ...)`, what
    re-syncing ranks does): the documented contract is that `resync_host_state` makes the host mirror see it -- pinned
    here together with what happens without it (the mirror notices exactly when the write bumped the version counter)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_GRAPH_AND_COUNTER], cwd=root, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    print(res)
    assert res["captured_with_live_exchange"] is False
    assert res["seen_after_resync"] is True                      # the documented route always works
    assert res["seen_without_resync"] == res["collective_bumped_version"], res
    assert res["n_updates_after"] >= 10, res


def _mailbox_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share cuda:0; gloo only ships the 64-byte handles
    import torch.nn as nn
    import qsparse_amd as qs
    from qsparse_amd import distributed as qdist
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    res = {}
    for mode in ("collective", "mailbox"):
        qs.set_qsparse_options(sync_statistics="mailbox" if mode == "mailbox" else None)
        for layout in ("nchw", "channels_last"):
            torch.manual_seed(0)
            pair = fuse_prune_quantize_pairs(nn.Sequential(
                nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2)),
                qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train())
            # a quantize-only site (convert's Sequential(act, QuantizeLayer)): its abs-max lines travel through a mailbox as well
            actq = fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.ReLU(), qs.quantize(bits=4, channelwise=-1, timeout=1))))[0].cuda().train()
            ys = []
            for step in range(9):
                g = torch.Generator().manual_seed(100 + step)
                full = (torch.randn(8, 16, 8, 8, generator=g) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16()
                shard = (full[rank * 4:(rank + 1) * 4].float() * (1.0 + 0.5 * rank)).bfloat16().cuda()
                if layout == "channels_last":
                    shard = shard.contiguous(memory_format=torch.channels_last)
                shard.requires_grad_(True)
                y = pair(shard)
                y.backward(torch.ones_like(y))
                ys.append(y.detach().float().cpu().numpy().copy())
                ys.append(actq(shard.detach()).detach().float().cpu().numpy().copy())
            torch.cuda.synchronize()
            steps = sum(b.step for b in (qdist._mailboxes or {}).values())
            n_boxes = len(qdist._mailboxes or {})
            res[(mode, layout)] = (tuple(t.detach().cpu().numpy().copy() for t in (pair[0][1].mask, pair[0][1].callback.magnitude, pair[1].weight,
                                                                                    actq[1].weight)),
                                   ys, steps, n_boxes)
            if mode == "mailbox":
                qdist.close_mailboxes()          # (checks the status word: nobody was waited for in vain)
    qs.set_qsparse_options(sync_statistics=None)
    from qsparse_amd import util
    util._options_["sync_statistics"] = None
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_mailbox_exchange_equals_the_collective_two_ranks_one_gpu():
    """the prototype of the exchange without a host collective (include/qsparse_hip.h, "peer-mapped mailboxes";
    `set_qsparse_options(sync_statistics="mailbox")`): two processes on one GPU map each other's mailbox through hipIpc, publish
    their statistics records into it and combine them in rank order -- masks, magnitudes, scales and outputs are those of the
    all-gather route, on both ranks, step by step (rank-order combine, bit-identical on every rank)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mailbox_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(out.get(timeout=300) for _ in procs)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for layout in ("nchw", "channels_last"):
        for rank in (0, 1):
            (state_c, ys_c, steps_c, _), (state_m, ys_m, steps_m, boxes_m) = res[rank][("collective", layout)], res[rank][("mailbox", layout)]
            assert steps_c == 0 and steps_m >= 12, (steps_c, steps_m)         # the mailboxes served the live steps of the second run only
            assert boxes_m == 2                                               # ... one for the pair, one for the quantize-only site
            for a, b in zip(state_c, state_m):
                assert np.array_equal(a, b), (layout, rank)
            for s, (a, b) in enumerate(zip(ys_c, ys_m)):
                assert np.array_equal(a, b), (layout, rank, s)
        for a, b in zip(res[0][("mailbox", layout)][0], res[1][("mailbox", layout)][0]):
            assert np.array_equal(a, b), layout                              # ranks agree


def _mailbox_timeout_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["QS_MAILBOX_MAX_SPINS"] = str(1 << 16)          # ~tens of milliseconds instead of seconds (read at import)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn as nn
    import qsparse_amd as qs
    from qsparse_amd import distributed as qdist
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, sync_statistics="mailbox")
    pair = fuse_prune_quantize_pairs(nn.Sequential(
        nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2)),
        qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train())

    def step(s):
        g = torch.Generator().manual_seed(200 + s)
        x = (torch.randn(4, 16, 8, 8, generator=g) * (1.0 + rank)).bfloat16().cuda()
        return pair(x)

    res = {}
    for s in range(5):
        y = step(s)
    torch.cuda.synchronize()
    qdist.check_mailboxes()                                     # everybody published so far
    res["finite_before"] = bool(torch.isfinite(y.float()).all())
    dist.barrier()
    if rank == 0:
        # rank 1 sits this step out: rank 0's wait runs out of spins, poisons the missing record and raises the status word
        y = step(5)
        torch.cuda.synchronize()
        res["nan_in_step"] = bool(torch.isnan(y.float()).any())
        try:
            qdist.check_mailboxes()
            res["check"] = "no error"
        except qdist.MailboxTimeout as e:
            res["check"] = str(e)
        # ... and without an explicit check: the next exchanges poll the word and raise by themselves
        res["poll"] = "no error"
        try:
            for s in range(6, 10):
                step(s)
                torch.cuda.synchronize()
        except qdist.MailboxTimeout as e:
            res["poll"] = str(e)
    dist.barrier()
    try:
        qdist.close_mailboxes()
    except qdist.MailboxTimeout:
        res["close_raised"] = True
    # a captured step must be refused
    if rank == 1:
        qs.set_qsparse_options(sync_statistics="mailbox")
        box = qdist.Mailbox.__new__(qdist.Mailbox)
        box.local, box.opened = None, []
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                try:
                    qdist.Mailbox.exchange(box, torch.zeros(4, device="cuda"))
                    res["capture"] = "no error"
                except RuntimeError as e:
                    res["capture"] = str(e)
        except Exception as e:      # noqa: BLE001  (the capture itself may be invalidated by the refusal: fine)
            res.setdefault("capture", f"outer {type(e).__name__}")
    from qsparse_amd import util
    util._options_["sync_statistics"] = None
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_mailbox_timeout_is_fatal_for_the_step_and_names_the_rank():
    """ADVICE r05 (medium) / VERDICT r05 item 5: a rank that misses a peer must not keep training on a stale record -- the wait
    kernel poisons the missing record (the step's output turns NaN) and the host raises MailboxTimeout naming the rank, from an
    explicit `check_mailboxes()` and, without one, from the polling of the following exchanges; a captured exchange is refused"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mailbox_timeout_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(out.get(timeout=300) for _ in procs)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert res[0]["finite_before"] and res[1]["finite_before"]
    assert res[0]["nan_in_step"] is True
    assert "rank 1 did not publish" in res[0]["check"], res[0]
    assert "rank 1 did not publish" in res[0]["poll"], res[0]
    assert res[0].get("close_raised") is True and not res[1].get("close_raised")
    assert "cannot be captured" in res[1]["capture"], res[1]
