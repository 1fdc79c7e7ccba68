"""world_size-2 gloo test of the data-parallel statistics exchange: with the exchange the
mask/scale state of both ranks is identical and equals the single-process state on the concatenated
batch statistics; without it the ranks drift (what the reference does)."""
import os
import socket

import pytest
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
    out.put((rank, p.mask.clone(), p.callback.magnitude.clone(), q.weight.clone(), a.weight.clone()))
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
    return res


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
    for fused in (False, True):
        pair = nn.Sequential(nn.Sequential(nn.Identity(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1,
                                                                    repetition=2)),
                             qs.quantize(bits=4, channelwise=-1, timeout=1)).cuda().train()
        if fused:
            fuse_prune_quantize_pairs(pair)
        for step in range(6):
            g = torch.Generator().manual_seed(100 + step)
            full = (torch.randn(8, 16, 8, 8, generator=g) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).bfloat16()
            shard = (full[rank * 4:(rank + 1) * 4].float() * (1.0 + 0.5 * rank)).bfloat16().cuda().requires_grad_(True)
            y = pair(shard)
            y.backward(torch.ones_like(y))
        res[fused] = tuple(t.detach().cpu().clone() for t in (pair[0][1].mask, pair[0][1].callback.magnitude, pair[1].weight))
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
    for fused in (False, True):
        for a, b in zip(res[0][fused], res[1][fused]):
            assert torch.equal(a, b), fused                      # ranks agree
    for a, b in zip(res[0][False], res[0][True]):
        assert torch.equal(a, b)                                  # fused exchange == unfused exchange
