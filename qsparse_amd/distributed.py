"""Data-parallel consistency of the operator state.

The path shards by batch: every element-wise kernel is rank-local and needs no collective.  The only
exchange is the C-sized statistics that drive the mask and the scale -- per-channel mean|x| (SUM, then
divide by the world size: equal shards make the mean of means the global mean), per-channel / tensor
abs-max (MAX) and min/max (MIN/MAX).  The reference has no distributed code at all (its masks and
scales silently diverge per rank because they are ``requires_grad=False`` parameters that DDP neither
reduces nor broadcasts); with one process this module is a no-op, so single-GPU parity is untouched.

Collectives go through ``torch.distributed`` (backend "nccl" == RCCL over xGMI on MI355X, "gloo" on
CPU).  Enabled automatically when a process group with world size > 1 exists; switch off with
``set_qsparse_options(sync_statistics=False)``.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from qsparse_amd.util import get_option


def stats_world_size() -> int:
    if get_option("sync_statistics") is False:
        return 1
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size()


def allreduce_mean(t: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    """mean over ranks of a small statistics tensor, in fp32 (returns a new fp32 tensor)."""
    world = world or stats_world_size()
    if world <= 1:
        return t
    buf = t.detach().to(torch.float32).contiguous().clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf / world


def allreduce_max_(t: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    world = world or stats_world_size()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def allreduce_min_(t: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    world = world or stats_world_size()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t


def sync_pair_statistics(stage: Optional[torch.Tensor], chan_absmax: Optional[torch.Tensor],
                         world: int) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """the two exchanges of the fused prune->quantize step, issued back to back (asynchronously) so that
    their latencies overlap; both are C-sized (<= 8 KB)."""
    works = []
    buf = None
    if stage is not None:
        buf = stage.detach().to(torch.float32).contiguous().clone()
        works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
    if chan_absmax is not None:
        works.append(dist.all_reduce(chan_absmax, op=dist.ReduceOp.MAX, async_op=True))
    for w in works:
        w.wait()
    if buf is not None:
        stage = buf / world
    return stage, chan_absmax
