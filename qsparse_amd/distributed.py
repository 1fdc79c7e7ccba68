"""Data-parallel consistency of the operator state.

The path shards by batch: every element-wise kernel is rank-local and needs no collective.  The only
exchange is the C-sized statistics that drive the mask and the scale -- per-channel mean|x| (SUM, then
divide by the world size: equal shards make the mean of means the global mean), per-channel / tensor
abs-max (MAX) and min/max (MIN/MAX).  The reference has no distributed code at all (its masks and
scales silently diverge per rank because they are ``requires_grad=False`` parameters that DDP neither
reduces nor broadcasts); with one process this module is a no-op, so single-GPU parity is untouched.

Collectives go through ``torch.distributed`` (backend "nccl" == RCCL over xGMI on MI355X, "gloo" on
CPU).  Enabled automatically when a process group with world size > 1 exists; switch off with
``set_qsparse_options(sync_statistics=False)``.  ``sync_statistics="always"`` runs the collectives even in a
process group of one rank -- the way to exercise the RCCL path on a single-GPU machine.
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


def exchange_active(world: Optional[int] = None) -> bool:
    world = world or stats_world_size()
    return world > 1 or (get_option("sync_statistics") == "always" and dist.is_available() and dist.is_initialized())


def allreduce_mean(t: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    """mean over ranks of a small statistics tensor, in fp32 (returns a new tensor; float64 statistics stay float64)."""
    world = world or stats_world_size()
    if not exchange_active(world):
        return t
    # (float64 statistics -- a float64 tensor's importance, `_hip.on_hip` -- keep their precision; everything else is fp32)
    buf = t.detach().to(torch.float64 if t.dtype == torch.float64 else torch.float32).contiguous().clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf / world


def allreduce_max_(t: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    world = world or stats_world_size()
    if exchange_active(world):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def allreduce_min_(t: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    world = world or stats_world_size()
    if exchange_active(world):
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t


# `ProcessGroup._allgather_base` is what all_gather_into_tensor itself ends in on the torch versions this package was measured
# with (2.10); anything else takes the public call
_PRIVATE_ALLGATHER = tuple(int(v) for v in torch.__version__.split("+")[0].split(".")[:2] if v.isdigit()) in ((2, 9), (2, 10), (2, 11))


def all_gather_records(gathered: torch.Tensor, record: torch.Tensor):
    """the one collective of a data-parallel site step: every rank's 2C-float record into `gathered` ([world * 2C], rank
    order).  RCCL over xGMI is point-to-point and a 2 KB all-gather is latency-bound at any world size: ONE collective per
    site is what matters (the records hold importance and abs-max together)."""
    global _PRIVATE_ALLGATHER
    if _PRIVATE_ALLGATHER:
        # the process group's own entry point: the public wrapper's argument checks cost ~7 us per call, seventeen times a step.
        # A private name: any surprise (another torch, a wrapped or fake process group, a changed signature) switches to the
        # public call for good -- BEFORE anything was exchanged, so the step's statistics are still those of this rank alone
        try:
            work = dist.group.WORLD._allgather_base(gathered, record)
        except (TypeError, AttributeError, NotImplementedError, RuntimeError):
            _PRIVATE_ALLGATHER = False
        else:
            work.wait()
            return
    dist.all_gather_into_tensor(gathered, record)


def gather_pair_statistics(stage: Optional[torch.Tensor], chan_absmax: Optional[torch.Tensor], world: int,
                           record: Optional[dict] = None) -> torch.Tensor:
    """GPU half of `sync_pair_statistics` without its last step: pack the rank's record (unless the last statistics
    launch has already written it, `record["filled"]`), all-gather -- ONE collective -- and hand the [world, 2C]
    records to `qs_pq_select`, which combines them in rank order itself."""
    from qsparse_amd import _hip
    ref = stage if stage is not None else chan_absmax
    C = stage.numel() if stage is not None else chan_absmax.shape[0]
    if record is not None and record["filled"]:
        rec = record["buf"]
    else:
        rec = _hip.stats_pack(stage, chan_absmax, C, ref.device)
    gathered = torch.empty(world * 2 * C, dtype=torch.float32, device=ref.device)
    dist.all_gather_into_tensor(gathered, rec)
    return gathered


def sync_pair_statistics(stage: Optional[torch.Tensor], chan_absmax: Optional[torch.Tensor],
                         world: int) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """the exchange of the fused prune->quantize step as ONE collective: every rank packs (importance, abs-max)
    into a 2C-float record, the records are all-gathered and combined in rank order on every rank (mean / max),
    so all ranks end with bit-identical statistics.  GPU tensors: pack/combine are HIP kernels."""
    ref = stage if stage is not None else chan_absmax
    if ref is None or not exchange_active(world):
        return stage, chan_absmax
    C = stage.numel() if stage is not None else chan_absmax.shape[0]   # chan_absmax: [C] or line-padded [C, 32]
    if ref.is_cuda:
        from qsparse_amd import _hip
        rec = _hip.stats_pack(stage, chan_absmax, C, ref.device)
        gathered = torch.empty(world * 2 * C, dtype=torch.float32, device=ref.device)
        dist.all_gather_into_tensor(gathered, rec)
        new_stage = _hip.stats_combine(gathered, world, C, stage is not None, chan_absmax)
        return new_stage, chan_absmax
    from qsparse_amd._hip import amax_values
    rec = torch.cat([stage.detach().float().view(-1) if stage is not None else torch.zeros(C),
                     amax_values(chan_absmax) if chan_absmax is not None else torch.zeros(C)])
    parts = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(parts, rec)
    g = torch.stack(parts)
    if stage is not None:
        acc = torch.zeros(C)
        for r in range(world):
            acc = acc + g[r, :C]
        stage = acc / world
    if chan_absmax is not None:
        amax_values(chan_absmax).copy_(g[:, C:].amax(0))
    return stage, chan_absmax
