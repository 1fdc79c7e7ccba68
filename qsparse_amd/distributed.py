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
    return world > 1 or (get_option("sync_statistics") in ("always", "mailbox") and dist.is_available() and dist.is_initialized())


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
        # A private name: a surprise in its AVAILABILITY or SIGNATURE (another torch, a wrapped or fake process group) switches to
        # the public call for good -- such errors are raised before anything was exchanged.  A RuntimeError is not one of them: an
        # aborted communicator or an asynchronous RCCL error surfaces as that, possibly after this rank took part, and must
        # propagate instead of being answered with a second collective on this rank alone
        try:
            work = dist.group.WORLD._allgather_base(gathered, record)
        except (TypeError, AttributeError, NotImplementedError):
            _PRIVATE_ALLGATHER = False
        else:
            work.wait()
            return
    dist.all_gather_into_tensor(gathered, record)


# ----------------------------------------------------------------------------------------------------------------------
# The mailbox form of the composite site's exchange (a prototype: `set_qsparse_options(sync_statistics="mailbox")`, the
# collective stays the default).  include/qsparse_hip.h, "peer-mapped mailboxes", has the protocol; this is its host side: one
# mailbox per site and rank, the 64-byte IPC handles shipped once through the process group's object collective, then two
# launches per step and no host collective at all.  Validated with two processes sharing one GPU (tests/test_distributed.py);
# on a multi-GPU node the peers' stores travel over xGMI into fine-grained memory, which this container cannot exercise.
# ----------------------------------------------------------------------------------------------------------------------
class _RawView:
    """device memory behind a raw pointer as a tensor (the CUDA array interface torch.as_tensor understands)"""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class MailboxTimeout(RuntimeError):
    """a peer's record did not arrive within the wait kernel's bound: this rank's statistics of that step are NaN (poisoned by
    the kernel) -- training must stop"""


class _StatusWatch:
    """One status word per device for every mailbox on it, watched WITHOUT a sync: the wait kernels write 1 + the missing rank
    into it; every exchange polls -- if the previous asynchronous copy of the word into pinned host memory has landed, its value
    is looked at (no device access) and the next copy is enqueued.  A timeout is therefore raised within an exchange or two of
    the step it happened in (`check()` reads the word with a sync: end of a step, shutdown)."""

    def __init__(self, device):
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.in_flight = False

    @staticmethod
    def _raise(value):
        raise MailboxTimeout(f"mailbox exchange: rank {value - 1} did not publish its statistics in time; this rank's records of "
                             "that step were poisoned (NaN)")

    def poll(self):
        if self.in_flight:
            if not self.event.query():
                return
            self.in_flight = False
            if int(self.host[0]):
                self._raise(int(self.host[0]))
        self.host.copy_(self.status, non_blocking=True)
        self.event.record()
        self.in_flight = True

    def check(self):
        value = int(self.status.item())
        if value:
            self._raise(value)


_watches = {}


def _watch(device) -> _StatusWatch:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    w = _watches.get(key)
    if w is None:
        w = _watches[key] = _StatusWatch(device)
    return w


class Mailbox:
    # ~seconds of polling before a missing peer is reported instead of waited for (QS_MAILBOX_MAX_SPINS: tests shorten it)
    MAX_SPINS = int(__import__("os").environ.get("QS_MAILBOX_MAX_SPINS", str(1 << 24)))

    def __init__(self, n: int, device):
        import ctypes

        from qsparse_amd import _hip
        self.lib, self.n, self.world, self.rank = _hip.load(), n, dist.get_world_size(), dist.get_rank()
        self.step = 0
        self.local, self.opened = None, []
        nbytes = self.lib.qs_mailbox_bytes(self.world, n)
        if nbytes == 0:
            raise RuntimeError(f"mailbox exchange: world size {self.world} is not supported")
        local = ctypes.c_void_p()
        # (fine-grained memory or an error -- every rank must then agree to keep the collective: the error is raised, not hidden)
        _hip._check(self.lib.qs_mailbox_alloc(nbytes, ctypes.byref(local)), "qs_mailbox_alloc (fine-grained device memory)")
        self.local = local.value
        handle = ctypes.create_string_buffer(64)
        _hip._check(self.lib.qs_mailbox_export(self.local, handle), "qs_mailbox_export")
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw))         # once per site
        boxes = (ctypes.c_void_p * self.world)()
        for r, h in enumerate(handles):
            if r == self.rank:
                boxes[r] = self.local
                continue
            peer = ctypes.c_void_p()
            _hip._check(self.lib.qs_mailbox_open(ctypes.create_string_buffer(h, 64), ctypes.byref(peer)), "qs_mailbox_open")
            boxes[r] = peer.value
            self.opened.append(peer.value)
        self.boxes = boxes
        self.watch = _watch(device)
        self.status = self.watch.status
        self.views = [torch.as_tensor(_RawView(self.local + 4 * (64 + half * self.world * n), self.world * n), device=device)
                      for half in (0, 1)]
        dist.barrier()          # every mailbox is mapped before anybody publishes into one

    def exchange(self, rec: torch.Tensor) -> torch.Tensor:
        """publish this rank's record, wait for everybody's: the [world * n] records of this step, in rank order (a view of the
        local mailbox, valid until the step after next)"""
        import ctypes

        from qsparse_amd import _hip
        if torch.cuda.is_current_stream_capturing():
            # a captured step would bake this step's number into the publish / wait launches: every replay would raise the same
            # flag value and read the same half
            raise RuntimeError("the mailbox exchange cannot be captured into a hipGraph: its launches carry the step number")
        self.watch.poll()        # (raises MailboxTimeout if an earlier wait on this device ran out of spins)
        self.step += 1
        stream = _hip._stream(rec)
        _hip._check(self.lib.qs_mailbox_publish(rec.data_ptr(), self.n, self.boxes, self.world, self.rank, self.step, stream),
                    "qs_mailbox_publish")
        out = ctypes.c_void_p()
        _hip._check(self.lib.qs_mailbox_wait(self.local, self.world, self.n, self.step, self.status.data_ptr(), self.MAX_SPINS,
                                             ctypes.byref(out), stream), "qs_mailbox_wait")
        return self.views[self.step & 1]

    def check(self):
        self.watch.check()

    def close(self):
        for p in self.opened:
            self.lib.qs_mailbox_close(p)
        self.opened = []
        if self.local:
            self.lib.qs_mailbox_free(self.local)
            self.local = None

    def __del__(self):           # a mailbox dropped without close() (its site was garbage-collected): unmap and free, never raise
        try:
            self.close()
        except Exception:        # noqa: BLE001
            pass


def mailbox_enabled() -> bool:
    return get_option("sync_statistics") == "mailbox" and dist.is_available() and dist.is_initialized()


_mailboxes = None     # owner (the site's QuantizeLayer) -> Mailbox; weak keys: a mailbox goes with its site, never into a copy or pickle


def _mailbox_of(owner, n: int, device) -> Mailbox:
    global _mailboxes
    if _mailboxes is None:
        import weakref
        _mailboxes = weakref.WeakKeyDictionary()
    box = _mailboxes.get(owner)
    if box is None or box.n != n or box.status.device != device:
        if box is not None:
            box.close()
        box = _mailboxes[owner] = Mailbox(n, device)
    return box


def mailbox_exchange(owner, rec: torch.Tensor) -> torch.Tensor:
    """`owner`: the module the mailbox belongs to; it is created at the site's first exchange -- on every rank at the same one"""
    return _mailbox_of(owner, rec.numel(), rec.device).exchange(rec)


def mailbox_max_(owner, acc: torch.Tensor) -> torch.Tensor:
    """the all-reduce (MAX) of a quantize-only site's abs-max accumulator lines through the site's mailbox: every rank publishes
    its lines, waits for everybody's, and takes the maximum as integer keys in place (qs_records_max) -- no host collective"""
    from qsparse_amd import _hip
    flat = acc.view(-1)
    rec = flat.view(torch.float32) if flat.dtype != torch.float32 else flat
    gathered = _mailbox_of(owner, rec.numel(), rec.device).exchange(rec)
    _hip._check(_hip.load().qs_records_max(gathered.data_ptr(), dist.get_world_size(), rec.numel(), rec.data_ptr(), _hip._stream(rec)),
                "qs_records_max")
    return acc


def check_mailboxes():
    """read the status word of every device that has mailboxes (one 4-byte device read each, with a sync): raises MailboxTimeout
    if any wait since the last check ran out of spins.  Cheap enough for the end of every step -- behind `optimizer.step()`, which
    syncs nothing but is where a training loop can afford it; the exchanges themselves poll without a sync."""
    for w in list(_watches.values()):
        w.check()


def close_mailboxes():
    """check, then unmap and free every mailbox (before destroy_process_group; every rank)"""
    global _mailboxes
    try:
        check_mailboxes()
    finally:
        for box in list((_mailboxes or {}).values()):
            box.close()
        _mailboxes = None
        _watches.clear()


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
