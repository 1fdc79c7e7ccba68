"""Magnitude pruning operators -- API of the reference's qsparse/sparse.py.

Same public names, arguments and state (``mask``, ``_n_updates``, ``_cur_sparsity``, ``callback.t``,
``callback.magnitude``) as mlzxy/qsparse v2.0.1.  For GPU tensors the tensor math is HIP: staged
mean-|x| (``qs_mean_dim``), running mean (``qs_running_mean``), k-th value + compare
(``qs_kth_value`` / ``qs_mask_ge``) and mask apply forward/backward (``qs_mask_apply``).  The step
counters are mirrored on the host (``HostMirror``) so a forward issues no ``.item()`` sync.
"""
import copy
from typing import Callable, Iterable

import numpy as np
import torch
import torch.nn as nn

from qsparse_amd import _hip
from qsparse_amd import distributed as qdist
from qsparse_amd.common import HostMirror, adopt_state_parameters, state_parameter
from qsparse_amd.imitation import imitate
from qsparse_amd.util import (_reduction_plan, _staged_mean_hip, calculate_mask_given_importance, get_option, logging,
                              squeeze_tensor_to_shape, threshold_rank)


class _MaskApply(torch.autograd.Function):
    """``x * mask`` for a GPU tensor; backward ``g * mask`` (autograd's MulBackward0 in the reference,
    qsparse/sparse.py:66,116,122,263).  ``relu_dim >= 0``: ``max(x, 0) * mask`` for a mask that varies along that
    dim only, with the ReLU's gate in the backward (a folded nn.ReLU; relu(x) is never materialised)."""

    @staticmethod
    def forward(ctx, x, mask, relu_dim=-1, act=1):
        """``act``: what ``pre_relu`` stands for -- 1 (nn.ReLU) or the handle of another folded activation"""
        ctx.relu_dim, ctx.act = relu_dim, act
        ctx.gate_meta = None
        if relu_dim >= 0:
            if ctx.needs_input_grad[0] and get_option("relu_gate"):
                # the backward needs one bit of x per element (x <= 0): recorded by this pass, x itself is not kept; the
                # bitmap is a saved tensor like any other (released with the graph, visible to saved-tensor hooks)
                y, gate = _hip.mask_apply(x, mask, pre_relu=act, want_gate=True)
                ctx.gate_meta = (gate.shape, gate.dtype, gate.channels_last)
                ctx.save_for_backward(mask, gate.bits)
                return y
            ctx.save_for_backward(mask, x)
            return _hip.mask_apply(x, mask, pre_relu=act)
        ctx.save_for_backward(mask)
        return _hip.mask_apply(x, mask)

    @staticmethod
    def backward(ctx, grad):
        if ctx.relu_dim >= 0:
            inf = float("inf")   # no clamp: qs_quant_ste_relu_bwd reduces to gate(x) * g * mask
            if ctx.gate_meta is not None:
                mask, bits = ctx.saved_tensors
                return _hip.ste_relu_bwd(grad, None, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=ctx.relu_dim,
                                         gate=_hip.ReluGate.from_saved(bits, *ctx.gate_meta), act=ctx.act), None, None, None
            mask, x = ctx.saved_tensors
            return _hip.ste_relu_bwd(grad, x, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=ctx.relu_dim,
                                     act=ctx.act), None, None, None
        (mask,) = ctx.saved_tensors
        return _hip.mask_apply(grad, mask), None, None, None


# ----------------------------------------------------------------------------------------------------------------------
# One FFI call per training step of a PruneLayer alone (`qs_site_fwd` / `qs_site_bwd` with QS_SITE_NO_QUANT): the staged mean,
# the select (running magnitude, mask, counters) and the mask apply of `MagnitudePruningCallback._forward_single_launch` issued
# from one ctypes call with a cached plan instead of three or four -- the same launches, the same arithmetic.
# ----------------------------------------------------------------------------------------------------------------------
class _PrunePlan:
    __slots__ = ("key", "c", "ref", "keep", "channels_last", "cd")

    def __init__(self):
        self.key = None

    def __deepcopy__(self, memo):        # raw pointers never travel
        return _PrunePlan()

    def __reduce__(self):
        return (_PrunePlan, ())


def _prune_plan(cb, x: torch.Tensor, mask: torch.Tensor, step_counter, act: int):
    """the cached `qs_site_plan` of a prune-only site for inputs like `x`, or None when the composite does not cover it (then the
    fine-grained launches run): 4-d NCHW / channels_last, 2-d, or contiguous 3-d token-major [B, T, C] activation (mask on the last
    dim: qs_site_plan layout 3) with a batch of at least two, a channel mask, state on x's device"""
    if x.dim() not in (2, 3, 4) or x.shape[0] < 2 or step_counter is None or not hasattr(cb, "magnitude"):
        return None
    flat, token = x.dim() == 2, x.dim() == 3
    cd = 2 if token else 1
    if token:
        N, C, H, W = x.shape[0], x.shape[2], x.shape[1], 1
    else:
        N, C, H, W = (x.shape[0], x.shape[1], 1, 1) if flat else x.shape
    if mask.numel() != C or _channel_dim(mask) != cd or (not flat and not token and (H < 2 or W < 2)):
        return None
    if token and (H < 2 or not x.is_contiguous()):
        return None
    cl = not x.is_contiguous()
    if cl and (flat or not x.is_contiguous(memory_format=torch.channels_last) or C % 8):
        return None
    if x.data_ptr() % 16 or x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return None
    if not token and (H * W + W) * 4 > _hip.LAST2_MAX_TILE_BYTES:
        return None
    state = (cb.magnitude, mask, step_counter, cb.t)
    if any((not t.is_cuda) or t.device != x.device for t in state) or cb.magnitude.numel() != C:
        return None
    graph_safe = bool(get_option("graph_safe"))
    key = (N, C, H, W, flat, token, x.dtype, cl, x.device, graph_safe, act) + tuple(t.data_ptr() for t in state)
    plan = cb.__dict__.get("_qs_prune_plan")
    if plan is not None and plan.key == key:
        return plan
    plan = _PrunePlan()
    plan.key, plan.channels_last, plan.cd = key, cl, cd
    stage = None if flat else torch.empty(C * H * W, dtype=x.dtype, device=x.device)
    stage_mean = torch.empty(C, dtype=x.dtype, device=x.device)
    plan.keep = (stage, stage_mean) + state
    c = _hip.SitePlanStruct()
    c.N, c.C, c.H, c.W = N, C, H, W
    c.layout, c.xdt, c.ydt, c.bits = (3 if token else 2 if flat else int(cl)), _hip.dt(x), _hip.dt(x), 8
    c.magnitude, c.mask = cb.magnitude.data_ptr(), mask.data_ptr()
    c.stage, c.stage_mean = (None if flat else stage.data_ptr()), stage_mean.data_ptr()
    c.absmax_stride = 1
    c.prune_n_updates, c.callback_t = step_counter.data_ptr(), cb.t.data_ptr()
    c.callback_t_from_device = int(graph_safe)
    c.act = int(act) or 1
    plan.c, plan.ref = c, __import__("ctypes").byref(c)
    cb.__dict__["_qs_prune_plan"] = plan
    return plan


class _PruneSiteStep(torch.autograd.Function):
    """a PruneLayer's training step through qs_site_fwd / qs_site_bwd (QS_SITE_NO_QUANT); same results as `_importance` +
    `qs_pq_select` + `_MaskApply`"""

    @staticmethod
    def forward(ctx, x, plan, flags, t_mag, k, mask):
        want_gate = bool((flags & _hip.SITE_PRE_RELU) and ctx.needs_input_grad[0])
        y = torch.empty_like(x)
        bits = torch.empty((x.numel() + 7) // 8, dtype=torch.uint8, device=x.device) if want_gate else None
        if want_gate:
            _hip.note_gate(bits)
        _hip.site_fwd(plan.ref, x, y, bits, flags, t_mag, k, 0)
        ctx.plan, ctx.has_gate = plan, want_gate
        ctx.x_meta = (x.shape, x.dtype, x.stride())
        ctx.save_for_backward(mask, bits if want_gate else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        mask, bits = ctx.saved_tensors
        plan = ctx.plan
        shape, dtype, strides = ctx.x_meta
        if (g.shape == shape and g.dtype == dtype and g.stride() == strides and g.data_ptr() % 16 == 0 and not _hip.logging_events()):
            gx = torch.empty_like(g)
            flags = _hip.SITE_NO_QUANT | (_hip.SITE_ELIDE if _hip.elide_mode == "all" else 0)
            _hip.site_bwd(plan.ref, g, bits if ctx.has_gate else None, gx, flags, 0.0, 0.0)
            return gx, None, None, None, None, None
        if ctx.has_gate:
            inf = float("inf")
            gate = _hip.ReluGate.from_saved(bits, shape, dtype, plan.channels_last)
            return (_hip.ste_relu_bwd(g, None, 1.0, False, -inf, inf, mask.reshape(-1), mask_channel_index=plan.cd, gate=gate, act=plan.c.act),
                    None, None, None, None, None)
        return _hip.mask_apply(g, mask), None, None, None, None, None


def _channel_dim(mask: torch.Tensor) -> int:
    """the single dim along which ``mask`` varies, or -1"""
    dims = [d for d, s in enumerate(mask.shape) if s != 1]
    return dims[0] if len(dims) == 1 else -1


def apply_mask(x: torch.Tensor, mask: torch.Tensor, pre_relu=False) -> torch.Tensor:
    """``pre_relu``: False, True (a folded nn.ReLU) or the handle of another folded activation (``_hip.activation``)"""
    if _hip.on_hip(x):
        if pre_relu:
            d = _channel_dim(mask)
            if d < 0 or mask.shape[d] != x.shape[d]:
                return _MaskApply.apply(_hip.act_torch(pre_relu, x), mask.detach())
            return _MaskApply.apply(x, mask.detach(), d, int(pre_relu))
        return _MaskApply.apply(x, mask.detach())
    return (torch.relu(x) if pre_relu else x) * mask


def _importance(x: torch.Tensor, shape, l0: bool = False, pre_relu: bool = False) -> torch.Tensor:
    """``squeeze_tensor_to_shape(x.abs(), shape)`` -- with the optional L0 substitution -- in one pass
    over ``x`` on the GPU (reference sparse.py:85-87).  ``pre_relu``: the importance of max(x, 0)."""
    if pre_relu:
        dims = _reduction_plan(x.shape, shape)
        if _hip.on_hip(x) and dims and not l0:
            return _staged_mean_hip(x, dims, take_abs=True, pre_relu=pre_relu)
        x = _hip.act_torch(pre_relu, x) if x.is_cuda else torch.relu(x)
    if _hip.on_hip(x):
        dims = _reduction_plan(x.shape, shape)
        flag = _hip.l0_flag(x) if l0 else None
        if l0 and not dims:   # nothing to reduce: materialise (x != 0).float() or |x| via the flag
            return torch.where(flag.bool(), (x != 0).float(), x.abs().float())
        if l0 and x.dtype != torch.float32:
            # the L0 substitute is a float32 tensor, |x| stays in x's dtype, and the staged means round to whichever
            # it is: for 2-byte inputs the choice changes the arithmetic, so it is taken on the host like the
            # reference's own `x.min().item()` (sparse.py:85)
            if not bool(flag.item()):
                flag = None
        return _staged_mean_hip(x, dims, take_abs=True, l0_flag=flag)
    if l0 and x.min().item() == 0:
        x = (x != 0).float()
    return squeeze_tensor_to_shape(x.abs(), shape)


class ArgumentError(TypeError, ValueError):
    """what `MagnitudePruningCallback(use_gradient=True, running_average=False)` raises.  The reference means to raise
    `argparse.ArgumentError(message)` there and, that class needing two arguments, raises `TypeError` instead (sparse.py:45):
    code written against either intention -- `except TypeError` as the reference behaves, `except ValueError` as it reads --
    keeps working."""


class MagnitudePruningCallback(nn.Module):
    def __init__(self, mask_refresh_interval: int = -1, stop_mask_refresh: int = float("inf"),
                 use_gradient: bool = False, running_average: bool = True, l0: bool = False,
                 forward_hook: Callable[[torch.Tensor, str], None] = None):
        """magnitude-based mask construction, the callback of ``prune`` (reference sparse.py:18-56).

        Args:
            mask_refresh_interval: steps between mask rebuilds (<=0 means every step).
            stop_mask_refresh: step after which the mask is frozen.
            use_gradient: rank by gradient magnitude (collected through a tensor hook) instead.
            running_average: rank by the running mean of the magnitude rather than the current input.
            l0: count non-zeros instead of averaging magnitudes when the input has exact zeros.
            forward_hook: called as ``hook(mask, name)`` after every training forward.
        """
        super().__init__()
        if use_gradient and not running_average:
            raise ArgumentError("the combination of `use_gradient=True` and `running_average=False` is not supported")
        self.mask_refresh_interval = mask_refresh_interval
        self.stop_mask_refresh = stop_mask_refresh
        self.use_gradient = use_gradient
        self.running_average = running_average
        self.l0 = l0
        self.forward_hook = forward_hook
        self.prev_grad_hook = None
        self.t = state_parameter(torch.full((1,), -1))
        self._t_host = HostMirror()

    def __setstate__(self, state):
        super().__setstate__(state)
        adopt_state_parameters(self)     # unpickling rebuilds plain Parameters

    @property
    def initted(self) -> bool:
        return self._t_host.read(self.t) != -1

    def initialize(self, mask: torch.Tensor):
        if self.running_average:
            self.magnitude = nn.Parameter(torch.zeros_like(mask, dtype=torch.float), requires_grad=False)     # (the mask's layout)

    def update_magnitude(self, x):
        """magnitude <- (t*magnitude + mean|x|) / (t+1)   (reference sparse.py:82-89)"""
        if not self.running_average:
            return
        with torch.no_grad():
            t = self._t_host.read(self.t)
            imp = qdist.allreduce_mean(_importance(x.detach(), self.magnitude.shape, self.l0))
            if _hip.on_hip(x):
                t_dev = self.t.data if (get_option("graph_safe") and self.t.is_cuda) else None
                _hip.running_mean(self.magnitude.data, imp, t, t_dev=t_dev)
            else:
                _hip.refuse_capture(x, "the running magnitude")
                self.magnitude.data[:] = _hip.true_div(t * self.magnitude + imp, t + 1)

    def receive_input(self, x: torch.Tensor):
        if not self.use_gradient:
            self.update_magnitude(x)
            return
        if self.prev_grad_hook is not None:
            self.prev_grad_hook.remove()
            self.prev_grad_hook = None
        if x.requires_grad:
            self.prev_grad_hook = x.register_hook(lambda grad: self.update_magnitude(grad))
        else:
            logging.error("meeting no-grad tensor")

    def prune_and_update_mask(self, x: torch.Tensor, sparsity: float, mask: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            importance = self.magnitude if self.running_average else qdist.allreduce_mean(
                _importance(x.detach(), mask.shape))
            if _hip.on_hip(x) and importance.dtype in _hip.HIP_DTYPES:
                imp = importance.detach().to(torch.float32).contiguous()
                n = imp.numel()
                k = threshold_rank(sparsity, n)
                if k >= n:
                    raise IndexError(f"index {k} is out of bounds for dimension 0 with size {n}")
                _hip.mask_ge(imp, _hip.kth_value(imp, k), mask.data)
            else:
                mask.data[:] = calculate_mask_given_importance(importance, sparsity)
        return apply_mask(x, mask)

    def refresh_due(self, t: int, sparsity: float) -> bool:
        """whether step ``t`` rebuilds the mask (reference sparse.py:110-113)."""
        return (sparsity >= 0 and t % self.mask_refresh_interval == 0 and t <= self.stop_mask_refresh
                and (t > 0 or not self.running_average))

    def begin_step(self, mask: torch.Tensor) -> int:
        """first-use initialisation; returns the step index ``t`` of this call."""
        if not self.initted:
            self.initialize(mask)
            self._t_host.write(self.t, 0)
            if self.mask_refresh_interval <= 0:
                self.mask_refresh_interval = 1
        return self._t_host.read(self.t)

    def end_step(self, mask: torch.Tensor, name: str):
        self._t_host.add(self.t, 1)
        if self.forward_hook is not None:
            self.forward_hook(mask, name)

    def _single_launch_select(self, x: torch.Tensor, mask: torch.Tensor) -> bool:
        """GPU tensors, running-average magnitudes of at most 65536 mask entries, no overridden policy methods:
        running mean + k-th value + mask (+ step counters) are then ONE launch (qs_pq_select)."""
        cls = type(self)
        return (_hip.on_hip(x) and self.running_average and not self.use_gradient and 2 <= mask.numel() <= 65536
                and mask.is_cuda and mask.is_contiguous() and hasattr(self, "magnitude")
                and cls.update_magnitude is MagnitudePruningCallback.update_magnitude
                and cls.receive_input is MagnitudePruningCallback.receive_input
                and cls.prune_and_update_mask is MagnitudePruningCallback.prune_and_update_mask)

    def _forward_single_launch(self, x, sparsity, mask, name, t, step_counter, pre_relu=False):
        update = t < self.stop_mask_refresh
        refresh = self.refresh_due(t, sparsity)
        self.__dict__["_bumped_step_counter"] = False
        t_on_dev = self.t.is_cuda and self.t.device == x.device
        if ((update or refresh) and t_on_dev and not self.l0 and not _hip.logging_events()
                and (not pre_relu or get_option("relu_gate") or not (torch.is_grad_enabled() and x.requires_grad))
                and not qdist.exchange_active(qdist.stats_world_size())):
            # one call into the library for the whole step (statistics, select, mask apply): qs_site_fwd(QS_SITE_NO_QUANT)
            plan = _prune_plan(self, x, mask, step_counter, int(pre_relu) or 1)
            if plan is not None:
                k = 0
                if refresh:
                    k = threshold_rank(sparsity, mask.numel())
                    if k >= mask.numel():
                        raise IndexError(f"index {k} is out of bounds for dimension 0 with size {mask.numel()}")
                flags = (_hip.SITE_NO_QUANT | (_hip.SITE_LIVE if update else 0) | (_hip.SITE_REFRESH if refresh else 0)
                         | (_hip.SITE_PRE_RELU if pre_relu else 0) | (_hip.SITE_ELIDE if _hip._elide_all() else 0))
                # (x * mask keeps the sign of x on a pruned channel: only elide_pruned="all" may skip those loads, as `mask_apply`)
                out = _PruneSiteStep.apply(x, plan, flags, t, k, mask)
                self.__dict__["_bumped_step_counter"] = True
                self._t_host.note_device_add(self.t, 1)
                if self.forward_hook is not None:
                    self.forward_hook(mask, name)
                return out
        if update or refresh:
            with torch.no_grad():
                imp = k = None
                if update:
                    imp = qdist.allreduce_mean(_importance(x.detach(), self.magnitude.shape, self.l0, pre_relu)).contiguous().view(-1)
                if refresh:
                    n = mask.numel()
                    k = threshold_rank(sparsity, n)
                    if k >= n:
                        raise IndexError(f"index {k} is out of bounds for dimension 0 with size {n}")
                bump = None
                if step_counter is not None and step_counter.is_cuda and step_counter.device == x.device:
                    bump = step_counter.data
                _hip.pq_select(self.magnitude.data.view(-1), imp, update, t, refresh, k or 0, mask.data.view(-1), None, False,
                               0, 8, None, bump_a=bump, bump_c=self.t.data if t_on_dev else None,
                               t_mag_dev=self.t.data if (update and t_on_dev and get_option("graph_safe")) else None)
                self.__dict__["_bumped_step_counter"] = bump is not None
            out = apply_mask(x, mask, pre_relu)
            if t_on_dev:
                self._t_host.note_device_add(self.t, 1)
                if self.forward_hook is not None:
                    self.forward_hook(mask, name)
                return out
        else:
            out = apply_mask(x, mask, pre_relu)
        self.end_step(mask, name)
        return out

    def forward(self, x: torch.Tensor, sparsity: float, mask: torch.Tensor, name="", step_counter=None,
                pre_relu: bool = False):
        """``pre_relu`` (used by the fused ReLU->prune site only): treat ``x`` as the input of a ReLU that has not
        been applied yet -- statistics, mask apply and backward then work on max(x, 0)."""
        if not self.training:
            return apply_mask(x, mask, pre_relu)
        t = self.begin_step(mask)
        if self._single_launch_select(x, mask):
            return self._forward_single_launch(x, sparsity, mask, name, t, step_counter, pre_relu)
        if pre_relu:
            x = _hip.act_torch(pre_relu, x) if x.is_cuda else torch.relu(x)
        if t < self.stop_mask_refresh:
            self.receive_input(x)
        if self.refresh_due(t, sparsity):
            out = self.prune_and_update_mask(x, sparsity, mask)
        else:
            out = apply_mask(x, mask)
        self.end_step(mask, name)
        return out


class UniformPruningCallback(MagnitudePruningCallback):
    """unstructured random pruning that ignores magnitudes; positions already pruned stay pruned
    (reference sparse.py:125-152).  Uses numpy's global RNG on the host, like the reference."""

    def initialize(self, mask: torch.Tensor):
        pass

    def receive_input(self, x: torch.Tensor):
        pass

    def prune_and_update_mask(self, x: torch.Tensor, sparsity: float, mask: torch.Tensor) -> torch.Tensor:
        cur_sparsity = (~mask).sum().item() / mask.numel()
        if cur_sparsity > sparsity:
            logging.warning("sparsity is decreasing, which shall not happen")
        budget = int(round((sparsity - cur_sparsity) * np.prod(mask.shape)))
        alive = mask.nonzero(as_tuple=True)
        chosen = torch.from_numpy(np.random.choice(len(alive[0]), size=budget, replace=False)).to(mask.device)
        mask.data[tuple(idx[chosen] for idx in alive)] = False
        return apply_mask(x, mask)


class PruneLayer(nn.Module):
    """stateful pruning operator with the cubic sparsity schedule (reference sparse.py:157-273)."""

    def __str__(self):
        return (f"PruneLayer(sparsity={self.sparsity}, start={self.start}, interval={self.interval}, "
                f"repetition={self.repetition}, dimensions={self.dimensions})")

    __repr__ = __str__

    def __init__(self, sparsity: float = 0.5, dimensions: Iterable[int] = {1},
                 callback: MagnitudePruningCallback = MagnitudePruningCallback(), start: int = 1000,
                 interval: int = 1000, repetition: int = 4, rampup: bool = False, name=""):
        super().__init__()
        if get_option("log_on_created"):
            logging.warning(f"[Prune{name if name == '' else f' @ {name}'}] start = {start} interval = {interval} "
                            f"repetition = {repetition} sparsity = {sparsity} dimensions = {dimensions}")
        first = 1 if rampup else 0
        self.schedules = [start + interval * (first + i) for i in range(repetition)]
        self.start = start
        self.interval = interval
        self.repetition = repetition
        self.sparsity = sparsity
        self.name = name
        self.callback = callback
        self.rampup_interval = 0 if rampup else interval
        self.dimensions = set(dimensions)
        self.register_parameter("mask", nn.Parameter(torch.tensor(-1, dtype=torch.int), requires_grad=False))
        for key in ("_n_updates", "_cur_sparsity"):   # shape-less placeholders until the first forward
            self.register_parameter(key, state_parameter(torch.tensor(-1, dtype=torch.int)))
        self._steps = HostMirror()
        self._sparsity_host = HostMirror()

    def __setstate__(self, state):
        super().__setstate__(state)
        adopt_state_parameters(self)     # unpickling rebuilds plain Parameters

    @property
    def initted(self) -> bool:
        return self._steps.read(self._n_updates) != -1

    def _lazy_init(self, x: torch.Tensor):
        assert len(x.shape) > 1
        mask_shape = [s if i in self.dimensions else 1 for i, s in enumerate(x.shape)]
        mask = torch.ones(*mask_shape, dtype=torch.bool, device=x.device)
        if (x.dim() == 4 and tuple(mask_shape) == tuple(x.shape) and not x.is_contiguous()
                and x.is_contiguous(memory_format=torch.channels_last)):
            # a full-shape mask of a channels_last tensor (the weight of a network moved with `.to(memory_format=...)`) is laid out
            # like that tensor: same values, same shape -- and x * mask, the running magnitude and the multi-tensor weight
            # kernels then walk both in one memory order
            mask = mask.contiguous(memory_format=torch.channels_last)
        self.mask = nn.Parameter(mask, requires_grad=False)
        if self.mask.numel() == 1:
            logging.warn(f"the mask shape of {self.name} is {tuple(self.mask.shape)}, which is not prunable")
        self._n_updates = state_parameter(torch.zeros(1, dtype=torch.int, device=x.device))
        self._cur_sparsity = state_parameter(torch.zeros(1, device=x.device))

    def scheduled_sparsity(self, n: int) -> float:
        """cubic ramp evaluated in Python doubles, stored as fp32 (reference sparse.py:252-257)."""
        ratio = (1.0 - (n - self.start + self.rampup_interval) / (self.interval * self.repetition)) ** 3
        return self.sparsity * (1 - ratio)

    def advance_schedule(self) -> int:
        """apply the sparsity schedule for the current step; returns the step index."""
        n = self._steps.read(self._n_updates)
        if self.training and n in self.schedules:
            self._sparsity_host.write(self._cur_sparsity, self.scheduled_sparsity(n))
            if get_option("log_during_train"):
                tag = self.name if self.name == "" else f" @ {self.name}"
                logging.warning(f"[Prune{tag}] [Step {n}] pruned {self._sparsity_host.read(self._cur_sparsity):.02f}")
        return n

    def current_sparsity(self) -> float:
        return self._sparsity_host.read(self._cur_sparsity)

    def is_active(self) -> bool:
        """whether the next forward multiplies by the mask (used by the fused ReLU->prune site)."""
        if not self.initted:
            return False
        return (not self.training) or self.mask.numel() == 1 or self._steps.read(self._n_updates) >= self.start

    @_hip.keeps_layout
    def forward(self, x: torch.Tensor, pre_relu: bool = False) -> torch.Tensor:
        """prune ``x`` according to the schedule; raises ``RuntimeError`` when a full-shape mask meets a
        different input shape in evaluation mode.  ``pre_relu``: see MagnitudePruningCallback.forward."""
        if not self.initted:
            self._lazy_init(x)
        n = self.advance_schedule()
        if not self.training or self.mask.numel() == 1:
            return apply_mask(x, self.mask, pre_relu)
        if pre_relu and not (n >= self.start and _hip.on_hip(x) and type(self.callback) is MagnitudePruningCallback):
            x, pre_relu = (_hip.act_torch(pre_relu, x) if x.is_cuda else torch.relu(x)), False
        if n >= self.start:
            if n == self.start and get_option("log_during_train"):
                logging.warning(f"Start pruning at {self.name} @ {n}")
            cb = self.callback
            if _hip.on_hip(x) and type(cb) is MagnitudePruningCallback:
                # the layer's step counter rides along in the callback's select launch when there is one
                cb.__dict__["_bumped_step_counter"] = False
                out = cb(x, self.current_sparsity(), mask=self.mask, name=self.name, step_counter=self._n_updates,
                         pre_relu=pre_relu)
                if cb.__dict__.get("_bumped_step_counter", False):
                    cb.__dict__["_bumped_step_counter"] = False
                    self._steps.note_device_add(self._n_updates, 1)
                    return out
            else:
                out = cb(x, self.current_sparsity(), mask=self.mask, name=self.name)
        else:
            out = x
        self._steps.add(self._n_updates, 1)
        return out


def prune(inp: nn.Module = None, sparsity: float = 0.5, dimensions: Iterable[int] = {1},
          callback: MagnitudePruningCallback = None, start: int = 1000, interval: int = 1000, repetition: int = 4,
          rampup: bool = False, name="") -> nn.Module:
    """build a ``PruneLayer`` (no ``inp``) for activations, or wrap ``inp`` so that its weight is read
    through one (reference sparse.py:276-339).

    Args:
        inp: module whose weight is to be pruned, or None for an activation operator.
        sparsity: target sparsity reached at the end of the schedule.
        dimensions: dims along which the mask varies ({1}: channel pruning of NCHW activations).
        callback: mask construction policy, default ``MagnitudePruningCallback()``.
        start / interval / repetition / rampup: the cubic sparsity schedule.
        name: used in log lines.
    """
    callback = callback or MagnitudePruningCallback()
    kwargs = dict(start=int(start), sparsity=sparsity, interval=int(interval), repetition=repetition, rampup=rampup,
                  name=name, callback=callback, dimensions=dimensions)
    if inp is None:
        layer = PruneLayer(**kwargs)
        layer._kwargs = kwargs
        return layer
    if isinstance(inp, nn.Module):
        return imitate(inp, "prune", PruneLayer(**kwargs))
    raise ValueError(f"{inp} is not a valid argument for prune")


def devise_layerwise_pruning_schedule(net: nn.Module, start: int = 1, interval: int = 10,
                                      mask_refresh_interval: int = 1, inplace=False):
    """stagger the prune layers of ``net`` one after another (reference sparse.py:343-359).  Like the
    reference it leaves ``rampup_interval`` untouched."""
    if not inplace:
        net = copy.deepcopy(net)
    layers = [m for m in net.modules() if isinstance(m, PruneLayer)]
    weight_only = all(layer.name.endswith(".prune") for layer in layers)
    for layer in layers:
        layer.start, layer.interval, layer.repetition = start, interval, 1
        layer.schedules = [start]
        layer.callback.mask_refresh_interval = mask_refresh_interval
        layer.callback.stop_mask_refresh = interval
        if weight_only:
            layer.callback.running_average = False
        start += interval + 1
    logging.danger(f"Pruning stops at iteration - {start}")
    return net
