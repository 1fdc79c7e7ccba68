"""Multi-tensor weight path.

``quantize(conv)`` reads ``conv.weight`` -- and, with ``bias_bits``, ``conv.bias`` -- through a ``QuantizeLayer`` (reference
quantize.py:559-571 via imitation.py:61-68): per tensor and training step an abs-max, a running-scale update and a
quantization -- three launches of 3-5 us each plus ~100 us of Python.  A converted ResNet-50 has 54 such layers.  They depend on
nothing but the parameters, so ``WeightBatcher`` evaluates all of them at the start of the root's forward pass with THREE
launches in total (``qs_multi_absmax``, ``qs_multi_scale_update``, ``qs_multi_quant_fwd`` over a device-resident table of
tensor descriptors built once per set of layers; same arithmetic, bit-identical results) and hands every layer its quantized
tensor when its forward asks for it.  The STE backward runs per group of eight tensors (gradients become ready layer by layer).

Tensor-wise AND per-channel Scaler / Decimal quantizers take part -- per-channel along dim 1 is the reference's default
(``quantize(bits=8)``: ``channelwise=1``, quantize.py:524) -- and so do bias quantizers, which share their layer's callback
object and therefore its running-mean count ``t`` (quantize.py:548,559-571: the weight's update sees ``t``, the bias's
``t + 1``).

``convert`` installs it on the network it returns (``set_qsparse_options(batch_weights=False)`` or
``convert(..., batch_weights=False)`` opt out; ``WeightBatcher.install(model)`` does the same by hand, ``.remove()``
undoes it).  It is safe by construction for anything a forward pass may do:

  * The reference evaluates an operator when -- and only when -- the layer's parameter is read (imitation.py:61-68).  A
    tensor whose precomputed value was NOT consumed by the end of the root's forward -- a branch the forward skipped, an
    exception half-way -- is rolled back to exactly the state it had before: running scale (from the backup
    ``qs_multi_scale_update`` wrote), ``_n_updates``, the callback's ``t`` (host and device copy) and ``_quantized``.  The
    forward hook that does this also runs when the forward raised.  Such a parameter receives NO gradient from its place in
    the hand-out node (like the layer the reference never evaluated: ``.grad`` stays ``None``, an optimizer with weight decay
    does not touch it) -- except under an initialised process group, where ``DistributedDataParallel`` counts the parameter as
    used and waits for a gradient: there it contributes zeros, as DDP's own unused parameters do.
  * A parameter that is read a second time in the same forward takes the inline path, as the second read of the reference.
  * A parameter written between the precomputation and its read is rolled back and re-evaluated inline.  The write is seen
    through ``Tensor._version`` (every in-place operation); the one route that bypasses the counter is ``param.data``.  The
    place where user code runs between the two is a forward pre-hook of the layer itself: layers that carry one are not
    batched.  (A raw ``.data`` write to a layer's weight from OTHER code running inside the same forward, before that
    layer's read, is not visible -- switch ``batch_weights`` off for such a network.)
  * A bias that is read BEFORE its layer's weight (the reference would then have updated the bias with ``t``, not ``t + 1``)
    rolls both back; they are evaluated inline in the order of the reads.
  * A PRUNED weight -- ``quantize(prune(conv))``: the quantizer's input is ``weight * mask`` (sparse.py:263) -- takes part too,
    with its whole prune operator: the kernels multiply by the mask themselves, forward and backward, advance the operator's
    counters, compute the importance -- ``|weight|`` element by element for a full-shape mask, the staged mean
    ``squeeze_tensor_to_shape(|weight|, mask.shape)`` for a mask over a subset of dims (``prune()``'s default: one entry per input
    channel; ``qs_multi_stage_mean``, one launch per stage level for all layers) --, average the running magnitude
    (``qs_multi_magnitude``) and rebuild the mask (``qs_multi_mask_refresh``: the callback's radix select + mask for all layers at
    once) whenever the callback would -- with the stock ``MagnitudePruningCallback()`` that is every read.  Only the few reads
    that change the sparsity stay per layer (and, per layer, a channels_last weight whose first reduced dim is not dim 0: its
    staged mean has an order of its own).  Everything is rolled back like the quantizer's state (magnitudes and masks from the
    backups the launches wrote).  A pruned weight WITHOUT a quantizer (``convert(model, prune(...), weight_layers=[...])`` alone)
    and one whose quantizer is still in its identity phase (it only counts the read, quantize.py:496-517) take part the same
    way: the table row hands out ``weight * mask``.
  * Layers whose operators carry hooks, layers on the CPU, group-wise quantizers, callbacks shared between layers, pruning
    callbacks other than ``MagnitudePruningCallback`` (or ranking by gradient / L0) never take part; they keep their inline path.

In evaluation mode under ``torch.no_grad()`` the quantized tensors are computed once and handed out again until a parameter
or a scale changes (serving: no weight-side launch at all per request).
"""
import os
from typing import List, Optional

import torch
import torch.nn as nn

from torch.nn.modules import module as _m

from qsparse_amd import _hip
from qsparse_amd import distributed as qdist
from qsparse_amd.quantize import DecimalQuantizer, QuantizeLayer, ScalerQuantizer
from qsparse_amd.sparse import MagnitudePruningCallback, PruneLayer
from qsparse_amd.util import _options_epoch, _reduction_plan, get_option, logging, threshold_rank

_ALIGN = 64   # elements between the starts of two outputs in the flat buffer (256 bytes)
_READY = {"weight": "_qs_ready_weight", "bias": "_qs_ready_bias"}   # layer.__dict__ keys of precomputed tensors waiting for their read
_ATTR = "_qs_weight_batcher"       # root.__dict__ key of the installed batcher
_QUANT = {"weight": "quantize", "bias": "quantize_bias"}


_GROUP = 8      # tensors per hand-out node: their gradients are clamped together, by ONE launch, when the group's
                # earliest layer has finished its backward (small groups keep DDP's bucketed all-reduce overlapping)


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _geometry(t: torch.Tensor, channel_index: int):
    """(outer, C, inner) of the CONTIGUOUS view of `t`'s memory around its channel dim, or None when there is none (the caller
    then keeps the per-layer entry points, which copy).  Tensor-wise: (1, 1, numel) for any dense layout."""
    if channel_index < 0:
        return (1, 1, t.numel()) if _dense(t) else None
    if channel_index >= t.dim() or t.numel() == 0:
        return None
    shape = tuple(t.shape)
    C = shape[channel_index]
    if t.is_contiguous():
        outer = 1
        for s in shape[:channel_index]:
            outer *= s
        inner = 1
        for s in shape[channel_index + 1:]:
            inner *= s
    elif t.dim() == 4 and channel_index == 1 and t.is_contiguous(memory_format=torch.channels_last):
        outer, inner = shape[0] * shape[2] * shape[3], 1        # [N][H][W][C] in memory
    else:
        return None
    return (1, 1, t.numel()) if C == 1 else (outer, C, inner)


class _GroupSte(torch.autograd.Function):
    """hands out the precomputed quantized tensors of a GROUP of consecutive layers through one autograd node whose
    backward applies the quantizers' STE clamp (reference quantize.py:66-77, 120-131) to all of their gradients with one
    multi-tensor launch (qs_multi_ste_bwd) -- instead of one node, one Python backward and one 4 us launch per layer."""

    @staticmethod
    def forward(ctx, meta, dead, *tensors):
        k = len(meta)
        ctx.meta, ctx.dead = meta, dead                     # per tensor: (is_decimal, lo_mul, hi_mul, passthrough, channel index, mask); rolled back?
        ctx.shapes = [(w.shape, w.stride()) for w in tensors[:k]]
        ctx.n_inputs = 2 + len(tensors)
        # autocast images (fused.py "Autocast image"; batch.py `_hand_out`): tensors[3k:4k], None where a member has none -- the
        # low-precision copy of the quantized weight its convolution takes instead of casting the float32 one, whose
        # low-precision weight gradient then arrives here directly (the outputs k, k+1, ... of this node)
        imgs = tensors[3 * k:4 * k] if len(tensors) >= 4 * k else (None,) * k
        ctx.img_slot = [None] * k
        ctx.save_for_backward(*tensors[2 * k:3 * k])        # the steps (scales or decimals: one per channel)
        # a member whose tensor is never read gets NO gradient, exactly like the layer the reference never evaluated
        # (zero_grad(set_to_none=True) + weight decay would otherwise start to move a parameter the forward did not use)
        ctx.set_materialize_grads(False)
        ys = tuple(y.view_as(y) for y in tensors[k:2 * k])
        # a frozen parameter (requires_grad=False: fine-tuning a head on a frozen backbone) hands out a quantized tensor that
        # does not require grad either, as layer by layer -- its convolution then skips the weight-gradient pass altogether
        frozen = [ys[i] for i in range(k) if not ctx.needs_input_grad[2 + i]]
        extra = []
        for i, img in enumerate(imgs):
            if img is not None:
                ctx.img_slot[i] = k + len(extra)
                extra.append(img.view_as(img))
                if not ctx.needs_input_grad[2 + i]:
                    frozen.append(extra[-1])
        if frozen:
            ctx.mark_non_differentiable(*frozen)
        return ys + tuple(extra)

    @staticmethod
    def backward(ctx, *grads):
        meta, steps = ctx.meta, ctx.saved_tensors
        k = len(meta)
        out = [None] * k
        grads = _merge_image_gradients(ctx, list(grads), k, steps[0].device)
        if any(ctx.dead):
            # the one combination in which "no gradient" is not an option: DistributedDataParallel counts this parameter as
            # used (it is reachable through the group's node) and waits for its gradient -- there a rolled-back member
            # contributes zeros, what DDP itself uses for a parameter that is unused on a rank
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                for i in range(k):
                    if ctx.dead[i] and grads[i] is None and ctx.needs_input_grad[2 + i]:
                        shape, stride = ctx.shapes[i]
                        out[i] = torch.empty_strided(shape, stride, dtype=torch.float32, device=steps[i].device).zero_()
        for decimal in (False, True):
            idx = [i for i in range(k) if grads[i] is not None and ctx.needs_input_grad[2 + i] and meta[i][0] == decimal
                   and not meta[i][3]]
            # the gradient's own memory view around the channel dim (a per-channel step needs the channel of every element)
            geo = {i: (_geometry(grads[i], meta[i][4] if steps[i].numel() > 1 else -1)
                       if (grads[i].dtype == torch.float32 and grads[i].data_ptr() % 16 == 0) else None) for i in idx}
            fast = [i for i in idx if geo[i] is not None]
            for i in idx:
                if geo[i] is None:                          # odd layouts / dtypes: the per-layer entry point
                    out[i] = _hip.ste_bwd(grads[i], steps[i], decimal, meta[i][4], meta[i][1], meta[i][2], False, torch.float32)
                    if meta[i][5] is not None:              # a pruned weight: the backward of `weight * mask` (sparse.py:263)
                        out[i] = out[i] * meta[i][5][0]
            if fast:
                gs = [grads[i] for i in fast]
                flat = torch.empty(sum((g.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for g in gs), dtype=torch.float32, device=gs[0].device)
                outs, off = [], 0
                for g in gs:
                    outs.append(flat.as_strided(g.shape, g.stride(), flat.storage_offset() + off))     # the gradient's own (dense) layout
                    off += (g.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                numels = [g.numel() for g in gs]
                per_channel = any(geo[i][1] > 1 for i in fast)
                # pruned weights: a gradient laid out like the weight (the rule) is what the mask geometry describes; any other layout
                # multiplies afterwards
                mk = {i: meta[i][5] for i in fast if meta[i][5] is not None}
                in_kernel = {i for i in mk if _strides(grads[i]) == mk[i][3]}
                _hip.multi_ste_bwd(len(fast), _hip.ptr_array(gs), _hip.ptr_array(outs), _hip.ptr_array([steps[i] for i in fast]),
                                   _hip.i64_array(numels), _hip.f32_array([meta[i][1] for i in fast]),
                                   _hip.f32_array([meta[i][2] for i in fast]), decimal, gs[0].device, nbytes=8 * sum(numels),
                                   channels=_hip.i32_array([geo[i][1] for i in fast]) if per_channel else None,
                                   inners=_hip.i64_array([geo[i][2] for i in fast]) if per_channel else None,
                                   masks=_hip.ptr_array([mk[i][0] if i in in_kernel else None for i in fast]) if in_kernel else None,
                                   mask_channels=_hip.i32_array([mk[i][1] if i in in_kernel else 0 for i in fast]) if in_kernel else None,
                                   mask_inners=_hip.i64_array([mk[i][2] if i in in_kernel else 1 for i in fast]) if in_kernel else None)
                for i in mk:
                    if i not in in_kernel:
                        outs[fast.index(i)] = outs[fast.index(i)] * mk[i][0]
                for i, o in zip(fast, outs):
                    out[i] = o
        for i in range(k):
            if out[i] is None and grads[i] is not None and meta[i][3]:
                out[i] = grads[i] if meta[i][5] is None else grads[i] * meta[i][5][0]      # backward_passthrough
        return (None, None) + tuple(out) + (None,) * (ctx.n_inputs - 2 - k)


def _merge_image_gradients(ctx, grads, k, device):
    """the gradient of member i = what reached its float32 tensor + float32(what reached its image), autograd's own accumulation
    (`ToCopyBackward` + add) -- the images' low-precision gradients of a group converted by ONE multi-tensor copy into a flat
    float32 buffer laid out like the weights (what `qs_multi_ste_bwd` reads), instead of one cast launch per layer"""
    override = ctx.__dict__.pop("_qs_override", None)          # a late hook replaced a member's whole gradient (fused._late_hook)
    gy = grads[:k]
    need = []
    for i in range(k):
        j = ctx.img_slot[i]
        g16 = grads[j] if j is not None else None
        if override is not None and i in override:
            gy[i] = override[i]
            continue
        if g16 is None:
            continue
        if gy[i] is None:
            need.append((i, g16))
        else:
            gy[i] = gy[i] + g16.float()
    if need:
        offs, total = [], 0
        for i, _ in need:
            offs.append(total)
            n = 1
            for sdim in ctx.shapes[i][0]:
                n *= sdim
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.empty(total, dtype=torch.float32, device=device)
        views = [flat.as_strided(ctx.shapes[i][0], ctx.shapes[i][1], flat.storage_offset() + o) for (i, _), o in zip(need, offs)]
        ok = all(v.shape == g.shape for v, (_, g) in zip(views, need))
        if ok:
            torch._foreach_copy_(views, [g for _, g in need])
        for v, (i, g) in zip(views, need):
            gy[i] = v if ok else g.float()
    return gy


def _operators(layer: nn.Module) -> List[str]:
    """the operators the layer's parameters are read through, outermost first (imitation.py stacks one subclass per operator;
    the batcher's own hand-out subclass is not one of them)"""
    ops = []
    for cls in type(layer).__mro__:
        name = cls.__dict__.get("_qs_imitation")
        if name is not None:
            ops.append(name)
        elif isinstance(cls.__dict__.get("weight"), property) and "_qs_batcher_base" not in cls.__dict__:
            ops.append("?")                  # a weight property of unknown origin (user code): hands off
    return ops


def _unit_ok(q, p) -> bool:
    if not isinstance(q, QuantizeLayer) or p is None:
        return False
    qc = q.callback
    return (type(qc) in (ScalerQuantizer, DecimalQuantizer) and qc.group_num <= 0 and q.batch_dimension == -1 and q.timeout > 0
            and q.channelwise < p.dim())


def _prune_ok(p) -> bool:
    """a prune operator whose idle steps the kernels can stand in for: the stock magnitude policy, ranking by value"""
    if not isinstance(p, PruneLayer):
        return False
    cb = p.callback
    return type(cb) is MagnitudePruningCallback and not cb.use_gradient and not cb.l0 and cb.forward_hook is None


def _strides(t: torch.Tensor):
    """strides that matter: those of the dims with more than one element"""
    return tuple(st for st, n in zip(t.stride(), t.shape) if n > 1)


def _mask_geometry(w: torch.Tensor, mask: torch.Tensor):
    """(mask_C, mask_inner) of a broadcast mask over a dense weight, in the weight's MEMORY order (contiguous, or channels_last
    as `model.to(memory_format=torch.channels_last)` leaves 4-d parameters -- masks and magnitudes included): (0, 1) for a
    full-shape mask laid out like the weight, else the mask varies along one run of dims of that order -- its byte for the
    element at memory offset e is (e // mask_inner) % mask_C; None when it does not (or has one element)"""
    if w.dim() != mask.dim() or any(m not in (1, s) for m, s in zip(mask.shape, w.shape)):
        return None
    if w.is_contiguous():
        order = list(range(w.dim()))
    elif w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last):
        order = [0, 2, 3, 1]
    else:
        return None
    shape = [w.shape[d] for d in order]
    mshape = [mask.shape[d] for d in order]
    varying = [i for i, (m, s) in enumerate(zip(mshape, shape)) if m == s and s > 1]
    if not varying:
        return None
    if len(varying) == sum(1 for s in shape if s > 1):
        # full shape: the mask must sit in memory exactly like the weight
        return (0, 1) if _strides(mask) == _strides(w) else None
    lo, hi = varying[0], varying[-1]
    if any(shape[i] > 1 and mshape[i] == 1 for i in range(lo, hi + 1)):
        return None
    # the mask's own bytes along that run must be consecutive in the same order
    run = [order[i] for i in range(lo, hi + 1) if shape[i] > 1]
    expect = 1
    for d in reversed(run):
        if mask.stride(d) != expect:
            return None
        expect *= mask.shape[d]
    C = 1
    for i in range(lo, hi + 1):
        C *= shape[i]
    inner = 1
    for sdim in shape[hi + 1:]:
        inner *= sdim
    return (C, inner)


def _stage_plan(w: torch.Tensor, mask: torch.Tensor):
    """the stages of squeeze_tensor_to_shape(|w|, mask.shape) (reference util.py:79-99: one keepdim mean per reduced dim,
    ascending) as (layout, pre, n, post) tuples for `qs_multi_stage_mean`, or None for a layout it does not serve.  A contiguous
    weight: every stage is the mean over the middle dim of a contiguous [pre, n, post] tensor.  A channels_last weight whose
    first reduced dim is dim 0: ATen reduces the NHWC memory directly into an NCHW-contiguous result (layout 1), the later stages
    are contiguous ones (the route of `util._staged_mean_hip`)."""
    try:
        dims = _reduction_plan(w.shape, mask.shape)
    except (AssertionError, ValueError):
        return None
    if not dims:
        return None
    shape = list(w.shape)
    stages = []
    if not w.is_contiguous():
        if not (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last) and dims[0] == 0 and shape[0] > 1):
            return None
        stages.append((1, shape[2] * shape[3], shape[0], shape[1]))          # x = [n][hw][C] in memory -> [C][hw]
        shape[0] = 1
        dims = dims[1:]
    for d in dims:
        pre = 1
        for sdim in shape[:d]:
            pre *= sdim
        post = 1
        for sdim in shape[d + 1:]:
            post *= sdim
        stages.append((0, pre, shape[d], post))
        shape[d] = 1
    return tuple(stages)


def _eligible(layer: nn.Module) -> bool:
    """a layer whose weight is read through exactly one quantizer (tensor-wise or per channel) -- and, underneath it, at most a
    prune operator"""
    params = getattr(layer, "_parameters", None)
    ops = _operators(layer) if params is not None else None
    if ops == ["prune"]:             # a pruned weight without a quantizer: the prune operator alone
        w = params.get("weight")
        return _prune_ok(getattr(layer, "prune", None)) and w is not None and w.dim() > 1
    if ops != ["quantize"] and not (ops == ["quantize", "prune"] and _prune_ok(getattr(layer, "prune", None))):
        return False
    if not _unit_ok(getattr(layer, "quantize", None), params.get("weight")):
        return False
    qb = getattr(layer, "quantize_bias", None)
    if isinstance(qb, QuantizeLayer) and params.get("bias") is not None:
        # a bias quantizer takes part together with its weight's (one callback, one count `t`), or the layer stays inline
        if not (_unit_ok(qb, params["bias"]) and qb.callback is layer.quantize.callback and qb.timeout == layer.quantize.timeout):
            return False
    return True


def _callback_owners(model: nn.Module) -> dict:
    """callback id -> number of distinct owners: an owner is a wrapped layer (its weight and bias quantizers count once) or a
    free-standing QuantizeLayer (an activation operator)"""
    inside, owners = set(), {}
    for m in model.modules():
        if isinstance(m, QuantizeLayer):
            continue
        subs = [m.__dict__.get("_modules", {}).get(k) for k in _QUANT.values()]
        subs = [q for q in subs if isinstance(q, QuantizeLayer)]
        inside.update(id(q) for q in subs)
        for c in {id(q.callback) for q in subs if q.callback is not None}:
            owners[c] = owners.get(c, 0) + 1
    for m in model.modules():
        if isinstance(m, QuantizeLayer) and id(m) not in inside and m.callback is not None:
            owners[id(m.callback)] = owners.get(id(m.callback), 0) + 1
    return owners


def _batchable(model: nn.Module) -> List[nn.Module]:
    """the eligible layers of `model` that own their quantizer callback.  `convert` gives every layer a fresh copy, but a
    hand-built network may hand ONE callback object to several layers (the reference allows it: the callback's running-mean
    count `t` then advances once per layer read, in forward order) -- such layers keep the inline path, whose order of
    evaluation is the forward's own."""
    owners = _callback_owners(model)
    prune_owners = {}                 # a pruning callback shared between layers: one `t`, one magnitude -- inline as well
    for m in model.modules():
        if isinstance(m, PruneLayer) and m.callback is not None:
            prune_owners[id(m.callback)] = prune_owners.get(id(m.callback), 0) + 1

    def own_prune(m):
        p = m.__dict__.get("_modules", {}).get("prune")
        return not isinstance(p, PruneLayer) or prune_owners.get(id(p.callback), 0) == 1
    def own_quantizer(m):
        q = m.__dict__.get("_modules", {}).get("quantize")
        return not isinstance(q, QuantizeLayer) or owners.get(id(q.callback), 0) == 1
    return [m for m in model.modules() if _eligible(m) and own_quantizer(m) and own_prune(m)]


def _hooked(q: QuantizeLayer) -> bool:
    """hooks on the quantizer or its callback (or global module hooks) must see the calls they were registered for: such a
    layer keeps its inline path"""
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in (q, q.callback))


def _hooked_prune(p: PruneLayer) -> bool:
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in (p, p.callback))


def _prune_step(p: PruneLayer, w: torch.Tensor, training: bool):
    """what the prune operator underneath a weight's quantizer would do on this read, if that is something the multi-tensor
    kernels can do in its place -- None otherwise (the inline path then runs the operator itself).  Reads state, changes none.
    Returns (mask_geometry or None, counts n_updates, counts t, averages magnitude, threshold rank of a mask rebuild or None,
    stages of the importance's staged mean or None)."""
    if not _prune_ok(p) or _hooked_prune(p) or not p.initted or p.training != training:
        return None
    mask = p.mask
    if (not mask.is_cuda or mask.device != w.device or mask.dtype != torch.bool or mask.is_inference()
            or not p._n_updates.is_cuda or mask.numel() == 1):
        return None
    geo = _mask_geometry(w, mask)
    if geo is None:
        return None
    if not training:                              # PruneLayer.forward in evaluation: weight * mask, nothing else
        return (geo, False, False, False, None, None)
    n = p._steps.read(p._n_updates)
    if n in p.schedules:
        return None                               # the sparsity changes on this read
    if n < p.start:
        return (None, True, False, False, None, None)   # not pruning yet: the layer only counts
    cb = p.callback
    if not cb.initted or not cb.t.is_cuda or cb.t.device != w.device:
        return None
    t = cb._t_host.read(cb.t)
    sparsity = p.current_sparsity()
    refresh = cb.refresh_due(t, sparsity)
    average = t < cb.stop_mask_refresh and cb.running_average
    if not (average or refresh):
        return (geo, True, True, False, None, None)
    # this read averages the running magnitude and / or rebuilds the mask (sparse.py:82-89, 58-66): the importance is
    # squeeze_tensor_to_shape(|weight|, mask.shape) -- element-wise for a full-shape mask, a staged mean otherwise -- which the
    # kernels compute for all such layers at once.  Under a process group the reference path averages it over the ranks: inline.
    if qdist.exchange_active(qdist.stats_world_size()):
        return None
    mag = getattr(cb, "magnitude", None)
    stages = None
    if geo == (0, 1):
        if _strides(mask) != _strides(w):
            return None
        mag_ok = mag is not None and mag.shape == w.shape and _strides(mag) == _strides(w)
    else:
        stages = _stage_plan(w, mask)
        if stages is None or not mask.is_contiguous():
            return None
        mag_ok = mag is not None and mag.shape == mask.shape and mag.is_contiguous()
    if cb.running_average and not (mag_ok and mag.is_cuda and mag.dtype == torch.float32):
        return None
    k = None
    if refresh:
        n_el = mask.numel()
        k = threshold_rank(sparsity, n_el)
        if k >= n_el or n_el >= 2 ** 32:
            return None                           # (k >= n: the inline path raises the reference's IndexError)
    return (geo, True, True, average, k, stages)


class _LaunchPlan(dict):
    """cached launch table (ctypes + device memory): never copied or pickled with the network"""

    def __deepcopy__(self, memo):
        return _LaunchPlan()

    def __reduce__(self):
        return (_LaunchPlan, ())


class _Unit:
    """one tensor of the multi-tensor launches: a layer's weight or bias with the QuantizeLayer it is read through"""
    __slots__ = ("layer", "attr", "slot", "channels")

    def __init__(self, layer, attr, slot, channels):
        self.layer, self.attr, self.slot, self.channels = layer, attr, slot, channels

    @property
    def q(self) -> Optional[QuantizeLayer]:
        q = self.layer.__dict__.get("_modules", {}).get(_QUANT[self.attr])
        return q if isinstance(q, QuantizeLayer) else None       # (None: a pruned weight without a quantizer)

    @property
    def param(self) -> torch.Tensor:
        return self.layer._parameters[self.attr]

    @property
    def p(self) -> Optional[PruneLayer]:
        """the prune operator underneath the weight's quantizer, if there is one"""
        if self.attr != "weight":
            return None
        p = self.layer.__dict__.get("_modules", {}).get("prune")
        return p if isinstance(p, PruneLayer) else None


class _Pending:
    """what has to be undone if a precomputed tensor is never read"""
    __slots__ = ("unit", "was_quantized", "t_dev", "version", "training", "dead", "index", "prune", "plain")

    def __init__(self, unit, was_quantized, t_dev, version, training, prune=None, plain=None):
        self.unit, self.was_quantized, self.t_dev, self.version, self.training = unit, was_quantized, t_dev, version, training
        self.dead, self.index = None, 0     # the hand-out group's "rolled back" flags and this tensor's place in them
        self.prune = prune                  # (counted n_updates, counted t, magnitude backup, mask backup) of a pruned weight
        self.plain = plain                  # not quantized on this read: None, or whether an idle quantizer counted the read


def _patched_class(base):
    """subclass of `base` whose `weight` / `bias` hand out a waiting precomputed tensor once, else read like `base`.  The
    properties find everything through the instance (nothing closes over a batcher), so a deep copy of the network keeps
    working on its own."""

    def reader(attr):
        def read(self_):
            entry = self_.__dict__.pop(_READY[attr], None)
            if entry is not None:
                y, pending, batcher = entry
                waiting_weight = self_.__dict__.get(_READY["weight"]) if attr == "bias" else None
                if waiting_weight is not None and waiting_weight[1].training:
                    # the bias is read BEFORE the weight: the reference would update the bias statistics with t (not t + 1)
                    # and the weight's with t + 1 -- undo both, the inline path follows the order of the reads
                    batcher._rollback(waiting_weight[1])
                    batcher._rollback(pending)
                elif self_._parameters[attr]._version == pending.version:
                    batcher._consumed(pending)
                    return y
                else:
                    batcher._rollback(pending)       # the parameter was written since the precomputation: evaluate inline
                    waiting_bias = self_.__dict__.get(_READY["bias"]) if attr == "weight" else None
                    if waiting_bias is not None and waiting_bias[1].training:
                        # ... and the bias after it: its precomputed update counted on the weight's having advanced the shared
                        # count first, which the inline evaluation is about to do again
                        batcher._rollback(waiting_bias[1])
            return getattr(base, attr).__get__(self_)
        return read

    return type(base.__name__, (base,), {"weight": property(reader("weight")), "bias": property(reader("bias")),
                                         "_qs_batcher_base": base})


def _unit_config(q: QuantizeLayer, qc):
    """every configuration attribute the launch table and the hand-out depend on (plain Python attributes: cheap to re-read)"""
    return (q.bits, q.channelwise, q.timeout, q.batch_dimension, type(qc), qc.use_float_scaler, qc.flip_axis, qc.backward_passthrough,
            qc.use_uint, qc.group_num, qc.__dict__.get("saturate"))


class _SteadyUnit:
    __slots__ = ("unit", "layer", "attr", "cls", "w", "w_ptr", "w_stride", "q", "qc", "qw", "qw_ptr", "qn", "qn_ptr", "t_dev", "hooks",
                 "config", "view", "key")


class _Steady:
    """what a training step of the weight path looks like once every tensor is quantized on every read (all timeouts passed, no
    prune operator underneath, nothing frozen): the launch table of the last full precomputation and, per tensor, the identities
    that table and the hand-out depend on.  The following steps only compare those (parameter object, storage and strides; the
    quantizer, its callback, its state tensors; hooks; configuration attributes; the options epoch; the step counter through
    its host mirror) and issue the same launches; anything else takes the full path, which re-arms."""
    __slots__ = ("epoch", "units", "plan", "todo", "weights", "metas", "static_steps", "dev")

    def __deepcopy__(self, memo):        # raw pointers and object identities: a copied network arms its own
        return None

    def __reduce__(self):
        return (_no_steady, ())


def _no_steady():
    return None


class _ImageStat:
    """what `fused.AutocastImageTensor` expects of the object that made an image"""
    __slots__ = ("image_made", "image_used")

    def __init__(self):
        self.image_made = self.image_used = False

    def __deepcopy__(self, memo):
        return _ImageStat()


def _weight_images(flat, units_attr, weights, offsets):
    """the autocast images of a step's quantized weights: ONE cast of the flat float32 buffer (instead of the cast autocast puts
    in front of every convolution / linear), viewed per weight like the float32 outputs; None for biases and outside autocast.
    Value-identical: RNE(y), exactly what that cast produces (fused.py "Autocast image")."""
    from qsparse_amd.fused import autocast_image_dtype
    dt = autocast_image_dtype()
    if dt is None or flat is None:
        return None
    flat16 = flat.to(dt)
    so = flat16.storage_offset()
    return [flat16.as_strided(w.shape, w.stride(), so + o) if (attr == "weight" and w.dim() >= 2) else None
            for attr, w, o in zip(units_attr, weights, offsets)]


def _dual_outputs(ys, imgs, k, stat):
    """the group's outputs as the layers receive them: a weight that has an image goes out as the Tensor subclass that carries it
    to its first autocast consumer (gradient slots: output i, image k + j of the hand-out node)"""
    from qsparse_amd.fused import _as_dual
    out, j = [], 0
    for i in range(k):
        if imgs is not None and imgs[i] is not None:
            out.append(_as_dual(ys[i], ys[k + j], stat, slots=(i, k + j)))
            j += 1
        else:
            out.append(ys[i])
    return out


class WeightBatcher:
    def __init__(self, model: nn.Module):
        for m in model.modules():                # one batcher per tree: an earlier one (convert installs one) steps aside
            old = m.__dict__.get(_ATTR)
            if old is not None:
                old.remove()
        self.model = model
        self.layers: List[nn.Module] = _batchable(model)
        self.units: List[_Unit] = []
        total = 0
        for layer in self.layers:
            for attr in ("weight", "bias"):
                q, p = getattr(layer, _QUANT[attr], None), layer._parameters.get(attr)
                if isinstance(q, QuantizeLayer) and p is not None:
                    C = 1 if q.channelwise < 0 else p.shape[q.channelwise]
                    self.units.append(_Unit(layer, attr, total, C))
                    total += C
                elif attr == "weight" and p is not None and not isinstance(getattr(layer, "quantize", None), QuantizeLayer):
                    self.units.append(_Unit(layer, attr, total, 0))       # prune only: no scale, no channel slot
        self._channels = total
        self._pending: List[_Pending] = []
        self._amax = None
        self._decimals = None
        self._backup = None
        self._eval_key = None
        self._eval_outs = None
        self._eval_decimals = None
        self._plan = None
        self._steady = None
        self._eval_flat = self._eval_offsets = None
        self._image_stat = _ImageStat()
        for layer in self.layers:
            if "_qs_batcher_base" not in type(layer).__dict__:
                layer.__class__ = _patched_class(type(layer))
        self._hook = model.register_forward_pre_hook(self._precompute)
        self._post = model.register_forward_hook(self._finish, always_call=True)
        model.__dict__[_ATTR] = self

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def install(model: nn.Module) -> Optional["WeightBatcher"]:
        """(re-)install on `model`: batchers found anywhere in its tree are removed first (a further ``convert`` changes
        the set of layers); returns None when no layer is eligible"""
        if not _batchable(model):
            for m in model.modules():
                old = m.__dict__.get(_ATTR)
                if old is not None:
                    old.remove()
            return None
        return WeightBatcher(model)

    def invalidate(self):
        """forget the quantized weights kept for evaluation.  They are reused, under ``torch.no_grad()`` only, while no
        parameter and no scale has been written -- detected through ``Tensor._version``, which optimizers,
        ``load_state_dict`` and any in-place op bump; writes through ``param.data`` do not: call this (or
        ``qs.resync_host_state(model)``, which does) after such a write between two no-grad evaluation forwards."""
        self._eval_key = self._eval_outs = self._eval_decimals = self._eval_flat = self._eval_offsets = None

    def remove(self):
        self._rollback_all()
        self._hook.remove()
        self._post.remove()
        for layer in self.layers:
            base = type(layer).__dict__.get("_qs_batcher_base")
            if base is not None:
                layer.__class__ = base
            for key in _READY.values():
                layer.__dict__.pop(key, None)
        self.model.__dict__.pop(_ATTR, None)

    # ------------------------------------------------------------------------------------------
    def _consumed(self, pending: _Pending):
        try:
            self._pending.remove(pending)
        except ValueError:
            pass

    def _rollback(self, p: _Pending):
        """put the tensor's quantizer back where it was before `_precompute` advanced it (stream-ordered device writes, no sync)"""
        self._consumed(p)
        u = p.unit
        u.layer.__dict__.pop(_READY[u.attr], None)
        if p.dead is not None:
            p.dead[p.index] = True               # its place in the group's autograd node delivers no gradient
        if not p.training:
            return                               # evaluation-mode hand-outs change no state
        q = u.q
        with torch.no_grad():
            if p.plain is not None:              # not quantized on that read: at most an idle quantizer counted it
                if p.plain:
                    q._steps.add(q._n_updates, -1)
            else:
                qc = q.callback
                q.weight.data.view(-1).copy_(self._backup[u.slot:u.slot + u.channels])
                q._steps.add(q._n_updates, -1)
                qc.t -= 1
                if p.t_dev is not None:
                    p.t_dev.sub_(1)
                    qc.__dict__["_t_dev_value"] = qc.t
            if p.prune is not None:          # the prune operator underneath: its counters and its running magnitude
                pl = u.p
                counted_n, counted_t, mag_backup, mask_backup = p.prune
                if counted_n:
                    pl._steps.add(pl._n_updates, -1)
                if counted_t:
                    pl.callback._t_host.add(pl.callback.t, -1)
                if mag_backup is not None:
                    pl.callback.magnitude.data.copy_(mag_backup)      # (same shape and strides: element by element)
                if mask_backup is not None:
                    pl.mask.data.copy_(mask_backup)
        if p.plain is None:
            q._quantized = p.was_quantized

    def _rollback_all(self):
        for p in list(self._pending):
            self._rollback(p)

    def _finish(self, module, args, output=None):
        """forward hook of the root (also after a forward that raised): whatever was precomputed and not read is undone"""
        if self._pending:
            self._rollback_all()

    # ------------------------------------------------------------------------------------------
    def _precompute(self, module, args):
        if module is not self.model:             # a replica (nn.DataParallel) shares the hook but not the layers
            return
        if self._pending:                        # leftovers of a forward whose hook could not run
            self._rollback_all()
        if not get_option("batch_weights"):
            return
        steady = self._steady
        if steady is not None:
            if self._run_steady(steady):
                return
            self._steady = None
        train, frozen = [], []          # tensors that update statistics this step / that only quantize
        plain = {}                      # unit -> whether an idle quantizer counts the read: pruned weights that are NOT quantized now
        prune_steps = {}                # id(layer) -> what its prune operator does on this read (`_prune_step`)
        skip_layer = None
        for u in self.units:
            layer = u.layer
            if layer is skip_layer:
                continue                # the weight does not take part this step: neither does its bias (shared count)
            q, w = u.q, u.param
            if q is None:               # prune(conv) without a quantizer: the prune operator alone, y = weight * mask
                pl = u.p
                ok = (pl is not None and w.is_cuda and w.dtype == torch.float32 and _dense(w) and w.data_ptr() % 4 == 0
                      and "_qs_batcher_base" in type(layer).__dict__ and not layer._forward_pre_hooks and not w.is_inference())
                ps = _prune_step(pl, w, pl.training) if ok else None
                if ps is not None:
                    prune_steps[id(layer)] = ps
                    plain[u] = False
                continue
            geo = _geometry(w, q.channelwise)
            ok = (w.is_cuda and w.dtype == torch.float32 and geo is not None and w.data_ptr() % 4 == 0
                  and (u.channels == 1 or geo[1] == u.channels))
            # re-wrapped since (a further imitation); hooks that want to see the quantizer's calls; a pre-hook on the layer (code
            # that runs between this precomputation and the read); first read ever (the inline path creates the layer's state
            # when -- and if -- it happens); state on another device; tensors created under torch.inference_mode() (no
            # version counter to see a write with)
            ok = ok and "_qs_batcher_base" in type(layer).__dict__ and not _hooked(q) and not layer._forward_pre_hooks
            ok = ok and q.initted and q.weight.is_cuda and q._n_updates.is_cuda and q.weight.numel() == u.channels
            ok = ok and q.weight.is_contiguous() and not (w.is_inference() or q.weight.is_inference())
            t = q._steps.read(q._n_updates) if ok else 0
            took_part = False
            pl = u.p
            if ok and pl is not None and (t < q.timeout or not (q.training or q._quantized)):
                # the quantizer above a pruned weight is in its identity phase (it only counts the read, quantize.py:496-517): the
                # prune operator alone -- its bias quantizer, if any, stays inline this step
                ps = _prune_step(pl, w, q.training)
                if ps is not None:
                    prune_steps[id(layer)] = ps
                    plain[u] = bool(q.training)
                skip_layer = layer
                continue
            if ok and pl is not None:
                # a pruned weight: only on the steps its prune operator leaves to the kernels (see `_prune_step`)
                prune_steps[id(layer)] = _prune_step(pl, w, q.training)
                ok = prune_steps[id(layer)] is not None
            if ok and t >= q.timeout:     # (below the timeout: identity phase, the inline path only counts)
                if q.training:
                    if t == q.timeout and get_option("log_during_train"):
                        logging.warn(f"quantizing {q.name} with {q.bits} bits")
                    train.append(u)
                    took_part = True
                elif q._quantized:
                    frozen.append(u)
                    took_part = True
            if not took_part and u.attr == "weight":
                skip_layer = layer
        if not train and not frozen and not plain:
            return
        todo = train + frozen + list(plain)
        plain_training = any(u.p.training for u in plain)
        weights = [u.param for u in todo]
        dev = weights[0].device
        if any(w.device != dev for w in weights):
            return
        if self._amax is None or self._amax.device != dev:
            self._amax = torch.zeros(max(self._channels, 1), dtype=torch.float32, device=dev)
            self._decimals = torch.zeros(max(self._channels, 1), dtype=torch.float32, device=dev)
            self._backup = torch.zeros(max(self._channels, 1), dtype=torch.float32, device=dev)
        sats = [None if u in plain else u.q.callback.code_range(u.q.bits) for u in todo]
        # evaluation / serving: nothing changes between calls unless someone writes a parameter or a scale (both bump
        # `_version`), so the quantized weights of the previous call are handed out again without a launch
        eval_key = None
        # (only under no_grad / inference_mode: a loop that runs the network in eval() WITH gradients -- fine-tuning with
        # frozen statistics, manual SGD or EMA through `p.data.add_()` -- may write parameters by the one route the version
        # counter does not see, and would be handed stale weights silently; such loops pay the launch per forward)
        if not train and not plain_training and not torch.is_grad_enabled():
            eval_key = tuple((id(u.layer), u.attr, w.data_ptr(), w._version, tuple(w.stride()), sat)
                             + ((u.q.weight.data_ptr(), u.q.weight._version) if u not in plain else ())
                             + ((u.p.mask.data_ptr(), u.p.mask._version) if u.p is not None else ())
                             for u, w, sat in zip(todo, weights, sats))
            if eval_key == self._eval_key:
                self._hand_out(todo, weights, self._eval_outs, {}, self._eval_decimals, prune_steps, plain, flat=self._eval_flat,
                               offsets=self._eval_offsets)
                return
        self._eval_key = None
        undo = {}
        with torch.no_grad():
            t_devs = [u.q.callback.device_t(dev) if i < len(train) else None for i, u in enumerate(todo)]       # (train units come first)
            # everything that does not change from step to step -- the launch table, the layout of the flat output buffer --
            # is built once per (set of tensors, parameter storage, state storage) and reused
            # (pruned weights: the mask and what the prune operator does this step are part of the table)
            psteps = [prune_steps.get(id(u.layer)) if u.attr == "weight" else None for u in todo]
            key = (len(train), tuple(sats)) + tuple(
                (id(u.layer), u.attr, w.data_ptr(), tuple(w.stride()), None if td is None else td.data_ptr())
                + ((u.q.weight.data_ptr(), u.q._n_updates.data_ptr()) if u not in plain
                   else ("plain", u.q._n_updates.data_ptr() if plain[u] else None))
                + ((ps, u.p.mask.data_ptr(), u.p._n_updates.data_ptr(), u.p.callback.t.data_ptr(),
                    u.p.callback.magnitude.data_ptr() if ((ps[3] or ps[4] is not None) and u.p.callback.running_average) else None)
                   if ps is not None else ())
                for u, w, td, ps in zip(todo, weights, t_devs, psteps))
            plans = self._plan if isinstance(self._plan, _LaunchPlan) else _LaunchPlan()     # (key -> table; never copied)
            plan = plans.get(key)
            if plan is None:
                # (a handful of tables: reads that rebuild masks alternate with reads that do not, training with evaluation)
                if len(plans) >= 4:
                    plans.pop(next(iter(plans)))
                plan = plans[key] = self._build_plan(key, todo, weights, t_devs, sats, len(train), dev, psteps, plain)
                self._plan = plans
            table = plan["table"]
            if train or plain_training:
                # the prune operators first, as on every read (sparse.py:99-122): running magnitudes, then the masks that are due
                # -- the quantizers' abs-max below sees weight * (new) mask
                for stage_table in plan["stage_tables"]:       # importances that are staged means, one launch per stage level
                    _hip.multi_stage_mean(stage_table, nbytes=plan["stage_bytes"])
                if plan["mag_backups"]:
                    _hip.multi_magnitude(table, nbytes=plan["mag_bytes"])
                if plan["mask_backups"]:
                    _hip.multi_mask_refresh(table, nbytes=plan["refresh_bytes"])
                if train:
                    _hip.multi_absmax(table, nbytes=plan["train_bytes"])
                    _hip.multi_scale_update(table)
                for i, u in enumerate(todo):
                    if u in plain and u.p.training:      # not quantized on this read: the prune operator's bookkeeping, an idle quantizer's count
                        ps, pl = psteps[i], u.p
                        if ps[1]:
                            pl._steps.note_device_add(pl._n_updates, 1)
                        if ps[2]:
                            pl.callback._t_host.note_device_add(pl.callback.t, 1)
                        if plain[u]:
                            u.q._steps.note_device_add(u.q._n_updates, 1)
                        undo[(id(u.layer), u.attr)] = _Pending(u, None, None, u.param._version, True,
                                                                (ps[1], ps[2], plan["mag_backups"].get(i), plan["mask_backups"].get(i)),
                                                                plain=plain[u])
                for i, (u, t_dev) in enumerate(zip(train, t_devs)):
                    q, qc = u.q, u.q.callback
                    ps, prune_undo = psteps[i], None
                    if ps is not None:          # the prune operator's own bookkeeping of this read (sparse.py:117,272)
                        pl = u.p
                        if ps[1]:
                            pl._steps.note_device_add(pl._n_updates, 1)
                        if ps[2]:
                            pl.callback._t_host.note_device_add(pl.callback.t, 1)
                        prune_undo = (ps[1], ps[2], plan["mag_backups"].get(i), plan["mask_backups"].get(i))
                    undo[(id(u.layer), u.attr)] = _Pending(u, q._quantized, t_dev, u.param._version, True, prune_undo)
                    qc._advance_t(t_dev, bumped_by_kernel=True)
                    q._quantized = True
                    q._steps.note_device_add(q._n_updates, 1)
            for u in frozen:       # evaluation: the decimals of the frozen scales (the inline path recomputes them per call as well)
                if not u.q.callback.use_float_scaler:
                    self._decimals[u.slot:u.slot + u.channels] = _hip.decimal_from_scale(u.q.weight.data.view(-1))
            flat = torch.empty(plan["total"], dtype=torch.float32, device=dev)
            so = flat.storage_offset()
            outs = [flat.as_strided(w.shape, w.stride(), so + o) for o, w in zip(plan["offsets"], weights)]   # w's own layout
            for i, (u, w) in enumerate(zip(todo, weights)):
                # the reference broadcasts the input against the parameter (`input / scaler`): a tensor-wise (1, 1) scale turns a
                # 1-d bias into a (1, C) tensor (quantize.py `_reference_shape`; nn.Linear takes it, nn.Conv2d rejects it)
                if u not in plain and u.q.weight.numel() == 1 and u.q.weight.dim() > w.dim():
                    outs[i] = outs[i].view(torch.broadcast_shapes(tuple(w.shape), tuple(u.q.weight.shape)))
            _hip.multi_quant_fwd(table, flat, advance=bool(train) or plain_training, nbytes=plan["all_bytes"])
            # a DecimalQuantizer's backward clamps with the decimal of ITS forward (the reference computes a fresh tensor per
            # call, quantize.py:312-325, and the Function saves that one, :41): the hand-out nodes get this step's values, not
            # the buffer the next precomputation overwrites (a ScalerQuantizer's saves the scale parameter itself, :108)
            decimals = self._decimals.clone() if plan["any_decimal"] else self._decimals
        if eval_key is not None:
            self._eval_key, self._eval_outs, self._eval_decimals = eval_key, outs, decimals
            self._eval_flat, self._eval_offsets = flat, plan["offsets"]
        self._hand_out(todo, weights, outs, undo, decimals, prune_steps, plain, flat=flat, offsets=plan["offsets"])
        if (train and len(train) == len(self.units) and not frozen and not plain and not any(ps is not None for ps in psteps)
                and not get_option("log_during_train") and os.environ.get("QS_NO_FAST_PATH", "0") != "1"):
            self._steady = self._arm_steady(train, weights, t_devs, plan, dev)

    # ------------------------------------------------------------------------------------------
    def _arm_steady(self, todo, weights, t_devs, plan, dev) -> Optional[_Steady]:
        st = _Steady()
        st.epoch, st.plan, st.todo, st.weights, st.dev = _options_epoch[0], plan, todo, weights, dev
        st.units = []
        for u, w, t_dev in zip(todo, weights, t_devs):
            q = u.q
            qc = q.callback
            if q._steps.read(q._n_updates) <= q.timeout:
                return None              # (the step AT the timeout logs; arm from the next one on)
            c = _SteadyUnit()
            c.unit, c.layer, c.attr, c.cls = u, u.layer, u.attr, type(u.layer)
            c.w, c.w_ptr, c.w_stride = w, w.data_ptr(), w.stride()
            c.q, c.qc, c.qw, c.qn, c.t_dev = q, qc, q.weight, q._n_updates, t_dev
            c.qw_ptr, c.qn_ptr = q.weight.data_ptr(), q._n_updates.data_ptr()
            c.hooks = tuple(d for m in (q, qc) for d in (m._forward_hooks, m._forward_pre_hooks, m._backward_hooks, m._backward_pre_hooks))
            c.hooks += (u.layer._forward_pre_hooks,)
            c.config = _unit_config(q, qc)
            # (the reference broadcasts the input against the parameter: a tensor-wise (1, 1) scale turns a 1-d bias into (1, C))
            c.view = (torch.broadcast_shapes(tuple(w.shape), tuple(q.weight.shape))
                      if (q.weight.numel() == 1 and q.weight.dim() > w.dim()) else None)
            c.key = (id(u.layer), u.attr)
            st.units.append(c)
        # the hand-out groups: everything but the tensors themselves is the same on every step
        st.metas, st.static_steps = [], []
        for base in range(0, len(todo), _GROUP):
            meta, steps = [], []
            for u in todo[base:base + _GROUP]:
                q, qc = u.q, u.q.callback
                is_decimal = not qc.use_float_scaler
                limit = 2.0 ** (q.bits - 1)
                notch = 1 if qc.flip_axis else 0
                meta.append((is_decimal, -limit + notch, limit - 1 + notch, bool(qc.backward_passthrough), q.channelwise, None))
                steps.append(None if is_decimal else q.weight.data)
            st.metas.append(tuple(meta))
            st.static_steps.append(steps)
        return st

    def _run_steady(self, st: _Steady) -> bool:
        """one steady-state training step of the weight path: True when it ran, False -- nothing touched -- when something the
        cached table depends on has changed"""
        if (st.epoch != _options_epoch[0] or _m._global_forward_hooks or _m._global_forward_pre_hooks or _hip.logging_events()
                or len(st.units) != len(self.units)):
            return False
        for c in st.units:
            layer = c.layer
            w = layer._parameters.get(c.attr)
            if w is not c.w or w.data_ptr() != c.w_ptr or w.stride() != c.w_stride or type(layer) is not c.cls:
                return False
            q = layer._modules.get(_QUANT[c.attr])
            if q is not c.q or not q.training or q._modules.get("callback") is not c.qc or not q._quantized:
                return False
            qp = q._parameters
            if (qp.get("weight") is not c.qw or qp.get("_n_updates") is not c.qn or c.qw.data_ptr() != c.qw_ptr
                    or c.qn.data_ptr() != c.qn_ptr):
                return False
            for d in c.hooks:
                if d:
                    return False
            qc = c.qc
            if (_unit_config(q, qc) != c.config or q._steps.read(c.qn) <= q.timeout or qc.__dict__.get("_t_dev") is not c.t_dev
                    or qc.__dict__.get("_t_dev_value") != qc.t):
                return False         # (a stale device copy of the count: `device_t` rebuilds it on the full path)
        plan = st.plan
        table = plan["table"]
        undo = {}
        with torch.no_grad():
            _hip.multi_absmax(table, nbytes=plan["train_bytes"])
            _hip.multi_scale_update(table)
            for c in st.units:
                q = c.q
                undo[c.key] = _Pending(c.unit, True, c.t_dev, c.w._version, True, None)
                c.qc._advance_t(c.t_dev, bumped_by_kernel=True)
                q._steps.note_device_add(c.qn, 1)
            flat = torch.empty(plan["total"], dtype=torch.float32, device=st.dev)
            so = flat.storage_offset()
            outs = []
            for c, o in zip(st.units, plan["offsets"]):
                y = flat.as_strided(c.w.shape, c.w_stride, so + o)
                outs.append(y if c.view is None else y.view(c.view))
            _hip.multi_quant_fwd(table, flat, advance=True, nbytes=plan["all_bytes"])
            decimals = self._decimals.clone() if plan["any_decimal"] else self._decimals
        self._eval_key = None
        todo, weights = st.todo, st.weights
        imgs = _weight_images(flat, [c.attr for c in st.units], weights, plan["offsets"])
        for gi, base in enumerate(range(0, len(todo), _GROUP)):
            group = st.units[base:base + _GROUP]
            steps = [s if s is not None else decimals[c.unit.slot:c.unit.slot + c.unit.channels].view(-1, 1)
                     for s, c in zip(st.static_steps[gi], group)]
            dead = [False] * len(group)
            gimgs = imgs[base:base + _GROUP] if imgs is not None else None
            ys = _GroupSte.apply(st.metas[gi], dead, *weights[base:base + _GROUP], *outs[base:base + _GROUP], *steps,
                                 *(gimgs if gimgs is not None else ()))
            ys = _dual_outputs(ys, gimgs, len(group), self._image_stat)
            for i, (c, y) in enumerate(zip(group, ys)):
                pending = undo[c.key]
                pending.dead, pending.index = dead, i
                self._pending.append(pending)
                c.layer.__dict__[_READY[c.attr]] = (y, pending, self)
        return True

    def _build_plan(self, key, todo, weights, t_devs, sats, n_train, dev, psteps, plain) -> _LaunchPlan:
        offsets, total = [], 0
        for w in weights:
            offsets.append(total)
            total += (w.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        rows, keep = [], []
        mag_backups, mag_bytes = {}, 0
        mask_backups, refresh_bytes = {}, 0
        levels, mask_rows, stage_bytes = {}, [], 0       # staged-mean launch tables per stage level; mask-level rows; bytes read
        trains_weight = {id(u.layer) for u in todo[:n_train] if u.attr == "weight"}
        for i, (u, w, sat) in enumerate(zip(todo, weights, sats)):
            r = _hip.MultiRow()
            if u in plain:               # not quantized on this read (kind 2): y = weight * mask, an idle quantizer counts
                keep.append(w)
                r.kind, r.x = 2, w.data_ptr()
                r.numel, r.y_off, r.outer, r.inner, r.C = w.numel(), offsets[i], 1, max(w.numel(), 1), 1
                r.code_lo, r.code_hi, r.denom = 1, 0, 1.0
                if plain[u]:
                    r.bump = u.q._n_updates.data_ptr()
                    keep.append(u.q._n_updates)
            else:
                q, qc = u.q, u.q.callback
                outer, C, inner = _geometry(w, q.channelwise)
                sl = slice(u.slot, u.slot + u.channels)
                is_decimal = not qc.use_float_scaler
                amax, dec, backup = self._amax[sl], self._decimals[sl], self._backup[sl]
                keep += [amax, dec, backup, t_devs[i], q.weight, q._n_updates, w]
                r.x, r.scale = w.data_ptr(), q.weight.data_ptr()
                r.amax, r.backup = amax.data_ptr(), backup.data_ptr()
                r.decimal = dec.data_ptr() if is_decimal else None
                r.t_dev = t_devs[i].data_ptr() if t_devs[i] is not None else None
                r.bump = q._n_updates.data_ptr()
                r.numel, r.y_off, r.outer, r.inner, r.C = w.numel(), offsets[i], outer, inner, C
                r.train, r.is_decimal = int(i < n_train), int(is_decimal)
                # the bias quantizer shares its weight quantizer's callback: when both update this step the bias sees t + 1
                r.t_offset = int(u.attr == "bias" and i < n_train and id(u.layer) in trains_weight)
                r.code_lo, r.code_hi = (1, 0) if sat is None else (int(sat[0]), int(sat[1]))
                r.denom = float(2 ** (q.bits - 1))
            ps = psteps[i]
            if ps is not None:                   # a pruned weight (see `_prune_step`)
                pl = u.p
                geo, counts_n, counts_t, averages, rank, stages = ps
                if geo is not None:
                    r.mask, r.mask_C, r.mask_inner = pl.mask.data_ptr(), geo[0], geo[1]
                    keep.append(pl.mask)
                if counts_n:
                    r.prune_n_updates = pl._n_updates.data_ptr()
                    keep.append(pl._n_updates)
                if counts_t:
                    r.prune_t = pl.callback.t.data_ptr()
                    keep.append(pl.callback.t)
                # who averages / re-ranks: the weight's own row (full-shape mask: the importance is |weight| element by element) or a
                # mask-level row whose x is the staged mean this step's stage launches leave in `imp`
                target, n_imp = r, w.numel()
                if stages is not None:
                    cur, n_imp = w, pl.mask.numel()
                    for level, (layout, pre, n_red, post) in enumerate(stages):
                        out = torch.empty(pre * post, dtype=torch.float32, device=dev)
                        st = _hip.MultiStage()
                        st.x, st.out, st.pre, st.n, st.post = cur.data_ptr(), out.data_ptr(), pre, n_red, post
                        st.take_abs, st.layout = int(level == 0), layout
                        levels.setdefault(level, []).append(st)
                        keep += [cur, out]
                        stage_bytes += 4 * pre * n_red * post
                        cur = out
                    target = _hip.MultiRow()
                    target.kind, target.x = 1, cur.data_ptr()
                    target.numel, target.outer, target.C, target.inner = n_imp, 1, 1, n_imp
                    target.code_lo, target.code_hi, target.denom = 1, 0, 1.0
                    target.prune_t = pl.callback.t.data_ptr()         # (read by the running mean; advanced through the weight's row)
                    mask_rows.append(target)
                if averages:
                    backup = torch.empty_like(pl.callback.magnitude)        # the magnitude's own memory layout
                    target.magnitude, target.mag_backup = pl.callback.magnitude.data_ptr(), backup.data_ptr()
                    keep += [pl.callback.magnitude, backup]
                    mag_backups[i] = backup
                    mag_bytes += 16 * n_imp
                if rank is not None:             # this read rebuilds the mask
                    cb = pl.callback
                    state = torch.zeros(258, dtype=torch.int32, device=dev)
                    mbackup = torch.empty_like(pl.mask)
                    target.refresh, target.select_k = 1, int(rank)
                    target.importance = cb.magnitude.data_ptr() if cb.running_average else None
                    target.select_state, target.mask_backup = state.data_ptr(), mbackup.data_ptr()
                    if target is not r:
                        target.mask, target.mask_C, target.mask_inner = pl.mask.data_ptr(), 0, 1
                    keep += [state, mbackup, pl.mask] + ([cb.magnitude] if cb.running_average else [])
                    mask_backups[i] = mbackup
                    refresh_bytes += (4 * 4 + 6) * n_imp
            rows.append(r)
        rows += mask_rows                                 # (after the tensors: `offsets` and the callers index rows by unit)
        stage_tables = [_hip.StageTable(levels[k], dev) for k in sorted(levels)]
        return _LaunchPlan(key=key, offsets=offsets, total=total, keep=keep, table=_hip.MultiTable(rows, dev),
                           stage_tables=stage_tables, stage_bytes=stage_bytes,
                           mag_backups=mag_backups, mag_bytes=mag_bytes, mask_backups=mask_backups, refresh_bytes=refresh_bytes,
                           train_bytes=4 * sum(w.numel() for w in weights[:n_train]), all_bytes=8 * sum(w.numel() for w in weights),
                           any_decimal=any(u not in plain and not u.q.callback.use_float_scaler for u in todo))

    def _hand_out(self, todo, weights, outs, undo, decimals, prune_steps=None, plain=(), flat=None, offsets=None):
        """park every quantized tensor on its layer, `_GROUP` consecutive tensors per autograd node (a node per layer in
        evaluation mode under no_grad costs nothing either way)"""
        imgs = _weight_images(flat, [u.attr for u in todo], weights, offsets) if offsets is not None else None
        for base in range(0, len(todo), _GROUP):
            group = todo[base:base + _GROUP]
            meta, steps = [], []
            for u in group:
                ps = prune_steps.get(id(u.layer)) if (prune_steps and u.attr == "weight") else None
                mask = (u.p.mask, ps[0][0], ps[0][1], _strides(u.param)) if (ps is not None and ps[0] is not None) else None
                if u in plain:      # not quantized on this read: the gradient passes (times the mask: the backward of weight * mask)
                    meta.append((False, 0.0, 0.0, True, -1, mask))
                    steps.append(decimals[:1])
                    continue
                q, qc = u.q, u.q.callback
                is_decimal = not qc.use_float_scaler
                limit = 2.0 ** (q.bits - 1)
                notch = 1 if qc.flip_axis else 0
                meta.append((is_decimal, -limit + notch, limit - 1 + notch, bool(qc.backward_passthrough), q.channelwise, mask))
                steps.append(decimals[u.slot:u.slot + u.channels].view(-1, 1) if is_decimal else q.weight.data)
            dead = [False] * len(group)
            gimgs = imgs[base:base + _GROUP] if imgs is not None else None
            ys = _GroupSte.apply(tuple(meta), dead, *weights[base:base + _GROUP], *outs[base:base + _GROUP], *steps,
                                 *(gimgs if gimgs is not None else ()))
            ys = _dual_outputs(ys, gimgs, len(group), self._image_stat)
            for i, (u, w, y) in enumerate(zip(group, weights[base:base + _GROUP], ys)):
                pending = undo.get((id(u.layer), u.attr))
                if pending is None:
                    pending = _Pending(u, None if u in plain else u.q._quantized, None, w._version, False,
                                       plain=(False if u in plain else None))
                pending.dead, pending.index = dead, i
                self._pending.append(pending)
                u.layer.__dict__[_READY[u.attr]] = (y, pending, self)
