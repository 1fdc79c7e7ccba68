"""Multi-tensor weight path.

``quantize(conv)`` reads ``conv.weight`` through a ``QuantizeLayer`` (reference quantize.py:559-571 via imitation.py:61-68):
per layer and training step an abs-max, a running-scale update and a quantization -- three launches of 3-5 us each
plus ~100 us of Python.  A converted ResNet-50 has 54 such layers.  They depend on nothing but the parameters, so
``WeightBatcher`` evaluates all of them at the start of the root's forward pass with THREE launches in total
(``qs_multi_absmax``, ``qs_multi_scale_update``, ``qs_multi_quant_fwd``; same arithmetic, bit-identical results) and
hands every layer its quantized weight when its forward asks for it.  The STE backward stays per layer (gradients
become ready one layer at a time).

``convert`` installs it on the network it returns (``set_qsparse_options(batch_weights=False)`` or
``convert(..., batch_weights=False)`` opt out; ``WeightBatcher.install(model)`` does the same by hand, ``.remove()``
undoes it).  It is safe by construction for anything a forward pass may do:

  * The reference evaluates a layer's operator when -- and only when -- the layer's weight is read
    (imitation.py:61-68).  A layer whose precomputed weight was NOT consumed by the end of the root's forward -- a branch
    the forward skipped, an exception half-way -- is rolled back to exactly the state it had before: running scale (from
    the backup ``qs_multi_scale_update`` wrote), ``_n_updates``, the callback's ``t`` (host and device copy) and
    ``_quantized``.  The forward hook that does this also runs when the forward raised.  Such a layer's weight receives NO
    gradient from its place in the hand-out node (like the layer the reference never evaluated: ``.grad`` stays ``None``, an
    optimizer with weight decay does not touch it) -- except under an initialised process group, where
    ``DistributedDataParallel`` counts the parameter as used and waits for a gradient: there it contributes zeros, as DDP's
    own unused parameters do.
  * A weight that is read a second time in the same forward takes the inline path, as the second read of the reference.
  * A weight written between the precomputation and its read is rolled back and re-evaluated inline.  The write is seen
    through ``Tensor._version`` (every in-place operation); the one route that bypasses the counter is ``param.data``.  The
    place where user code runs between the two is a forward pre-hook of the layer itself: layers that carry one are not
    batched.  (A raw ``.data`` write to a layer's weight from OTHER code running inside the same forward, before that
    layer's read, is not visible -- switch ``batch_weights`` off for such a network.)
  * Layers whose quantizer carries hooks, layers on the CPU, pruned or bias-quantized layers and per-channel quantizers
    never take part; they keep their inline path.

Only layers whose weight is read through exactly one tensor-wise Scaler / Decimal ``QuantizeLayer`` (no weight pruning,
no bias quantizer, float32 parameter on the GPU) take part.  In evaluation mode the quantized weights are computed once
and handed out again until a parameter or a scale changes (serving: no weight-side launch at all per request).
"""
from typing import List, Optional

import torch
import torch.nn as nn

from torch.nn.modules import module as _m

from qsparse_amd import _hip
from qsparse_amd.quantize import DecimalQuantizer, QuantizeLayer, ScalerQuantizer
from qsparse_amd.util import get_option, logging

_ALIGN = 64   # elements between the starts of two outputs in the flat buffer (256 bytes)
_READY = "_qs_ready_weight"        # layer.__dict__ key of a precomputed weight waiting for its read
_ATTR = "_qs_weight_batcher"       # root.__dict__ key of the installed batcher


_GROUP = 8      # layers per hand-out node: their weight gradients are clamped together, by ONE launch, when the group's
                # earliest layer has finished its backward (small groups keep DDP's bucketed all-reduce overlapping)


class _GroupSte(torch.autograd.Function):
    """hands out the precomputed quantized weights of a GROUP of consecutive layers through one autograd node whose
    backward applies the quantizers' STE clamp (reference quantize.py:66-77, 120-131) to all of their gradients with one
    multi-tensor launch (qs_multi_ste_bwd) -- instead of one node, one Python backward and one 4 us launch per layer."""

    @staticmethod
    def forward(ctx, meta, dead, *tensors):
        k = len(meta)
        ctx.meta, ctx.dead = meta, dead                     # per layer: (is_decimal, lo_mul, hi_mul, passthrough); rolled back?
        ctx.shapes = [(w.shape, w.stride()) for w in tensors[:k]]
        ctx.save_for_backward(*tensors[2 * k:])             # the steps (scale or decimal, one element each)
        # a member whose weight is never read gets NO gradient, exactly like the layer the reference never evaluated
        # (zero_grad(set_to_none=True) + weight decay would otherwise start to move a parameter the forward did not use)
        ctx.set_materialize_grads(False)
        ys = tuple(y.view_as(y) for y in tensors[k:2 * k])
        # a frozen weight (requires_grad=False: fine-tuning a head on a frozen backbone) hands out a quantized weight that
        # does not require grad either, as layer by layer -- its convolution then skips the weight-gradient pass altogether
        frozen = [ys[i] for i in range(k) if not ctx.needs_input_grad[2 + i]]
        if frozen:
            ctx.mark_non_differentiable(*frozen)
        return ys

    @staticmethod
    def backward(ctx, *grads):
        meta, steps = ctx.meta, ctx.saved_tensors
        k = len(meta)
        out = [None] * k
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
            fast = [i for i in idx if grads[i].dtype == torch.float32 and grads[i].data_ptr() % 16 == 0
                    and (grads[i].is_contiguous() or (grads[i].dim() == 4 and grads[i].is_contiguous(memory_format=torch.channels_last)))]
            for i in idx:
                if i not in fast:                           # odd layouts / dtypes: the per-layer entry point
                    out[i] = _hip.ste_bwd(grads[i], steps[i], decimal, -1, meta[i][1], meta[i][2], False, torch.float32)
            if fast:
                gs = [grads[i] for i in fast]
                flat = torch.empty(sum((g.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for g in gs), dtype=torch.float32, device=gs[0].device)
                outs, off = [], 0
                for g in gs:
                    outs.append(flat.as_strided(g.shape, g.stride(), flat.storage_offset() + off))     # the gradient's own (dense) layout
                    off += (g.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                numels = [g.numel() for g in gs]
                _hip.multi_ste_bwd(len(fast), _hip.ptr_array(gs), _hip.ptr_array(outs), _hip.ptr_array([steps[i] for i in fast]),
                                   _hip.i64_array(numels), _hip.f32_array([meta[i][1] for i in fast]),
                                   _hip.f32_array([meta[i][2] for i in fast]), decimal, gs[0].device, nbytes=8 * sum(numels))
                for i, o in zip(fast, outs):
                    out[i] = o
        for i in range(k):
            if out[i] is None and grads[i] is not None and meta[i][3]:
                out[i] = grads[i]                           # backward_passthrough
        return (None, None) + tuple(out) + (None,) * (2 * k)


def _imitation_depth(layer: nn.Module) -> int:
    """number of operators the layer's weight is read through (imitation.py stacks one subclass per operator); the
    batcher's own hand-out subclass is not one of them"""
    return sum(1 for cls in type(layer).__mro__
               if isinstance(cls.__dict__.get("weight"), property) and "_qs_batcher_base" not in cls.__dict__)


def _eligible(layer: nn.Module) -> bool:
    q = getattr(layer, "quantize", None)
    w = layer._parameters.get("weight") if hasattr(layer, "_parameters") else None
    if not isinstance(q, QuantizeLayer) or w is None or _imitation_depth(layer) != 1:
        return False
    if isinstance(getattr(layer, "quantize_bias", None), QuantizeLayer) or hasattr(layer, "prune"):
        return False
    qc = q.callback
    return (type(qc) in (ScalerQuantizer, DecimalQuantizer) and qc.group_num <= 0 and q.channelwise < 0
            and q.batch_dimension == -1 and q.timeout > 0)


def _batchable(model: nn.Module) -> List[nn.Module]:
    """the eligible layers of `model` that own their quantizer callback.  `convert` gives every layer a fresh copy, but a
    hand-built network may hand ONE callback object to several layers (the reference allows it: the callback's running-mean
    count `t` then advances once per layer read, in forward order) -- such layers keep the inline path, whose order of
    evaluation is the forward's own."""
    layers = [m for m in model.modules() if _eligible(m)]
    owners = {}
    for m in model.modules():            # every QuantizeLayer of the tree counts, not only the eligible layers' own
        if isinstance(m, QuantizeLayer) and m.callback is not None:
            owners[id(m.callback)] = owners.get(id(m.callback), 0) + 1
    return [m for m in layers if owners.get(id(m.quantize.callback), 0) == 1]


def _hooked(q: QuantizeLayer) -> bool:
    """hooks on the quantizer or its callback (or global module hooks) must see the calls they were registered for: such a
    layer keeps its inline path"""
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in (q, q.callback))


class _LaunchPlan(dict):
    """cached pointer arrays of the three launches (ctypes objects): never copied or pickled with the network"""

    def __deepcopy__(self, memo):
        return _LaunchPlan()

    def __reduce__(self):
        return (_LaunchPlan, ())


class _Pending:
    """what has to be undone if a precomputed layer's weight is never read"""
    __slots__ = ("layer", "slot", "was_quantized", "t_dev", "version", "training", "dead", "index")

    def __init__(self, layer, slot, was_quantized, t_dev, version, training):
        self.layer, self.slot, self.was_quantized, self.t_dev, self.version, self.training = (layer, slot, was_quantized, t_dev,
                                                                                             version, training)
        self.dead, self.index = None, 0     # the hand-out group's "rolled back" flags and this layer's place in them


def _patched_class(base):
    """subclass of `base` whose `weight` hands out a waiting precomputed tensor once, else reads like `base`.  The
    property finds everything through the instance (nothing closes over a batcher), so a deep copy of the network keeps
    working on its own."""

    def read_weight(self_):
        entry = self_.__dict__.pop(_READY, None)
        if entry is not None:
            y, pending, batcher = entry
            w = self_._parameters["weight"]
            if w._version == pending.version:
                batcher._consumed(pending)
                return y
            batcher._rollback(pending)           # the parameter was written since the precomputation: evaluate inline
        return base.weight.__get__(self_)

    return type(base.__name__, (base,), {"weight": property(read_weight), "_qs_batcher_base": base})


class WeightBatcher:
    def __init__(self, model: nn.Module):
        for m in model.modules():                # one batcher per tree: an earlier one (convert installs one) steps aside
            old = m.__dict__.get(_ATTR)
            if old is not None:
                old.remove()
        self.model = model
        self.layers: List[nn.Module] = _batchable(model)
        self._pending: List[_Pending] = []
        self._amax = None
        self._decimals = None
        self._backup = None
        self._eval_key = None
        self._eval_outs = None
        self._eval_decimals = None
        self._plan = None
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
        self._eval_key = self._eval_outs = self._eval_decimals = None

    def remove(self):
        self._rollback_all()
        self._hook.remove()
        self._post.remove()
        for layer in self.layers:
            base = type(layer).__dict__.get("_qs_batcher_base")
            if base is not None:
                layer.__class__ = base
            layer.__dict__.pop(_READY, None)
        self.model.__dict__.pop(_ATTR, None)

    # ------------------------------------------------------------------------------------------
    def _consumed(self, pending: _Pending):
        try:
            self._pending.remove(pending)
        except ValueError:
            pass

    def _rollback(self, p: _Pending):
        """put the layer back where it was before `_precompute` advanced it (stream-ordered device writes, no sync)"""
        self._consumed(p)
        p.layer.__dict__.pop(_READY, None)
        if p.dead is not None:
            p.dead[p.index] = True               # its place in the group's autograd node delivers no gradient
        if not p.training:
            return                               # evaluation-mode hand-outs change no state
        q, qc = p.layer.quantize, p.layer.quantize.callback
        with torch.no_grad():
            q.weight.data.view(-1).copy_(self._backup[p.slot:p.slot + 1])
            q._steps.add(q._n_updates, -1)
            qc.t -= 1
            if p.t_dev is not None:
                p.t_dev.sub_(1)
                qc.__dict__["_t_dev_value"] = qc.t
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
        train, frozen = [], []          # layers that update statistics this step / that only quantize
        for layer in self.layers:
            q, w = layer.quantize, layer._parameters["weight"]
            dense = w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last))
            if not (w.is_cuda and w.dtype == torch.float32 and dense and w.data_ptr() % 16 == 0):
                continue                # (tensor-wise quantization does not care about the order of a dense tensor's elements)
            if "_qs_batcher_base" not in type(layer).__dict__ or _hooked(q) or layer._forward_pre_hooks:
                continue                # re-wrapped since (a further imitation); hooks that want to see the quantizer's calls;
                                        # a pre-hook on the layer (code that runs between this precomputation and the read)
            if not q.initted:
                continue                # first read ever: the inline path creates the layer's state when (and if) it happens
            if not (q.weight.is_cuda and q._n_updates.is_cuda):
                continue
            if w.is_inference() or q.weight.is_inference():
                continue                # created under torch.inference_mode(): no version counter to see a write with
            t = q._steps.read(q._n_updates)
            if t < q.timeout:
                continue                # identity phase: the inline path only counts
            if q.training:
                if t == q.timeout and get_option("log_during_train"):
                    logging.warn(f"quantizing {q.name} with {q.bits} bits")
                train.append(layer)
            elif q._quantized:
                frozen.append(layer)
        if not train and not frozen:
            return
        todo = train + frozen
        dev = todo[0]._parameters["weight"].device
        if any(l._parameters["weight"].device != dev for l in todo):
            return
        n_all = len(self.layers)
        if self._amax is None or self._amax.device != dev or self._amax.shape[0] != n_all:
            self._amax = torch.zeros(n_all, _hip.AMAX_LINE_STRIDE, dtype=torch.float32, device=dev)   # one line per layer
            self._decimals = torch.zeros(n_all, dtype=torch.float32, device=dev)
            self._backup = torch.zeros(n_all, dtype=torch.float32, device=dev)
        slot = {id(l): i for i, l in enumerate(self.layers)}
        weights = [l._parameters["weight"] for l in todo]
        # evaluation / serving: nothing changes between calls unless someone writes a parameter or a scale (both bump
        # `_version`), so the quantized weights of the previous call are handed out again without a launch
        eval_key = None
        # (only under no_grad / inference_mode: a loop that runs the network in eval() WITH gradients -- fine-tuning with
        # frozen statistics, manual SGD or EMA through `p.data.add_()` -- may write parameters by the one route the version
        # counter does not see, and would be handed stale weights silently; such loops pay the launch per forward)
        if not train and not torch.is_grad_enabled():
            eval_key = tuple((id(l), w.data_ptr(), w._version, l.quantize.weight.data_ptr(), l.quantize.weight._version,
                              tuple(w.stride()), l.quantize.callback.code_range(l.quantize.bits)) for l, w in zip(todo, weights))
            if eval_key == self._eval_key:
                self._hand_out(todo, weights, self._eval_outs, slot, {}, self._eval_decimals)
                return
        self._eval_key = None
        undo = {}
        with torch.no_grad():
            # everything that does not change from step to step -- the pointer arrays of the three launches, the layout of
            # the flat output buffer -- is built once per (set of layers, parameter storage) and reused
            sats = [l.quantize.callback.code_range(l.quantize.bits) for l in todo]
            key = (len(train), get_option("graph_safe"), tuple(sats)) + tuple((id(l), w.data_ptr(), w.numel(), l.quantize.weight.data_ptr(),
                                                                   l.quantize._n_updates.data_ptr()) for l, w in zip(todo, weights))
            plan = self._plan if self._plan is not None and self._plan.get("key") == key else None
            if plan is None:
                offsets, total = [], 0
                for w in weights:
                    offsets.append(total)
                    total += (w.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                plan = _LaunchPlan(key=key, offsets=offsets, total=total, numels=[w.numel() for w in weights], keep=[])
                if train:
                    amax = [self._amax[slot[id(l)]] for l in train]
                    decs = [self._decimals[slot[id(l)]:slot[id(l)] + 1] if not l.quantize.callback.use_float_scaler else None
                            for l in train]
                    backups = [self._backup[slot[id(l)]:slot[id(l)] + 1] for l in train]
                    plan["keep"] += amax + decs + backups
                    plan.update(w_ptrs=_hip.ptr_array(weights[:len(train)]), w_numels=_hip.i64_array(plan["numels"][:len(train)]),
                                amax_ptrs=_hip.ptr_array(amax), scale_ptrs=_hip.ptr_array([l.quantize.weight.data for l in train]),
                                dec_ptrs=_hip.ptr_array(decs), backup_ptrs=_hip.ptr_array(backups),
                                bits=(_hip.c_int * len(train))(*[l.quantize.bits for l in train]),
                                bump_ptrs=_hip.ptr_array([l.quantize._n_updates.data for l in train]),
                                w_bytes=4 * sum(plan["numels"][:len(train)]))
                groups = []
                for decimal in (False, True):
                    idx = [i for i, l in enumerate(todo) if (not l.quantize.callback.use_float_scaler) == decimal]
                    if idx:
                        params = [self._decimals[slot[id(todo[i])]:slot[id(todo[i])] + 1] if decimal else todo[i].quantize.weight.data
                                  for i in idx]
                        plan["keep"] += params
                        # opt-in saturation per layer: (lo, hi) with lo <= hi, or the empty range (1, 0) for "none"
                        los = _hip.i32_array([1 if sats[i] is None else sats[i][0] for i in idx]) if any(sats[i] for i in idx) else None
                        his = _hip.i32_array([0 if sats[i] is None else sats[i][1] for i in idx]) if los is not None else None
                        groups.append((decimal, idx, _hip.ptr_array([weights[i] for i in idx]), _hip.ptr_array(params),
                                       _hip.i64_array([plan["numels"][i] for i in idx]), 8 * sum(plan["numels"][i] for i in idx),
                                       los, his))
                plan["groups"] = groups
                self._plan = plan
            if train:
                graph_safe = get_option("graph_safe")
                t_devs = [l.quantize.callback.device_t(dev) if graph_safe else None for l in train]
                _hip.multi_absmax(len(train), plan["w_ptrs"], plan["w_numels"], plan["amax_ptrs"], dev, nbytes=plan["w_bytes"])
                _hip.multi_scale_update(len(train), plan["amax_ptrs"], plan["scale_ptrs"], plan["dec_ptrs"],
                                        _hip.i64_array([l.quantize.callback.t for l in train]),
                                        _hip.ptr_array(t_devs) if graph_safe else None, plan["bits"], plan["bump_ptrs"], dev,
                                        backup_ptrs=plan["backup_ptrs"])
                for l, t_dev in zip(train, t_devs):
                    q, qc = l.quantize, l.quantize.callback
                    undo[id(l)] = _Pending(l, slot[id(l)], q._quantized, t_dev, l._parameters["weight"]._version, True)
                    qc._advance_t(t_dev, bumped_by_kernel=True)
                    q._quantized = True
                    q._steps.note_device_add(q._n_updates, 1)
            if frozen:       # evaluation: the decimals of the frozen scales (the inline path recomputes them per call as well)
                for l in frozen:
                    qc = l.quantize.callback
                    if not qc.use_float_scaler:
                        self._decimals[slot[id(l)]:slot[id(l)] + 1] = _hip.decimal_from_scale(l.quantize.weight.data.view(-1))
            flat = torch.empty(plan["total"], dtype=torch.float32, device=dev)
            base = flat.data_ptr()
            so = flat.storage_offset()
            outs = [flat.as_strided(w.shape, w.stride(), so + o) for o, w in zip(plan["offsets"], weights)]   # w's own layout
            for decimal, idx, x_ptrs, param_ptrs, numels, nbytes, los, his in plan["groups"]:
                y_ptrs = (_hip.ctypes.c_void_p * len(idx))(*[base + 4 * plan["offsets"][i] for i in idx])
                _hip.multi_quant_fwd(len(idx), x_ptrs, y_ptrs, param_ptrs, numels, decimal, dev, nbytes=nbytes, code_lo=los,
                                     code_hi=his)
            # a DecimalQuantizer's backward clamps with the decimal of ITS forward (the reference computes a fresh tensor per
            # call, quantize.py:312-325, and the Function saves that one, :41): the hand-out nodes get this step's values, not
            # the buffer the next precomputation overwrites (a ScalerQuantizer's saves the scale parameter itself, :108)
            decimals = self._decimals.clone() if any(g[0] for g in plan["groups"]) else self._decimals
        if eval_key is not None:
            self._eval_key, self._eval_outs, self._eval_decimals = eval_key, outs, decimals
        self._hand_out(todo, weights, outs, slot, undo, decimals)

    def _hand_out(self, todo, weights, outs, slot, undo, decimals):
        """park every layer's quantized weight on the layer, `_GROUP` consecutive layers per autograd node (a node per layer
        in evaluation mode under no_grad costs nothing either way)"""
        for base in range(0, len(todo), _GROUP):
            group = todo[base:base + _GROUP]
            meta, steps = [], []
            for l in group:
                q, qc = l.quantize, l.quantize.callback
                is_decimal = not qc.use_float_scaler
                limit = 2.0 ** (q.bits - 1)
                notch = 1 if qc.flip_axis else 0
                meta.append((is_decimal, -limit + notch, limit - 1 + notch, bool(qc.backward_passthrough)))
                steps.append(decimals[slot[id(l)]:slot[id(l)] + 1].view(1, 1) if is_decimal else q.weight.data)
            dead = [False] * len(group)
            ys = _GroupSte.apply(tuple(meta), dead, *weights[base:base + _GROUP], *outs[base:base + _GROUP], *steps)
            for i, (l, w, y) in enumerate(zip(group, weights[base:base + _GROUP], ys)):
                pending = undo.get(id(l))
                if pending is None:
                    pending = _Pending(l, slot[id(l)], l.quantize._quantized, None, w._version, False)
                pending.dead, pending.index = dead, i
                self._pending.append(pending)
                l.__dict__[_READY] = (y, pending, self)
