"""Multi-tensor weight path.

``quantize(conv)`` reads ``conv.weight`` through a ``QuantizeLayer`` (reference quantize.py:559-571 via imitation.py):
per layer and training step an abs-max, a running-scale update and a quantization -- three launches of 3-5 us each
plus ~100 us of Python.  A converted ResNet-50 has 54 such layers.  They depend on nothing but the parameters, so
``WeightBatcher`` evaluates all of them at the start of the forward pass with THREE launches in total
(``qs_multi_absmax``, ``qs_multi_scale_update``, ``qs_multi_quant_fwd``; same arithmetic, bit-identical results) and
hands every layer its quantized weight when its forward asks for it.  The STE backward stays per layer (gradients
become ready one layer at a time).

    model = qs.convert(...).cuda()
    qs.WeightBatcher(model)        # once; .remove() undoes it

Only layers whose weight is read through exactly one tensor-wise Scaler / Decimal ``QuantizeLayer`` (no weight pruning,
no bias quantizer, float32 parameter on the GPU) take part; every other layer keeps its inline path.  In evaluation
mode the quantized weights are computed once and handed out again until a parameter or a scale changes (serving: no
weight-side launch at all per request).  A layer that the
forward pass never reaches has its statistics advanced all the same -- unlike the inline path; do not use the batcher
for networks that skip layers data-dependently.
"""
from typing import Dict, List

import torch
import torch.nn as nn

from qsparse_amd import _hip
from qsparse_amd.quantize import DecimalQuantizer, QuantizeLayer, ScalerQuantizer
from qsparse_amd.util import get_option, logging

_ALIGN = 64   # elements between the starts of two outputs in the flat buffer (256 bytes)


class _PrecomputedSte(torch.autograd.Function):
    """hands out a quantized weight computed by the batched kernels; backward = the quantizer's STE clamp."""

    @staticmethod
    def forward(ctx, w, y, step, is_decimal, bits, notch, passthrough):
        ctx.is_decimal, ctx.bits, ctx.notch, ctx.passthrough = is_decimal, bits, notch, passthrough
        ctx.save_for_backward(step)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        if ctx.passthrough:
            return g, None, None, None, None, None, None
        (step,) = ctx.saved_tensors
        limit = 2.0 ** (ctx.bits - 1)
        gx = _hip.ste_bwd(g, step, ctx.is_decimal, -1, -limit + ctx.notch, limit - 1 + ctx.notch, False, torch.float32)
        return gx, None, None, None, None, None, None


def _imitation_depth(layer: nn.Module) -> int:
    return sum(1 for cls in type(layer).__mro__ if isinstance(cls.__dict__.get("weight"), property))


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


class WeightBatcher:
    def __init__(self, model: nn.Module):
        self.model = model
        self.layers: List[nn.Module] = [m for m in model.modules() if _eligible(m)]
        if len({id(m.quantize.callback) for m in self.layers}) != len(self.layers):
            raise ValueError("WeightBatcher needs one quantizer callback per layer (convert() makes them so)")
        self._cache: Dict[int, torch.Tensor] = {}
        self._orig = {}
        self._amax = None
        self._decimals = None
        self._eval_key = None
        self._eval_outs = None
        for layer in self.layers:
            self._patch(layer)
        self._hook = model.register_forward_pre_hook(self._precompute)

    def _patch(self, layer: nn.Module):
        base = type(layer)
        self._orig[id(layer)] = base
        cache = self._cache

        def read_weight(self_):
            y = cache.pop(id(self_), None)
            return y if y is not None else base.weight.__get__(self_)

        layer.__class__ = type(base.__name__, (base,), {"weight": property(read_weight)})

    def invalidate(self):
        """forget the quantized weights kept for evaluation.  They are reused while no parameter and no scale has been
        written -- detected through ``Tensor._version``, which optimizers, ``load_state_dict`` and any in-place op bump;
        writes through ``param.data`` do not, call this after such a write."""
        self._eval_key = self._eval_outs = None

    def remove(self):
        self._hook.remove()
        for layer in self.layers:
            layer.__class__ = self._orig[id(layer)]
        self._cache.clear()

    # ------------------------------------------------------------------------------------------
    def _precompute(self, module, args):
        self._cache.clear()
        train, frozen = [], []          # layers that update statistics this step / that only quantize
        for layer in self.layers:
            q, w = layer.quantize, layer._parameters["weight"]
            dense = w.is_contiguous() or (w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last))
            if not (w.is_cuda and w.dtype == torch.float32 and dense and w.data_ptr() % 16 == 0):
                continue                # (tensor-wise quantization does not care about the order of a dense tensor's elements)
            if not q.initted:
                q._lazy_init(w)
            if not (q.weight.is_cuda and q._n_updates.is_cuda):
                continue
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
        slot = {id(l): i for i, l in enumerate(self.layers)}
        weights = [l._parameters["weight"] for l in todo]
        # evaluation / serving: nothing changes between calls unless someone writes a parameter or a scale (both bump
        # `_version`), so the quantized weights of the previous call are handed out again without a launch
        eval_key = None
        if not train:
            eval_key = tuple((id(l), w.data_ptr(), w._version, l.quantize.weight.data_ptr(), l.quantize.weight._version,
                              tuple(w.stride())) for l, w in zip(todo, weights))
            if eval_key == self._eval_key:
                for l, w, y in zip(todo, weights, self._eval_outs):
                    self._hand_out(l, w, y, slot)
                return
        self._eval_key = None
        with torch.no_grad():
            if train:
                graph_safe = get_option("graph_safe")
                tw = [l._parameters["weight"] for l in train]
                amax = [self._amax[slot[id(l)]] for l in train]
                scales = [l.quantize.weight.data for l in train]
                decs = [self._decimals[slot[id(l)]:slot[id(l)] + 1] if isinstance(l.quantize.callback, DecimalQuantizer)
                        and not l.quantize.callback.use_float_scaler else None for l in train]
                t_devs = [l.quantize.callback.device_t(dev) if graph_safe else None for l in train]
                _hip.multi_absmax(len(train), _hip.ptr_array(tw), _hip.i64_array([w.numel() for w in tw]), _hip.ptr_array(amax), dev)
                _hip.multi_scale_update(len(train), _hip.ptr_array(amax), _hip.ptr_array(scales), _hip.ptr_array(decs),
                                        _hip.i64_array([l.quantize.callback.t for l in train]), _hip.ptr_array(t_devs),
                                        (_hip.c_int * len(train))(*[l.quantize.bits for l in train]),
                                        _hip.ptr_array([l.quantize._n_updates.data for l in train]), dev)
                for l, t_dev in zip(train, t_devs):
                    q, qc = l.quantize, l.quantize.callback
                    qc._advance_t(t_dev, bumped_by_kernel=True)
                    q._quantized = True
                    q._steps.note_device_add(q._n_updates, 1)
            if frozen:       # evaluation: the decimals of the frozen scales (the inline path recomputes them per call as well)
                for l in frozen:
                    qc = l.quantize.callback
                    if not qc.use_float_scaler:
                        self._decimals[slot[id(l)]:slot[id(l)] + 1] = _hip.decimal_from_scale(l.quantize.weight.data.view(-1))
            offsets, total = [], 0
            for w in weights:
                offsets.append(total)
                total += (w.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            outs = [flat[o:o + w.numel()].as_strided(w.shape, w.stride()) for o, w in zip(offsets, weights)]   # w's own layout
            numels = [w.numel() for w in weights]
            for decimal in (False, True):
                idx = [i for i, l in enumerate(todo) if (not l.quantize.callback.use_float_scaler) == decimal]
                if not idx:
                    continue
                params = [self._decimals[slot[id(todo[i])]:slot[id(todo[i])] + 1] if decimal else todo[i].quantize.weight.data
                          for i in idx]
                _hip.multi_quant_fwd(len(idx), _hip.ptr_array([weights[i] for i in idx]), _hip.ptr_array([outs[i] for i in idx]),
                                     _hip.ptr_array(params), _hip.i64_array([numels[i] for i in idx]), decimal, dev)
        if eval_key is not None:
            self._eval_key, self._eval_outs = eval_key, outs
        for l, w, y in zip(todo, weights, outs):
            self._hand_out(l, w, y, slot)

    def _hand_out(self, l, w, y, slot):
        q, qc = l.quantize, l.quantize.callback
        is_decimal = not qc.use_float_scaler
        step = self._decimals[slot[id(l)]:slot[id(l)] + 1].view(1, 1) if is_decimal else q.weight.data
        self._cache[id(l)] = _PrecomputedSte.apply(w, y, step, is_decimal, q.bits, 1 if qc.flip_axis else 0,
                                                   bool(qc.backward_passthrough))
