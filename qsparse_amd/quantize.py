"""Quantization operators with straight-through gradients -- API of the reference's qsparse/quantize.py.

Same public names, arguments, state (``weight``, ``_n_updates``, optional ``callback.groups``) and
numerics as mlzxy/qsparse v2.0.1, but the tensor math runs as hand-written HIP kernels whenever the
tensor lives on the GPU (``libqsparse_hip.so``: qs_quant_*_fwd, qs_quant_ste_bwd, qs_absmax, qs_minmax,
qs_scale_update, qs_lines_update); CPU tensors take the equivalent ATen expression so that the host
logic can be exercised without a device.  GPU tensors never take the ATen route: if the library is
missing they raise ``QsparseHipError``.

Numerical contract (reference file:line in comments): forward never saturates (the clamp at
quantize.py:56-62/110-116 acts on a temporary), scaler rounds half-to-even after a true division,
decimal truncates, the backward clamps gradient *values* into ``[(-L+notch)*s, (L-1+notch)*s]``.
"""
import math
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from qsparse_amd import _hip
from qsparse_amd import distributed as qdist
from qsparse_amd.common import (HostMirror, TensorOrFloat, TensorOrInt, adopt_state_parameters, ensure_tensor,
                                state_parameter)
from qsparse_amd.imitation import imitate
from qsparse_amd.util import get_option, logging


# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------
def _on_channel(p, ndim: int, channel_index: int, length: int):
    """reshape a multi-element parameter so that it broadcasts along ``channel_index``."""
    if isinstance(p, torch.Tensor) and p.numel() > 1:
        assert len(p) == length, "channel of input and decimal must be equal in channel-wise quantization"
        view = [1] * ndim
        view[channel_index] = -1
        return p.view(*view)
    return p


def _quotient_dtype(x: torch.Tensor, param) -> torch.dtype:
    """dtype ATen gives ``x / param`` -- the precision the reference rounds the quotient to."""
    return torch.result_type(x, param)


def _int32(t: torch.Tensor) -> torch.Tensor:
    """``t.int()`` as the reference's CPU evaluates it: x86's cvttss2si / cvttsd2si return INT32_MIN for a NaN, an infinity and
    anything outside the int32 range, which the HIP kernels restate.  ATen's device cast saturates instead, so a
    GPU tensor that takes the ATen expression (float64, integers: `_hip.on_hip`) spells the rule out; on the CPU it is the cast."""
    if not t.is_cuda or not t.is_floating_point():
        return t.int()
    ok = (t > -2147483649.0) & (t < 2147483648.0)          # (a NaN compares false)
    return torch.where(ok, t, torch.full_like(t, -2147483648.0)).int()


def _pow2(d):
    """``2.0 ** d`` as the reference's CPU evaluates it (quantize.py:52-53: libm's pow, exact for an integer exponent).  ATen's
    device pow is not -- powf(2, 16) is 65535.996 on gfx950, one code off at x = -0.5 with 16 fractional bits -- so a GPU exponent
    tensor that takes the ATen expression (float64 / integer inputs, `_hip.on_hip`) gets its integer-valued entries as exact powers
    of two, assembled from the exponent bits; on the CPU, and for a Python number, it is the operator."""
    if not (isinstance(d, torch.Tensor) and d.is_cuda):
        return 2.0 ** d
    p = 2.0 ** d
    wide = p.dtype == torch.float64
    whole = d.round() if d.is_floating_point() else d
    lim, bias, shift, idt, fdt = (1022, 1023, 52, torch.int64, torch.float64) if wide else (126, 127, 23, torch.int32, torch.float32)
    exact = ((whole.clamp(-lim, lim).to(idt) + bias) << shift).view(fdt).to(p.dtype)
    return torch.where((d == whole) & (whole.abs() <= lim), exact, p)


def _out_dtype(x: torch.Tensor) -> torch.dtype:
    """float32 like the reference (type promotion), or the input dtype with the ``preserve_dtype`` extension"""
    if get_option("preserve_dtype") and x.dtype in (torch.bfloat16, torch.float16):
        return x.dtype
    return torch.float32


def code_range(bits: int, notch: int = 0, use_uint: bool = False, saturate=None):
    """``(code_lo, code_hi)`` of the opt-in forward saturation, or None when it is off (the reference's behaviour).
    ``saturate``: True / False, or None to follow ``set_qsparse_options(saturate=...)``.  The range is the one the
    reference's forward names in its lost clamp (quantize.py:56-62, 110-116): ``[0, 2^bits - 1]`` with ``use_uint``, else
    ``[-2^(bits-1) + notch, 2^(bits-1) - 1 + notch]`` (``notch = 1`` with ``flip_axis``)."""
    if saturate is None:
        saturate = get_option("saturate")
    if not saturate:
        return None
    limit = 2 ** (bits - 1)
    return (0, 2 * limit - 1) if use_uint else (-limit + notch, limit - 1 + notch)


def _gpu_dtype_guard(x: torch.Tensor, qd: torch.dtype):
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16) or qd not in (torch.float32, x.dtype):
        raise _hip.QsparseHipError(
            f"HIP quantizers support float32/bfloat16/float16 inputs with float32 parameters, got {x.dtype} -> {qd}")


def _reference_shape(y: torch.Tensor, x: torch.Tensor, param) -> torch.Tensor:
    """the reference broadcasts the input against the parameter tensor (``input / scaler``): a tensor-wise parameter of
    shape (1, 1) -- what QuantizeLayer holds -- turns a 1-d input such as a bias into a (1, C) output (which nn.Linear
    accepts and nn.Conv2d rejects, in the reference as here).  The kernels work on x's shape; restore that quirk."""
    if isinstance(param, torch.Tensor) and param.numel() == 1 and param.dim() > x.dim():
        return y.view(torch.broadcast_shapes(tuple(x.shape), tuple(param.shape)))
    return y


def _with_codes(ctx, y: torch.Tensor, codes, return_codes: bool):
    """forward result of the quantizer Functions: ``y``, or ``(y, codes)`` with the integer codes as an int32,
    non-differentiable second output (a NaN level index of the line quantizer becomes INT32_MIN, as on the GPU)"""
    if not return_codes:
        return y
    if codes.dtype != torch.int32:
        codes = torch.where(codes != codes, torch.full_like(codes, -2.0 ** 31), codes).to(torch.int32)
    ctx.mark_non_differentiable(codes)
    return y, codes


class _SteFunction(torch.autograd.Function):
    """shared backward of the scaler and decimal quantizers (reference quantize.py:66-77, 120-131)."""

    @staticmethod
    def _backward(ctx, grad_output, step_is_decimal: bool):
        if ctx.backward_passthrough:
            return (grad_output,) + (None,) * 8
        limit = 2.0 ** (ctx.bits - 1)
        lo_mul, hi_mul = -limit + ctx.notch, limit - 1 + ctx.notch
        (step,) = ctx.saved_tensors
        if _hip.on_hip(grad_output) and ctx.x_dtype in _hip.HIP_DTYPES:
            out_dtype = ctx.x_dtype if grad_output.dtype == torch.float32 else grad_output.dtype
            gx = _hip.ste_bwd(grad_output, step, step_is_decimal, ctx.channel_index, lo_mul, hi_mul, False, out_dtype)
            return (gx,) + (None,) * 8
        s = _pow2(-step) if step_is_decimal else step
        if s.numel() > 1:
            s = _on_channel(s, grad_output.dim(), ctx.channel_index, grad_output.shape[ctx.channel_index])
        # values are clamped; the reference's masked assignment `v[v != grad_output] = 0` compares v with ITSELF (clamp_ returned
        # grad_output), so it zeroes nothing but NaNs: a NaN gradient, and every element once a NaN scale made the bounds NaN
        out = torch.clamp(grad_output, lo_mul * s, hi_mul * s)
        out[out != out] = 0
        return (out,) + (None,) * 8


class ScalerQuantization(_SteFunction):
    """``q = int(round(x / s)); y = float(q) * s`` with STE backward (reference quantize.py:80-131)."""

    @staticmethod
    def forward(ctx, input: torch.Tensor, bits: int = 8, scaler: TensorOrFloat = 0.1, channel_index: int = 1,
                use_uint: bool = False, backward_passthrough: bool = False, flip_axis: bool = False,
                return_codes: bool = False, saturate=None):
        """``return_codes`` (extension): also return the int32 codes ``q`` -- on the GPU the kernel's own ``codes`` output
        of the same pass, never a second evaluation.  ``saturate`` (extension, see ``code_range``): clamp the codes to the
        bit width's range -- the clamp the reference's forward spells out and loses; the backward is unchanged"""
        ctx.backward_passthrough = backward_passthrough
        ctx.notch = 1 if flip_axis else 0
        ctx.bits, ctx.channel_index, ctx.x_dtype = bits, channel_index, input.dtype
        ctx.save_for_backward(ensure_tensor(scaler).detach())
        sat = code_range(bits, ctx.notch, use_uint, saturate)
        if _hip.on_hip(input):
            qd = _quotient_dtype(input, scaler)
            _gpu_dtype_guard(input, qd)
            if isinstance(scaler, torch.Tensor) and scaler.numel() > 1:
                assert len(scaler) == input.shape[channel_index], \
                    "channel of input and decimal must be equal in channel-wise quantization"
            y, codes = _hip.quant_fwd("scaler", input, scaler, channel_index, qd, out_dtype=_out_dtype(input),
                                      want_codes=return_codes, saturate=sat)
            return _with_codes(ctx, _reference_shape(y, input, scaler), codes, return_codes)
        s = _on_channel(scaler, input.dim(), channel_index, input.shape[channel_index])
        codes = _int32(torch.round(input / s))
        if sat is not None:      # (the reference's own line with the assignment it lacks; off by default: see module docstring)
            codes = codes.clamp(sat[0], sat[1])
        return _with_codes(ctx, (codes.float() * s).to(_out_dtype(input)), codes, return_codes)

    @staticmethod
    def backward(ctx, grad_output, grad_codes=None):
        return _SteFunction._backward(ctx, grad_output, step_is_decimal=False)


class DecimalQuantization(_SteFunction):
    """``q = int(x * 2^d); y = float(q) * 2^-d`` with STE backward (reference quantize.py:24-77)."""

    @staticmethod
    def forward(ctx, input: torch.Tensor, bits: int = 8, decimal: TensorOrInt = 5, channel_index: int = 1,
                use_uint: bool = False, backward_passthrough: bool = False, flip_axis: bool = False,
                return_codes: bool = False, saturate=None):
        ctx.backward_passthrough = backward_passthrough
        ctx.notch = 1 if flip_axis else 0
        ctx.bits, ctx.channel_index, ctx.x_dtype = bits, channel_index, input.dtype
        ctx.save_for_backward(ensure_tensor(decimal).detach().float())
        sat = code_range(bits, ctx.notch, use_uint, saturate)
        if _hip.on_hip(input):
            qd = _quotient_dtype(input, decimal.float() if isinstance(decimal, torch.Tensor) else 1.0)
            _gpu_dtype_guard(input, qd)
            if isinstance(decimal, torch.Tensor) and decimal.numel() > 1:
                assert len(decimal) == input.shape[channel_index], \
                    "channel of input and decimal must be equal in channel-wise quantization"
            y, codes = _hip.quant_fwd("decimal", input, decimal, channel_index, qd, out_dtype=_out_dtype(input),
                                      want_codes=return_codes, saturate=sat)
            return _with_codes(ctx, _reference_shape(y, input, decimal), codes, return_codes)
        to_int = _on_channel(_pow2(decimal), input.dim(), channel_index, input.shape[channel_index])
        to_float = _on_channel(_pow2(-decimal), input.dim(), channel_index, input.shape[channel_index])
        codes = _int32(input * to_int)
        if sat is not None:
            codes = codes.clamp(sat[0], sat[1])
        return _with_codes(ctx, (codes.float() * to_float).to(_out_dtype(input)), codes, return_codes)

    @staticmethod
    def backward(ctx, grad_output, grad_codes=None):
        return _SteFunction._backward(ctx, grad_output, step_is_decimal=True)


class LineQuantization(torch.autograd.Function):
    """asymmetric uniform quantization between per-channel ``(start, end)`` lines, identity backward
    (reference quantize.py:134-185)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, bits: int = 8, lines=(-0.1, 0.9), channel_index=-1, inplace=False,
                float_zero_point=True, return_codes=False):
        """``return_codes`` (extension): also return the int32 level index in [0, 2^bits - 1] of every element"""
        if not isinstance(lines, torch.Tensor):
            lines = torch.tensor(lines).view(-1, 2).to(x.device)
        if channel_index >= 0:
            assert x.shape[channel_index] == lines.shape[0]
        assert lines.shape[1] == 2
        if _hip.on_hip(x):
            if x.dtype not in (torch.float32, torch.bfloat16, torch.float16) or lines.dtype != torch.float32:
                raise _hip.QsparseHipError(f"HIP line quantizer: unsupported dtypes {x.dtype} / {lines.dtype}")
            if return_codes:
                return _with_codes(ctx, *_hip.quant_line_fwd(x, lines, bits, channel_index, float_zero_point, want_codes=True), True)
            return _hip.quant_line_fwd(x, lines, bits, channel_index, float_zero_point)
        levels = 2 ** bits
        view = [1] * x.dim()
        if channel_index >= 0:
            view[channel_index] = -1
        lo, hi = lines[:, 0].view(view), lines[:, 1].view(view)
        xc = torch.clamp(x, lo, hi)
        step = (hi - lo) / levels
        step = torch.where(step == 0, torch.full_like(step, 0.0001), step)
        if float_zero_point:   # training form (:168-181): separate multiply and add
            idx = ((xc - lo) / step).round().clamp(0, levels - 1)
            return _with_codes(ctx, idx * step + lo, idx, return_codes)
        zero_point = (lo / step).round()   # evaluation form (:161-166): integer zero point
        idx = ((xc / step).round() - zero_point).clamp(0, levels - 1)
        return _with_codes(ctx, (idx + zero_point) * step, idx, return_codes)

    @staticmethod
    def backward(ctx, grad_output, grad_codes=None):
        return (grad_output,) + (None,) * 6


def quantize_with_decimal(input: torch.Tensor, bits: int = 8, decimal: TensorOrInt = 5, channel_index: int = -1,
                          use_uint: bool = False, backward_passthrough: bool = False,
                          flip_axis: bool = False, return_codes: bool = False, saturate=None) -> torch.Tensor:
    """power-of-two uniform quantization (reference quantize.py:188-208).

    Args mirror the reference: ``decimal`` is the number of fractional bits (int, or per-channel
    tensor viewed along ``channel_index``); ``use_uint`` is accepted and, as in the reference, has no
    effect; ``backward_passthrough`` skips the gradient clamp; ``flip_axis`` shifts the clamp
    interval by one step.  ``return_codes`` (extension of this package): return ``(y, q)`` with the int32 codes
    ``q = int(x * 2^d)`` the output was built from (``y == q.float() * 2^-d``) -- the integers an int8/int32 inference
    engine consumes (reference tests/test_quantize.py:73-101).  ``saturate`` (extension): clamp the codes to the bit
    width's range (``code_range``); with it ``use_uint`` selects the unsigned range the reference names."""
    return DecimalQuantization.apply(input, bits, decimal, channel_index, use_uint, backward_passthrough, flip_axis,
                                     return_codes, saturate)


def quantize_with_scaler(input: torch.Tensor, bits: int = 8, scaler: TensorOrFloat = 0.1, channel_index: int = -1,
                         use_uint: bool = False, backward_passthrough: bool = False,
                         flip_axis: bool = False, return_codes: bool = False, saturate=None) -> torch.Tensor:
    """scaling-factor based uniform quantization (reference quantize.py:210-230).  ``return_codes`` (extension):
    return ``(y, q)`` with the int32 codes ``q = int(round(x / s))``, ``y == q.float() * s``.  ``saturate`` (extension):
    ``q = clamp(q, code_range(bits, ...))`` -- what quantize.py:110-116 spells out -- default: the ``saturate`` option (off)."""
    return ScalerQuantization.apply(input, bits, scaler, channel_index, use_uint, backward_passthrough, flip_axis,
                                    return_codes, saturate)


def quantize_with_line(x: torch.Tensor, bits: int = 8,
                       lines: Union[Tuple[float, float], List[Tuple[float, float]]] = (-0.1, 0.9),
                       channel_index: int = -1, inplace: bool = False, float_zero_point: bool = True,
                       return_codes: bool = False) -> torch.Tensor:
    """asymmetric uniform quantization (reference quantize.py:232-255).  ``return_codes`` (extension): return
    ``(y, idx)`` with the int32 level index ``idx`` in ``[0, 2^bits - 1]``: ``y == idx * step + start`` in the training
    form, ``y == (idx + round(start / step)) * step`` with ``float_zero_point=False``."""
    return LineQuantization.apply(x, bits, lines, channel_index, inplace, float_zero_point, return_codes)


# ----------------------------------------------------------------------------------------------
# quantizer callbacks (policy + running statistics)
# ----------------------------------------------------------------------------------------------
class BaseQuantizer(nn.Module):
    """callback protocol of ``quantize`` (reference quantize.py:258-272)."""

    weight_size = 1

    def optimize(self, tensor, bits, weight=None, batched=False, channel_index=-1) -> torch.Tensor:
        """return the updated weight for this step"""
        raise NotImplementedError

    def forward(self, tensor, bits, weight=None, batched=False, channel_index=-1) -> torch.Tensor:
        """return the quantized tensor"""
        raise NotImplementedError

    def get_weight_shape(self, x, channelwise):
        return (1 if channelwise < 0 else x.shape[channelwise], self.weight_size)


def _absmax_rows_cpu(x: torch.Tensor, channel_index: int) -> torch.Tensor:
    a = x.abs()
    if channel_index == -1:
        rows = a.reshape(1, -1)
    elif channel_index == 0:
        rows = a.reshape(a.shape[0], -1)
    else:
        rows = a.transpose(0, channel_index).contiguous().view(a.shape[channel_index], -1)
    return rows.max(dim=1).values


class DecimalQuantizer(BaseQuantizer):
    """abs-max statistics + power-of-two scales (algorithm 3 of the MDPI paper; reference
    quantize.py:275-367).  ``optimize`` updates the running scale, ``forward`` quantizes."""

    weight_size = 1

    def __init__(self, use_uint: bool = False, backward_passthrough: bool = False, flip_axis: bool = False,
                 group_num=-1, group_timeout=512, saturate=None):
        """``saturate`` (extension; Scaler / Decimal): clamp the integer codes to the bit width's range, see ``code_range``;
        None follows ``set_qsparse_options(saturate=...)`` (default off: the reference never saturates, quirk B1)"""
        super().__init__()
        self.saturate = saturate
        self.use_uint = use_uint
        self.backward_passthrough = backward_passthrough
        self.flip_axis = flip_axis
        self.use_float_scaler = False
        self.function = DecimalQuantization.apply
        self.t = 0
        self.group_timeout = group_timeout
        self.groups = None
        self.group_num = group_num

    def device_t(self, device) -> torch.Tensor:
        """device-resident copy of ``self.t`` (int64, one element) for graph-safe kernels; recreated whenever the
        host value was changed by anyone but the step itself"""
        buf = self.__dict__.get("_t_dev")
        if buf is None or buf.device != device or self.__dict__.get("_t_dev_value") != self.t:
            buf = torch.full((1,), int(self.t), dtype=torch.int64, device=device)
            self.__dict__["_t_dev"] = buf
            self.__dict__["_t_dev_value"] = self.t
        return buf

    def tensor_accumulator(self, device) -> torch.Tensor:
        """the zeroed [16, 32] accumulator of a tensor-wise abs-max on `device` (see _hip.absmax), created once"""
        bufs = self.__dict__.setdefault("_absmax_bufs", {})
        key = (1, True, device)
        buf = bufs.get(key)
        if buf is None:
            buf = bufs[key] = _hip.tensor_amax_accumulator(device)
        return buf

    def _advance_t(self, t_dev: torch.Tensor = None, bumped_by_kernel: bool = False):
        self.t += 1
        if t_dev is not None:
            if not bumped_by_kernel:
                t_dev.add_(1)
            self.__dict__["_t_dev_value"] = self.t

    def quantize(self, tensor, bits, scaler, channel_index=-1, **kwargs):
        if self.use_float_scaler:
            param = scaler
        elif scaler.is_cuda:
            param = _hip.decimal_from_scale(scaler).view(scaler.shape)
        else:
            param = (1 / scaler).nan_to_num(posinf=1, neginf=1).log2().round()   # quantize.py:316
        args = (tensor, bits, param, channel_index, self.use_uint, self.backward_passthrough, self.flip_axis)
        sat = self.__dict__.get("saturate")
        if sat is not None or get_option("saturate"):     # (a user-supplied `function` keeps the reference's seven arguments)
            args += (False, sat)
        return self.function(*args)

    def code_range(self, bits: int):
        """the (lo, hi) this quantizer saturates its codes to, or None (off)"""
        return code_range(bits, 1 if self.flip_axis else 0, self.use_uint, self.__dict__.get("saturate"))

    def export(self, tensor, bits, scaler, channel_index=-1) -> dict:
        """integer form of ``self(tensor, bits, scaler, channel_index)`` (extension, see qsparse_amd/export.py): the
        codes are the second output of the very call ``forward`` makes -- group-wise scales included."""
        with torch.no_grad():
            if self.t >= self.group_timeout and self.group_num > 0 and scaler.numel() > self.group_num:
                scaler = self._group_scales(scaler)
            if self.use_float_scaler:
                y, codes = self.function(tensor, bits, scaler, channel_index, self.use_uint, self.backward_passthrough,
                                         self.flip_axis, True, self.__dict__.get("saturate"))
                return dict(kind="scaler", codes=codes, values=y, scale=scaler.detach().clone())
            decimal = (_hip.decimal_from_scale(scaler).view(scaler.shape) if scaler.is_cuda
                       else (1 / scaler).nan_to_num(posinf=1, neginf=1).log2().round())
            y, codes = self.function(tensor, bits, decimal, channel_index, self.use_uint, self.backward_passthrough,
                                     self.flip_axis, True, self.__dict__.get("saturate"))
            return dict(kind="decimal", codes=codes, values=y, decimal=decimal.to(torch.int32))

    def optimize(self, x, bits, weight=None, batched=False, channel_index=-1, **kwargs):
        """running mean of ``max|x| / 2^(bits-1)`` (reference quantize.py:327-349)."""
        with torch.no_grad():
            wshape = self.get_weight_shape(x, channel_index)
            if batched and channel_index >= 0 and x.shape[0] != 1:
                # the reference computes the per-channel maximum over (batch, spatial) and then fails in
                # `.view(-1, batch_size)` (quantize.py:341-343); keep the failure, with a clearer message
                raise RuntimeError(
                    "channel-wise Scaler/Decimal quantization of a batched activation is not supported by the "
                    f"reference (shape {tuple(x.shape)}, channelwise={channel_index}); use channelwise=-1 or "
                    "AdaptiveQuantizer")
            if _hip.on_hip(x):
                # three launches per step: abs-max accumulated into a persistent zeroed buffer, running-mean update
                # (which also clears that buffer and bumps the layer's step counter), quantization
                n_stat = wshape[0]
                bufs = self.__dict__.setdefault("_absmax_bufs", {})
                # (a callback shared by a tensor-wise layer and a channel-wise layer over ONE channel has n_stat == 1 twice
                # with different accumulator layouts: the layout is part of the key)
                key = (n_stat, channel_index < 0, x.device)
                buf = bufs.get(key)
                if buf is None:     # tensor-wise: 16 partial accumulators on lines of their own (see _hip.absmax)
                    buf = bufs[key] = (_hip.tensor_amax_accumulator(x.device) if channel_index < 0 else
                                       torch.zeros(n_stat, dtype=torch.float32, device=x.device))
                stat = _hip.absmax(x, channel_index, accumulate_into=buf, pre_relu=kwargs.get("pre_relu", False))
                if batched:      # activations differ per rank; weights and biases (batched=False) are identical under DDP
                    stat = qdist.allreduce_max_(stat)
                if weight is None:
                    weight = torch.zeros(wshape, device=x.device)
                t_dev = self.device_t(x.device) if get_option("graph_safe") else None
                counter = kwargs.get("step_counter")
                bump = counter.data if (counter is not None and counter.is_cuda and counter.device == x.device) else None
                _hip.scale_update(stat, weight.data, self.t, bits, t_dev=t_dev, clear_absmax=True, bump=bump, stat_dtype=x.dtype,
                                  advance_t_dev=True, lines=_hip.TENSOR_AMAX_LINES if channel_index < 0 else 1)
                self.__dict__["_bumped_step_counter"] = bump is not None
                self._advance_t(t_dev, bumped_by_kernel=True)
                return weight
            else:
                _hip.refuse_capture(x, "the scale statistics")
                stat = _absmax_rows_cpu(x, channel_index)
                if batched:
                    stat = qdist.allreduce_max_(stat)
                new_weight = (stat / (2 ** (bits - 1))).view(wshape)
                if self.t == 0:
                    weight = new_weight
                else:
                    weight.data[:] = _hip.true_div(self.t * weight + new_weight, self.t + 1)
        self.t += 1
        return weight

    def _group_scales(self, scaler):
        """optional group-wise quantization: cluster channels once with sklearn (host), then share the
        mean scale inside each group (reference quantize.py:352-366)."""
        if self.groups is None:
            from sklearn.cluster import AgglomerativeClustering

            logging.danger(f"clustering {len(scaler)} channels into {self.group_num} groups")
            labels = AgglomerativeClustering(n_clusters=self.group_num).fit(scaler.detach().cpu().numpy()).labels_
            self.groups = nn.Parameter(torch.from_numpy(labels).to(scaler.device), requires_grad=False)
        # the group means on the host copy of the (C-sized) scales: the reference's own ATen CPU arithmetic, bit for bit -- the
        # device's mean kernel sums in another order and multiplies by 1/k -- and one round trip instead of a boolean-index sync
        # per group
        shared = scaler.detach().cpu().clone() if scaler.is_cuda else torch.clone(scaler)
        groups = self.groups.detach().cpu() if self.groups.is_cuda else self.groups
        for gi in range(self.group_num):
            member = groups == gi
            shared[member] = shared[member].mean(dim=0)
        return shared.to(scaler.device)

    def forward(self, tensor, bits, scaler, channel_index=-1, **kwargs):
        if self.t >= self.group_timeout and self.group_num > 0 and scaler.numel() > self.group_num:
            scaler = self._group_scales(scaler)
        return self.quantize(tensor, bits, scaler, channel_index, **kwargs)


class ScalerQuantizer(DecimalQuantizer):
    """abs-max statistics with free (non power-of-two) scales (reference quantize.py:370-378)."""

    weight_size = 1

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_float_scaler = True
        self.function = ScalerQuantization.apply


class AdaptiveQuantizer(DecimalQuantizer):
    """min/max statistics + asymmetric lines (algorithm 2 of the MDPI paper; reference
    quantize.py:381-430)."""

    weight_size = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.function = LineQuantization.apply

    def quantize(self, tensor, bits, lines, channel_index=-1, **kwargs):
        return self.function(tensor, bits, lines, channel_index, kwargs.get("inplace", False), self.training)

    def export(self, tensor, bits, lines, channel_index=-1) -> dict:
        """integer form of the EVALUATION-mode output (integer zero point, reference quantize.py:161-166):
        ``values == (codes + zero_point) * step``"""
        with torch.no_grad():
            y, codes = self.function(tensor, bits, lines, channel_index, False, False, True)
            step = (lines[:, 1] - lines[:, 0]) / 2 ** bits
            step = torch.where(step == 0, torch.full_like(step, 0.0001), step)
            return dict(kind="line", codes=codes, values=y, lines=lines.detach().clone(), step=step,
                        zero_point=(lines[:, 0] / step).round().to(torch.int32))

    @staticmethod
    def _bounds_cpu(x, channel_index, batched):
        batch = x.shape[0]
        if channel_index >= 0:
            if batched:
                xt = x if channel_index == 1 else x.transpose(1, channel_index)
                rows = xt.reshape(-1, math.prod(xt.shape[2:]))
            else:
                xt = x if channel_index == 0 else x.transpose(0, channel_index)
                rows = xt.contiguous().view(-1, math.prod(xt.shape[1:]))
        else:
            rows = x.reshape(len(x) if batched else 1, -1)
        lo, hi = rows.min(dim=1).values, rows.max(dim=1).values
        if batched:   # min of per-sample minima, max of per-sample maxima (:415-418)
            lo, hi = lo.view(batch, -1).min(dim=0).values, hi.view(batch, -1).max(dim=0).values
        return torch.stack([lo, hi], dim=1).view(-1, 2)

    def optimize(self, x, bits, weight=None, channel_index=-1, batched=False, **kwargs):
        with torch.no_grad():
            if batched and channel_index >= 0 and channel_index != 1:
                # reference quantize.py:398-401: the channel dim is transposed next to the batch dim and the result is `.view`ed as
                # rows -- which raises for every batch of more than one sample (an ARGUMENT error: it fails for the contiguous tensor
                # too, like the batched channel-wise Scaler of :341-343; layouts the reference cannot `.view`, e.g. channels_last,
                # are a different matter and are served).  The same call on a meta tensor of x's shape raises the same error.
                probe = torch.empty(tuple(x.shape), device="meta", dtype=x.dtype).transpose(1, channel_index)
                probe.view(-1, math.prod(tuple(probe.shape)[2:]))
            if _hip.on_hip(x):
                # nothing to exchange between the reduction and the update (one process, or a weight): ONE reduction launch
                # into persistent key buffers, which the running-mean launch converts and resets -- two launches instead of
                # four (key initialisation, reduction, key -> float, running mean)
                keyed = weight is not None and not (batched and qdist.exchange_active())
                if keyed:
                    n_stat = weight.shape[0]
                    bufs = self.__dict__.setdefault("_minmax_keys", {})
                    keys = bufs.get((n_stat, x.device))
                    if keys is None:
                        keys = bufs[(n_stat, x.device)] = _hip.minmax_key_buffers(n_stat, x.device)
                lo, hi = _hip.minmax(x, channel_index, accumulate_into=keys if keyed else None)
                if batched and not keyed:
                    qdist.allreduce_min_(lo), qdist.allreduce_max_(hi)
                if weight is None:
                    self.t += 1
                    return torch.stack([lo, hi], dim=1)
                assert weight.shape == (lo.numel(), 2)
                t_dev = self.device_t(x.device) if get_option("graph_safe") else None
                _hip.lines_update(lo, hi, weight.data, self.t + 1, t_dev=t_dev, advance_t_dev=True, from_keys=keyed)
                self._advance_t(t_dev, bumped_by_kernel=True)
                return weight
            _hip.refuse_capture(x, "the line statistics")
            bounds = self._bounds_cpu(x, channel_index, batched)
            if batched and qdist.exchange_active():
                bounds = torch.stack([qdist.allreduce_min_(bounds[:, 0].contiguous()),
                                      qdist.allreduce_max_(bounds[:, 1].contiguous())], dim=1)
            self.t += 1
            if weight is None:
                return bounds
            assert bounds.shape == weight.shape
            return _hip.true_div(weight * (self.t - 1) + bounds, self.t)


# ----------------------------------------------------------------------------------------------
# the layer
# ----------------------------------------------------------------------------------------------
class _QuantStep(torch.autograd.Function):
    """a lone tensor-wise ScalerQuantizer step through ONE call into the library (qs_quantize_step: abs-max, running scale,
    quantization -- or the quantization alone); backward = the STE clamp, with the gate of a folded ReLU when `pre_relu`"""

    @staticmethod
    def forward(ctx, x, scale, amax, bits, t, t_dev, n_updates, pre_relu, update, notch, out_dtype, saturate=None, image_dtype=None,
                add_cell=None, act_out=None):
        # add_cell: of the promoting add that produced x (fused.grad_image_cell), or None
        # act_out: the caller applied an nn.GELU under no_grad (fused.act_rider) -- act_out = gelu(x) is what the quantizer reads, x (the
        # differentiable argument) is the GELU's input: kept, and the backward kernel multiplies by the GELU's derivative there
        ctx.add_cell = add_cell
        act_in = None
        if act_out is not None:
            act_in, x = x, act_out
        want_gate = bool(pre_relu and ctx.needs_input_grad[0] and get_option("relu_gate"))
        y = torch.empty_like(x, dtype=out_dtype)
        # an owned nn.ReLU(inplace=True) in front of this quantizer: the apply kernel writes relu(x) back into x itself
        cell = _hip.owned_relu_cell() if pre_relu else None
        widen = bool(pre_relu and out_dtype == torch.float32 and _hip.load().qs_quant_image_ok(1, 1, x.numel(), 0, 0, 1, _hip.dt(x)))
        xback = bool(cell is not None and widen)
        # the autocast image (fused.py): RNE(y) in the autocast dtype from the same pass, for the convolution behind this site --
        # when its gradient can come back the same way (a recorded gate) or no gradient is needed at all
        make_image = bool(image_dtype is not None and widen and (want_gate or not ctx.needs_input_grad[0]))
        gate_bits = torch.empty((x.numel() + 7) // 8, dtype=torch.uint8, device=x.device) if (want_gate or xback or make_image) else None
        if want_gate:
            _hip.note_gate(gate_bits)
        img = torch.empty_like(x, dtype=image_dtype) if make_image else None
        _hip.quantize_step(x, y, gate_bits, amax, scale, bits, t, t_dev, n_updates, pre_relu, update, saturate, xback=xback, image=img)
        if xback:
            cell["done"] = True
        ctx.bits, ctx.notch, ctx.pre_relu, ctx.has_gate = bits, notch, pre_relu, want_gate
        ctx.x_shape, ctx.x_dtype = x.shape, x.dtype
        ctx.channels_last = x.dim() in (4, 5) and not x.is_contiguous()
        ctx.has_act_x = act_in is not None
        ctx.save_for_backward(scale, gate_bits if want_gate else (x if pre_relu else x.new_empty(0)),
                              act_in if act_in is not None else x.new_empty(0))
        if make_image:
            ctx.set_materialize_grads(False)
            return y, img, img.detach()      # (twice: a gradient slot of its own for a second autocast consumer, fused.py "Second image")
        return y

    @staticmethod
    def backward(ctx, g, g16=None, g16b=None):
        n_in = 15
        override = ctx.__dict__.pop("_qs_override", None)
        if override is not None:         # a late hook on the output replaced its whole gradient (fused._late_hook)
            g, g16, g16b = override[0], None, None
        if g16 is None and g16b is not None:
            g16, g16b = g16b, None
        if g is None and g16 is None:
            return (None,) * n_in
        scale, second, act_x = ctx.saved_tensors
        limit = 2.0 ** (ctx.bits - 1)
        lo_mul, hi_mul = -limit + ctx.notch, limit - 1 + ctx.notch
        if ctx.has_act_x:
            # the caller's nn.GELU: one pass -- STE clamp of (g [+ g16]), rounded to x's dtype, times the GELU's derivative at act_x
            # (qs_ste_relu_bwd_args::act_x) -- when every operand lies in memory like act_x; else the ordinary routes below, then
            # ATen's gelu_backward as autograd's GeluBackward0 would run it.  (The site's own activation is none or the folded
            # identity: its gate changes nothing.)
            from qsparse_amd import fused
            if g16b is not None:
                g, g16b = (g16b.float() if g is None else g + g16b.float()), None

            def like_x(t_):
                return t_ is None or (t_.shape == act_x.shape and t_.stride() == act_x.stride() and t_.data_ptr() % 16 == 0)

            if (like_x(g) and like_x(g16) and (g is None or g.dtype in (torch.float32, act_x.dtype)) and (g16 is None or g is None or g.dtype == torch.float32)
                    and _hip.dense_any_order(act_x) and not _hip.logging_events()):
                n = act_x.numel()

                def flat(t_):
                    return None if t_ is None else t_.as_strided((n,), (1,))

                out = _hip.ste_act_bwd(flat(g), flat(act_x), scale, False, lo_mul, hi_mul, g2=flat(g16))
                fused.ROUTES["act_backward"] += 1
                return (out.as_strided(act_x.shape, act_x.stride()),) + (None,) * (n_in - 1)
            if g16 is not None:
                g, g16 = fused._whole(g, g16, None), None
        if ctx.pre_relu:
            gate = _hip.ReluGate.from_saved(second, ctx.x_shape, ctx.x_dtype, ctx.channels_last) if ctx.has_gate else None
            from qsparse_amd import fused
            if g16 is not None and gate is None:          # (cannot happen: an image is only made with a gate; stay correct anyway)
                g, g16, g16b = fused._whole(g, g16, g16b), None, None
            # the riders of the all-float32 kernel form: the second consumer's 2-byte share as a third stream, the image of gx for
            # the promoting add that produced this site's input (fused.py)
            f32_gated = gate is not None and ctx.x_dtype == torch.float32 and (g is None or g.dtype == torch.float32)
            if g16b is not None and not (f32_gated and g16 is not None and g16b.dtype == g16.dtype and g16b.shape == g16.shape):
                g, g16b = (g16b.float() if g is None else g + g16b.float()), None
            cell = ctx.add_cell if (f32_gated and not _hip.logging_events() and (g is None or g.dtype == torch.float32)) else None
            kw = {}
            if g16 is not None:
                kw["g2"] = g16
            if g16b is not None:
                kw["g3"] = g16b
            if cell is not None:
                kw["gx_image_dtype"] = cell["dtype"]
            gx = _hip.ste_relu_bwd(g, None if gate is not None else second, scale, False, lo_mul, hi_mul, None, gate=gate,
                                   act=ctx.pre_relu, **kw)
            if cell is not None:
                gx, gimg = gx
                cell["gx"], cell["g16"] = gx, gimg
                if _hip.image_byte_delta is not None:
                    _hip.image_byte_delta["apply_bwd"] += gimg.numel() * gimg.element_size()
        else:
            out_dtype = ctx.x_dtype if g.dtype == torch.float32 else g.dtype
            gx = _hip.ste_bwd(g, scale, False, -1, lo_mul, hi_mul, False, out_dtype)
        if ctx.has_act_x:
            gx = torch.ops.aten.gelu_backward(gx if gx.dtype == act_x.dtype else gx.to(act_x.dtype), act_x)
        return (gx,) + (None,) * (n_in - 1)


def _callback_hooked(cb: nn.Module) -> bool:
    """hooks that must see the callback's own calls (`callback.optimize` is a plain method, `callback(...)` a module call):
    forward / forward-pre / backward hooks on the callback, or hooks installed globally for every module -- the composite
    route never calls the callback, so such a layer keeps the protocol route (same predicate as fused._hooked, batch._hooked)"""
    from torch.nn.modules import module as _m
    if (_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks
            or _m._global_backward_pre_hooks):
        return True
    return bool(cb._forward_hooks or cb._forward_pre_hooks or cb._backward_hooks or cb._backward_pre_hooks)


def _dense(x: torch.Tensor) -> bool:
    if x.is_contiguous():
        return True
    return x.dim() in (4, 5) and x.is_contiguous(memory_format=torch.channels_last if x.dim() == 4 else torch.channels_last_3d)


class QuantizeLayer(nn.Module):
    """stateful quantization operator (reference quantize.py:434-518): identity for the first
    ``timeout`` training steps, then statistics update (training) + quantization."""

    def __str__(self):
        return (f"QuantizeLayer(bits={self.bits}, timeout={self.timeout}, "
                f"callback={self.callback.__class__.__name__}, channelwise={self.channelwise})")

    __repr__ = __str__

    def __init__(self, bits: int = 8, channelwise: int = 1, timeout: int = 1000, callback: BaseQuantizer = None,
                 batch_dimension: int = 0, name: str = ""):
        super().__init__()
        if get_option("log_on_created"):
            logging.info(f"[Quantize{name if name == '' else f' @ {name}'}] bits={bits} channelwise={channelwise} "
                         f"timeout={timeout}")
        self.name = name
        self.channelwise = channelwise
        self.timeout = timeout
        self.bits = bits
        self.callback = callback
        self.batch_dimension = batch_dimension   # 0: activation, -1: weight/bias
        self._quantized = False
        self._steps = HostMirror()

    @property
    def initted(self) -> bool:
        return hasattr(self, "_n_updates")

    def _lazy_init(self, x):
        rows = 1 if self.channelwise < 0 else x.shape[self.channelwise]
        self.weight = nn.Parameter(torch.zeros(rows, self.callback.weight_size, device=x.device), requires_grad=False)
        self._n_updates = state_parameter(torch.zeros(1, dtype=torch.int, device=x.device))

    def __setstate__(self, state):
        super().__setstate__(state)
        adopt_state_parameters(self)     # unpickling rebuilds plain Parameters

    def is_active(self) -> bool:
        """whether the next forward quantizes (used by the fused prune->quantize path)."""
        if self.timeout <= 0 or not self.initted:
            return False
        t = self._steps.read(self._n_updates)
        return t >= self.timeout and (self.training or self._quantized)

    def single_call_step(self, x: torch.Tensor, t: int, pre_relu: bool = False, act_in=None):
        """the active step of a tensor-wise ScalerQuantizer on a dense GPU tensor through one call into the library
        (qs_quantize_step); None when this layer / input is not one it covers -- the caller then takes the protocol route
        (callback.optimize, callback.forward), which the composite reproduces launch for launch"""
        cb = self.callback
        if (type(cb) is not ScalerQuantizer or self.channelwise != -1 or cb.group_num > 0 or cb.backward_passthrough
                or not x.is_cuda or x.dim() < 2 or x.numel() == 0 or x.dtype not in (torch.float32, torch.bfloat16, torch.float16)
                or x.data_ptr() % 16 or not _dense(x) or _hip.logging_events()
                or self.weight.device != x.device or self._n_updates.device != x.device or _callback_hooked(cb)):
            return None
        update = self.training
        if not update and not self._quantized:
            return None
        if not pre_relu and self.batch_dimension == 0:
            # an activation quantizer with no foldable activation in front (nn.GELU, nothing at all, the network input): under
            # autocast it folds the identity so that it can hand out its image like the others (fused.identity_fold_handle)
            from qsparse_amd import fused
            pre_relu = fused.identity_fold_handle(x)
        t_dev = cb.device_t(x.device) if (update and get_option("graph_safe")) else None
        mode = _hip.QSTEP_ALL if update else _hip.QSTEP_APPLY
        if update and self.batch_dimension == 0 and qdist.exchange_active():
            # an activation's abs-max is exchanged between ranks (weights are identical on every rank): the abs-max launch,
            # ONE all-reduce (MAX) of the accumulator lines, then running scale + quantization -- two calls around the collective.
            # (The lines hold non-negative floats: reduced as int32 their order is the floats' and a NaN stays the maximum.)
            acc = cb.tensor_accumulator(x.device)
            if self.__dict__.get("_qs_accumulator_armed"):
                acc.zero_()              # an earlier step died between the two calls (failed collective): stale maxima
            self.__dict__["_qs_accumulator_armed"] = True
            _hip.quantize_step(x, None, None, acc, self.weight.data, self.bits, cb.t, None, None, pre_relu, _hip.QSTEP_ABSMAX)
            if qdist.mailbox_enabled():        # (the lines travel through the site's mailbox: no host collective)
                qdist.mailbox_max_(self, acc)
            else:
                qdist.allreduce_max_(acc.view(torch.int32))
            mode = _hip.QSTEP_FINISH
        image_dtype, stat = None, None
        if pre_relu:                # a quantize-only activation site (convert's Sequential(act, QuantizeLayer)): the autocast image
            from qsparse_amd import fused
            stat = self.__dict__.get("_qs_image_stat")
            if stat is None:
                stat = self.__dict__["_qs_image_stat"] = fused.ImageStat()
            fused.image_bookkeeping(stat)       # (nobody took the last image -- no autocast matmul behind this site: stop making them)
            if stat.image_ok and (not (torch.is_grad_enabled() and (x.requires_grad or act_in is not None)) or get_option("relu_gate")):
                image_dtype = fused.autocast_image_dtype()
        cell = None
        if pre_relu:
            from qsparse_amd import fused
            cell = fused.grad_image_cell(x) if act_in is None else None
        # (act_in: x is fused.act_rider(gelu, act_in) -- detached; the gradient flows to act_in through the GELU's backward)
        if act_in is None:
            y = _QuantStep.apply(x, self.weight.data, cb.tensor_accumulator(x.device) if update else None, self.bits, cb.t, t_dev,
                                 self._n_updates.data if update else None, pre_relu, mode, 1 if cb.flip_axis else 0, _out_dtype(x),
                                 cb.code_range(self.bits), image_dtype, cell)
        else:
            y = _QuantStep.apply(act_in, self.weight.data, cb.tensor_accumulator(x.device) if update else None, self.bits, cb.t, t_dev,
                                 self._n_updates.data if update else None, pre_relu, mode, 1 if cb.flip_axis else 0, _out_dtype(x),
                                 cb.code_range(self.bits), image_dtype, None, x)
        if type(y) is tuple:
            y = fused._as_dual(y[0], y[1], stat, img_b=y[2])
        self.__dict__["_qs_accumulator_armed"] = False
        if update:
            if t == self.timeout and get_option("log_during_train"):
                logging.warn(f"quantizing {self.name} with {self.bits} bits")
            cb._advance_t(t_dev, bumped_by_kernel=True)
            self._quantized = True
            self._steps.note_device_add(self._n_updates, 1)
        return y

    @_hip.keeps_layout
    def forward(self, x):
        if not self.initted:
            self._lazy_init(x)
        if self.timeout <= 0:   # timeout=0 disables the operator for good (reference :496)
            return x
        t = self._steps.read(self._n_updates)
        out = x
        counter_on_device = False
        if t >= self.timeout and isinstance(x, torch.Tensor):
            y = self.single_call_step(x, t)
            if y is not None:
                return y
        if t >= self.timeout:
            if self.training:
                if t == self.timeout and get_option("log_during_train"):
                    logging.warn(f"quantizing {self.name} with {self.bits} bits")
                extra = {}
                if _hip.on_hip(x) and type(self.callback) in (DecimalQuantizer, ScalerQuantizer):
                    self.callback.__dict__["_bumped_step_counter"] = False
                    extra["step_counter"] = self._n_updates   # bumped inside the running-mean kernel
                new_weight = self.callback.optimize(x, self.bits, self.weight, batched=self.batch_dimension == 0,
                                                    channel_index=self.channelwise, **extra)
                counter_on_device = bool(extra) and self.callback.__dict__.get("_bumped_step_counter", False)
                if new_weight is not None and new_weight is not self.weight:
                    self.weight.data[:] = new_weight
                self._quantized = True
            if self._quantized:
                out = self.callback(x, self.bits, self.weight, channel_index=self.channelwise,
                                    inplace=self.batch_dimension == 0)
        if self.training:
            if counter_on_device:
                self._steps.note_device_add(self._n_updates, 1)
            else:
                self._steps.add(self._n_updates, 1)
        return out


def quantize(inp: nn.Module = None, bits: int = 8, channelwise: int = 1, timeout: int = 1000,
             callback: BaseQuantizer = None, bias_bits: int = -1, name: str = "") -> nn.Module:
    """build a ``QuantizeLayer`` (no ``inp``) or wrap ``inp`` so that its weight (and, with
    ``bias_bits``, its bias) is read through one (reference quantize.py:521-585)."""
    callback = callback or ScalerQuantizer()
    kwargs = dict(bits=bits, channelwise=channelwise, timeout=timeout, callback=callback, bias_bits=bias_bits, name=name)

    def make(batch_dimension=0, is_bias=False):
        if is_bias and bias_bits == -1:
            return lambda a: a
        return QuantizeLayer(bits=bias_bits if is_bias else bits,
                             channelwise=(0 if channelwise >= 0 else -1) if is_bias else channelwise,
                             timeout=int(timeout), callback=callback, name=name, batch_dimension=batch_dimension)

    if inp is None:
        layer = make()
        layer._kwargs = kwargs
        return layer
    if isinstance(inp, nn.Module):
        return imitate(inp, "quantize", make(-1), make(-1, is_bias=True))
    raise ValueError(f"{inp} is not a valid argument for quantize")
