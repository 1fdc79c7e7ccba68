"""CPU oracle for the quantize/prune hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement, on CPU tensors, of the arithmetic that mlzxy/qsparse v2.0.1 performs on
its quantize/prune path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; ``qsparse_amd`` never does (``tests/test_boundary.py`` greps for
it).  The reference is a chain of eager ATen CPU operators, so the oracle states the same chain with
the same operator order and the same rounding points, but as plain functions over explicit state
(no nn.Module, no autograd, no hidden counters).

PARITY IS PINNED: every function below is checked bit-for-bit against fixtures recorded from the real
reference (``tests/golden/*.npz``, written by ``tests/golden/generate.py`` which imports
/root/reference); see ``tests/test_oracle_golden.py``.  A second, independent plain-C restatement of
the element-wise and order-statistic pieces lives in ``oracle/qs_oracle.c`` and is cross-checked
against this file in the same test module.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math
from typing import List, Optional, Sequence, Union

import torch

Number = Union[int, float]


# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------
def _bshape(ndim: int, channel_index: int) -> List[int]:
    """broadcast shape that puts a per-channel vector on `channel_index` (quantize.py:100-107)."""
    s = [1] * ndim
    s[channel_index] = -1
    return s


def _per_channel(p, x: torch.Tensor, channel_index: int):
    """view a (C,1)/(C,) parameter onto the channel axis when it has more than one element,
    otherwise return it untouched  (quantize.py:48-53, 101-107)."""
    if isinstance(p, torch.Tensor) and p.numel() > 1:
        assert len(p) == x.shape[channel_index]
        return p.view(*_bshape(x.dim(), channel_index))
    return p


# ----------------------------------------------------------------------------------------------
# A1/A2  scaler quantizer  (qsparse/quantize.py:87-131)
# ----------------------------------------------------------------------------------------------
def scaler_codes(x: torch.Tensor, scaler, channel_index: int = -1) -> torch.Tensor:
    """int32 codes  q = int(round(x / s))   (quantize.py:109).  True division, half-to-even."""
    s = _per_channel(scaler, x, channel_index)
    return (x / s).round().int()


def scaler_fwd(x: torch.Tensor, bits: int, scaler, channel_index: int = -1) -> torch.Tensor:
    """y = float(q) * s  (quantize.py:109-117).  The clamp at :110-116 acts on a temporary and is
    therefore dead: outputs do NOT saturate; `use_uint` and `bits` do not influence the forward."""
    s = _per_channel(scaler, x, channel_index)
    q = (x / s).round().int()
    return q.float() * s


def ste_bounds(bits: int, step, flip_axis: bool = False):
    """gradient clamp interval [(-L+notch)*s, (L-1+notch)*s]  (quantize.py:69-75, 123-129)."""
    limit = 2.0 ** (bits - 1)
    notch = 1 if flip_axis else 0
    return (-limit + notch) * step, (limit - 1 + notch) * step


def ste_bwd(g: torch.Tensor, bits: int, step, channel_index: int = -1, flip_axis: bool = False,
            backward_passthrough: bool = False, x_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """backward of the scaler / decimal quantizers (quantize.py:66-77, 120-131): the incoming
    gradient VALUES are clamped to the interval above (line :76/:130 zeroes nothing but NaNs because `v` aliases
    `grad_output`: pinned by fixture F17); autograd then casts the result to the forward input's dtype.
    `step` is the scaler (ScalerQuantization) or 2**-decimal (DecimalQuantization)."""
    if backward_passthrough:
        out = g
    else:
        s = _per_channel(step, g, channel_index)
        if not isinstance(s, torch.Tensor):
            s = torch.tensor(s)
        lo, hi = ste_bounds(bits, s, flip_axis)
        out = g.clone().clamp_(lo, hi)
        out[out != out] = 0          # :76/:130 -- `v` IS grad_output, so `v != grad_output` is true for NaNs only: they become 0
    return out if x_dtype is None else out.to(x_dtype)


# ----------------------------------------------------------------------------------------------
# A3  decimal (power-of-two) quantizer  (qsparse/quantize.py:31-63)
# ----------------------------------------------------------------------------------------------
def decimal_codes(x: torch.Tensor, decimal, channel_index: int = -1) -> torch.Tensor:
    """q = int(x * 2**d): truncation toward zero (quantize.py:45-55)."""
    toi = 2.0 ** decimal
    toi = _per_channel(toi, x, channel_index)
    return (x * toi).int()


def decimal_fwd(x: torch.Tensor, bits: int, decimal, channel_index: int = -1) -> torch.Tensor:
    """y = float(q) * 2**-d   (quantize.py:44-63); the clamp at :56-62 is dead code."""
    tof = 2.0 ** -decimal
    toi = 2.0 ** decimal
    tof = _per_channel(tof, x, channel_index)
    toi = _per_channel(toi, x, channel_index)
    q = (x * toi).int()
    return q.float() * tof


def decimal_from_scale(scale: torch.Tensor) -> torch.Tensor:
    """d = round(log2(1/s)) with non-finite reciprocals mapped to 1  (quantize.py:316)."""
    return (1 / scale).nan_to_num(posinf=1, neginf=1).log2().round()


# ----------------------------------------------------------------------------------------------
# A4  line (asymmetric) quantizer  (qsparse/quantize.py:141-185); backward is the identity
# ----------------------------------------------------------------------------------------------
def line_fwd(x: torch.Tensor, bits: int, lines, channel_index: int = -1,
             float_zero_point: bool = True) -> torch.Tensor:
    N = 2 ** bits
    shape = [1] * x.dim()
    if not isinstance(lines, torch.Tensor):
        lines = torch.tensor(lines).view(-1, 2)                      # :151-152
    if channel_index >= 0:
        shape[channel_index] = -1                                     # :153-155
        assert x.shape[channel_index] == lines.shape[0]
    assert lines.shape[1] == 2
    start, end = lines[:, 0].view(shape), lines[:, 1].view(shape)     # :157
    xc = torch.clamp(x, start, end)                                   # :158
    step = (end - start) / N                                          # :159
    step = torch.where(step == 0, torch.full_like(step, 0.0001), step)  # :160
    if not float_zero_point:                                          # :161-166
        qa = (xc / step).round()
        qstart = (start / step).round()
        qa = (qa - qstart).clamp(0, N - 1)
        return (qa + qstart) * step
    qa = xc - start                                                   # :175-181 (same values as :168-173)
    qa = qa / step
    qa = qa.round().clamp(0, N - 1)
    qa = qa * step
    return qa + start


# ----------------------------------------------------------------------------------------------
# A5  scale statistics  (qsparse/quantize.py:327-349, 393-430)
# ----------------------------------------------------------------------------------------------
def absmax_scale(x: torch.Tensor, bits: int, channel_index: int = -1, batched: bool = False) -> torch.Tensor:
    """new scale = max|x| / 2**(bits-1), over the whole tensor or per channel; shape (C or 1, 1)
    (quantize.py:329-343).  The batched channel-wise case with batch > 1 fails in the reference at
    the `.view(-1, batch_size)` of :341-342 exactly when C*... is inconsistent; restated verbatim."""
    a = x.abs()
    batch_size = a.shape[0] if batched else -1
    rows = 1 if channel_index < 0 else a.shape[channel_index]
    if channel_index == -1:
        a = a.view(1, -1)
    elif channel_index != 0:
        a = a.transpose(0, channel_index).contiguous().view(rows, -1)
    else:
        a = a.view(a.shape[0], -1)
    new = a.max(dim=1).values / (2 ** (bits - 1))
    if batched and channel_index >= 0:
        new = new.view(-1, batch_size).mean(dim=1)
    return new.view(rows, 1)


def running_mean_absmax(weight: torch.Tensor, new: torch.Tensor, t: int) -> torch.Tensor:
    """t == 0: w = new;  else w = (t*w + new)/(t+1)  with a Python-int t  (quantize.py:344-348)."""
    if t == 0:
        return new
    return (t * weight + new) / (t + 1)


def adaptive_lines(x: torch.Tensor, channel_index: int = -1, batched: bool = False) -> torch.Tensor:
    """per-row (min, max) then min-of-mins / max-of-maxes over the batch  (quantize.py:393-420)."""
    batch_size = x.shape[0]
    if channel_index >= 0:
        if batched:
            if channel_index != 1:
                # (:399-401 `.view` the transposed tensor: a RuntimeError for every batch of more than one sample -- restated on a
                #  contiguous stand-in so that the verdict does not depend on the layout x happens to have)
                torch.empty(tuple(x.shape), device="meta").transpose(1, channel_index).view(-1, math.prod(tuple(x.transpose(1, channel_index).shape)[2:]))
                x = x.transpose(1, channel_index)
            shape = tuple(x.shape)
            x = x.reshape(-1, math.prod(shape[2:]))
        else:
            if channel_index != 0:
                x = x.transpose(0, channel_index)
            shape = tuple(x.shape)
            x = x.contiguous().view(-1, math.prod(shape[1:]))
    else:
        x = x.reshape(len(x) if batched else 1, -1)
    lb = x.min(dim=1).values
    ub = x.max(dim=1).values
    lines = torch.stack([lb, ub], dim=1)
    if batched:
        lines = lines.view(batch_size, -1, 2)
        lines = torch.stack([lines[:, :, 0].min(dim=0).values, lines[:, :, 1].max(dim=0).values], dim=1)
    return lines.view(-1, 2)


def running_mean_adaptive(weight: torch.Tensor, new: torch.Tensor, t_after: int) -> torch.Tensor:
    """w = (w*(t-1) + new)/t with t already incremented (a Python int)  (quantize.py:427-430)."""
    return (weight * (t_after - 1) + new) / t_after


# ----------------------------------------------------------------------------------------------
# A6  staged mean to the mask shape  (qsparse/util.py:79-99) and magnitude update (sparse.py:82-89)
# ----------------------------------------------------------------------------------------------
def squeeze_mean(x: torch.Tensor, shape: Sequence[int]) -> torch.Tensor:
    """successive keepdim means over every dim where `shape` is 1 and x is not, ascending; each
    stage's result is rounded to x's dtype (ATen mean: fp32 accumulate, fp32 divide, round)."""
    assert x.dim() == len(shape), "mismatch between the input tensor and mask"
    for i, (sx, sm) in enumerate(zip(x.shape, shape)):
        if sx != sm:
            if sm != 1:
                raise ValueError("mismatch between the input tensor and mask")
            x = x.mean(i, keepdim=True)
    return x


def magnitude_update(magnitude: torch.Tensor, x: torch.Tensor, t: int, l0: bool = False) -> torch.Tensor:
    """m = (t*m + squeeze(|x|)) / (t+1) in fp32 (sparse.py:82-89); the L0 variant replaces x by
    float(x != 0) when min(x) == 0 (:85-86)."""
    if l0 and x.min().item() == 0:
        x = (x != 0).float()
    a = squeeze_mean(x.abs(), magnitude.shape)
    return (t * magnitude + a) / (t + 1)


# ----------------------------------------------------------------------------------------------
# A7  mask from importance  (qsparse/util.py:103-117)
# ----------------------------------------------------------------------------------------------
def kth_index(sparsity: float, n: int) -> int:
    """index of the threshold in the ascending order:  max(int(s*n - 1), 0) + 1  (util.py:115-116)."""
    return max(int(sparsity * n - 1), 0) + 1


def mask_from_importance(importance: torch.Tensor, sparsity: float) -> torch.Tensor:
    values = importance.flatten().sort()[0]
    threshold = values[kth_index(sparsity, len(values))]
    return importance >= threshold


# ----------------------------------------------------------------------------------------------
# A8  sparsity schedule  (qsparse/sparse.py:186-188, 251-257)
# ----------------------------------------------------------------------------------------------
def schedule_steps(start: int, interval: int, repetition: int, rampup: bool) -> List[int]:
    return [start + interval * ((1 if rampup else 0) + i) for i in range(repetition)]


def scheduled_sparsity(n: int, sparsity: float, start: int, interval: int, repetition: int, rampup: bool) -> float:
    """value stored into the fp32 `_cur_sparsity` at step n, read back as a Python float."""
    rampup_interval = 0 if rampup else interval
    ratio = (1.0 - (n - start + rampup_interval) / (interval * repetition)) ** 3
    return torch.tensor(sparsity * (1 - ratio), dtype=torch.float32).item()


# ----------------------------------------------------------------------------------------------
# A9  mask apply (sparse.py:66,116,122,263) and its autograd backward
# ----------------------------------------------------------------------------------------------
def mask_apply(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return x * mask


# ----------------------------------------------------------------------------------------------
# layer state machines as explicit-state simulators
# ----------------------------------------------------------------------------------------------
class QuantizeSim:
    """QuantizeLayer + Scaler/Decimal/Adaptive quantizer  (quantize.py:275-518) with explicit state.

    state: weight (C|1, weight_size) fp32, n_updates int, quantized bool, t (callback counter; shared
    between a weight layer and its bias layer in the reference -- pass the same `shared` dict)."""

    def __init__(self, kind: str = "scaler", bits: int = 8, channelwise: int = 1, timeout: int = 1000,
                 batch_dimension: int = 0, shared: Optional[dict] = None, flip_axis: bool = False,
                 backward_passthrough: bool = False):
        assert kind in ("scaler", "decimal", "adaptive")
        self.kind, self.bits, self.channelwise, self.timeout = kind, bits, channelwise, int(timeout)
        self.batch_dimension = batch_dimension
        self.shared = shared if shared is not None else {"t": 0}
        self.flip_axis, self.backward_passthrough = flip_axis, backward_passthrough
        self.weight = None
        self.n_updates = 0
        self.quantized = False

    def step(self, x: torch.Tensor, training: bool = True) -> torch.Tensor:
        if self.weight is None:                                            # quantize.py:482-493
            rows = 1 if self.channelwise < 0 else x.shape[self.channelwise]
            self.weight = torch.zeros(rows, 2 if self.kind == "adaptive" else 1)
        t = self.n_updates
        out = x
        if self.timeout > 0:                                               # :496-517
            if t >= self.timeout:
                if training:
                    batched = self.batch_dimension == 0
                    if self.kind == "adaptive":
                        new = adaptive_lines(x, self.channelwise, batched)
                        # QuantizeLayer always hands its `weight` in, so the `weight is None` branch
                        # (:421-425) is never taken and `t` stays the Python int of :307
                        self.shared["t"] += 1
                        self.weight = running_mean_adaptive(self.weight, new, self.shared["t"]).to(torch.float32)
                    else:
                        new = absmax_scale(x, self.bits, self.channelwise, batched)
                        self.weight = running_mean_absmax(self.weight, new, self.shared["t"]).to(torch.float32)
                        self.shared["t"] += 1
                    self.quantized = True
                if self.quantized:
                    out = self.apply(x, training)
            if training:
                self.n_updates += 1
        return out

    def apply(self, x: torch.Tensor, training: bool = True) -> torch.Tensor:
        if self.kind == "scaler":
            return scaler_fwd(x, self.bits, self.weight, self.channelwise)
        if self.kind == "decimal":
            return decimal_fwd(x, self.bits, decimal_from_scale(self.weight), self.channelwise)
        return line_fwd(x, self.bits, self.weight, self.channelwise, float_zero_point=training)

    def grad(self, g: torch.Tensor, x_dtype: torch.dtype) -> torch.Tensor:
        """gradient w.r.t. the layer input for the most recent `step` (identity while inactive)."""
        if not self.quantized or self.kind == "adaptive":
            return g.to(x_dtype)
        step = self.weight if self.kind == "scaler" else 2.0 ** -decimal_from_scale(self.weight)
        return ste_bwd(g, self.bits, step, self.channelwise, self.flip_axis, self.backward_passthrough, x_dtype)


class PruneSim:
    """PruneLayer + MagnitudePruningCallback (sparse.py:18-122, 157-273) with explicit state."""

    def __init__(self, sparsity: float = 0.5, dimensions=(1,), start: int = 1000, interval: int = 1000,
                 repetition: int = 4, rampup: bool = False, mask_refresh_interval: int = -1,
                 stop_mask_refresh: float = float("inf"), running_average: bool = True, l0: bool = False,
                 use_gradient: bool = False):
        assert running_average or not use_gradient                         # sparse.py:44-47
        self.use_gradient, self.hook_armed = use_gradient, False
        self.sparsity, self.dimensions = sparsity, set(dimensions)
        self.start, self.interval, self.repetition, self.rampup = int(start), int(interval), repetition, rampup
        self.schedules = schedule_steps(self.start, self.interval, repetition, rampup)
        self.mask_refresh_interval, self.stop_mask_refresh = mask_refresh_interval, stop_mask_refresh
        self.running_average, self.l0 = running_average, l0
        self.mask = None
        self.n_updates = 0
        self.cur_sparsity = 0.0
        self.t = -1
        self.magnitude = None

    def step(self, x: torch.Tensor, training: bool = True, requires_grad: bool = True) -> torch.Tensor:
        """`requires_grad`: whether the layer input requires grad (only the use_gradient mode looks at it, :73)."""
        self._x_requires_grad = requires_grad
        if self.mask is None:                                              # sparse.py:228-249
            assert x.dim() > 1
            self.mask = torch.ones([s if i in self.dimensions else 1 for i, s in enumerate(x.shape)], dtype=torch.bool)
        if self.n_updates in self.schedules and training:                  # :251-257
            self.cur_sparsity = scheduled_sparsity(self.n_updates, self.sparsity, self.start, self.interval,
                                                   self.repetition, self.rampup)
        if not training or self.mask.numel() == 1:                         # :262-263
            return x * self.mask
        if self.n_updates >= self.start:                                   # :265-269
            out = self._callback(x, self.cur_sparsity)
        else:
            out = x
        self.n_updates += 1
        return out

    def _callback(self, x: torch.Tensor, sparsity: float) -> torch.Tensor:   # sparse.py:99-120 (training branch)
        if self.t == -1:
            if self.running_average:
                self.magnitude = torch.zeros(self.mask.shape, dtype=torch.float32)
            self.t = 0
            if self.mask_refresh_interval <= 0:
                self.mask_refresh_interval = 1
        t = self.t
        if self.use_gradient:
            # receive_input (sparse.py:69-78): the previous hook is removed, a new one is registered on THIS input if it
            # requires grad; the magnitude is updated when (and if) that input's gradient arrives -- see receive_grad.
            # Past stop_mask_refresh receive_input is not called (:107-108): this step's input carries no hook (the old
            # one stays on the old input, whose backward has long run)
            self.hook_armed = t < self.stop_mask_refresh and bool(getattr(self, "_x_requires_grad", True))
        elif t < self.stop_mask_refresh and self.running_average:
            self.magnitude = magnitude_update(self.magnitude, x.detach(), t, self.l0)
        if sparsity >= 0 and (t % self.mask_refresh_interval == 0 and t <= self.stop_mask_refresh) and (
                t > 0 or not self.running_average):
            importance = self.magnitude if self.running_average else squeeze_mean(x.detach().abs(), self.mask.shape)
            self.mask = mask_from_importance(importance, sparsity)
        self.t += 1
        return x * self.mask

    def grad(self, g: torch.Tensor, active: bool = True) -> torch.Tensor:
        """autograd backward of `x * mask` (MulBackward0): g * mask in g's dtype."""
        return g * self.mask if active else g

    def receive_grad(self, grad_x: torch.Tensor):
        """use_gradient mode: the tensor hook of sparse.py:74-75 fires with the TOTAL gradient of the layer input (all
        consumers of x) during backward, i.e. after `forward` advanced t (:117): update_magnitude(grad) with the
        already-incremented t (:82-89).  Call once per backward that reaches the most recent training input."""
        if self.use_gradient and self.hook_armed:
            self.magnitude = magnitude_update(self.magnitude, grad_x.detach(), self.t, self.l0)
