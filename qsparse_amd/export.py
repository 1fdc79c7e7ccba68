"""Integer export of a converted network -- the inference-side step after quantization-aware training.

The reference has no export API; what it pins is the PROPERTY an export must have: its float-simulated quantized
layers equal true integer arithmetic on the codes (reference tests/test_quantize.py:73-101, built on
``DecimalQuantization.forward`` qsparse/quantize.py:44-63 truncating and on power-of-two scales).  This module gives
that property a product entry point:

    ex = qs.export_integer(model)            # {module path: LayerExport}
    e = ex["1.stages.3.conv2"]
    e.weight.codes        int32 (int8 view with .int8() when bits <= 8), the weight's shape
    e.weight.scale / .decimal / (.step, .zero_point)   what turns codes back into values
    e.weight.mask         bool, the weight-side pruning mask (if the layer is pruned)
    e.weight.dequantize() == the layer's effective weight, bit for bit

Every code tensor is the SECOND OUTPUT of the very quantizer call the layer's forward makes (``return_codes=True``:
on the GPU the kernels' own ``codes`` output, qs_quant_*_fwd in include/qsparse_hip.h) -- never a second evaluation of
the arithmetic.  Activation-side operators (``QuantizeLayer`` / ``PruneLayer`` modules in the tree) export their
parameters: scale or decimal or lines, bits, channel mask.  Nothing here touches the training state: quantizers are
read in evaluation mode and no counter moves.
"""
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn as nn

from qsparse_amd.quantize import QuantizeLayer
from qsparse_amd.sparse import PruneLayer
from qsparse_amd.util import logging


@dataclass
class QuantizedTensor:
    """integer form of one quantized tensor.  ``kind``: "scaler" (``values = codes * scale``), "decimal"
    (``values = codes * 2^-decimal``) or "line" (``values = (codes + zero_point) * step``); per-channel parameters have one
    entry per index of ``channel_index`` (-1: tensor-wise)."""
    kind: str
    bits: int
    channel_index: int
    codes: torch.Tensor
    values: torch.Tensor = field(repr=False)
    scale: Optional[torch.Tensor] = None
    decimal: Optional[torch.Tensor] = None
    lines: Optional[torch.Tensor] = None
    step: Optional[torch.Tensor] = None
    zero_point: Optional[torch.Tensor] = None
    mask: Optional[torch.Tensor] = None

    def _on_channel(self, p: torch.Tensor) -> torch.Tensor:
        if p.numel() == 1 or self.channel_index < 0:
            return p.reshape(())
        view = [1] * self.codes.dim()
        view[self.channel_index] = -1
        return p.reshape(view)

    def dequantize(self) -> torch.Tensor:
        """float32 values from the integers alone -- equal, bit for bit, to what the layer computes with"""
        q = self.codes.float()
        if self.kind == "scaler":
            return q * self._on_channel(self.scale)
        if self.kind == "decimal":      # 2^-d evaluated on the host (C-sized): exact powers of two whatever the device's pow does
            return q * self._on_channel(torch.pow(2.0, -self.decimal.cpu().float()).to(q.device))
        return (q + self._on_channel(self.zero_point.float())) * self._on_channel(self.step)

    def int8(self) -> torch.Tensor:
        """the codes as int8 -- raises if a code does not fit.  The reference's forward does not saturate (the tensor's
        largest element maps to +2^(bits-1), one above the two's-complement range): train / export with
        ``set_qsparse_options(saturate=True)`` or ``ScalerQuantizer(saturate=True)`` and every code fits its bit width."""
        lo, hi = (int(self.codes.min()), int(self.codes.max())) if self.codes.numel() else (0, 0)
        if lo < -128 or hi > 127:
            raise OverflowError(f"codes span [{lo}, {hi}]: not representable in int8 (no forward saturation, quirk B1; "
                                "see the `saturate` option)")
        return self.codes.to(torch.int8)

    def uint8(self) -> torch.Tensor:
        """the codes as uint8 (``use_uint`` quantizers under ``saturate``, line quantizers of up to 8 bits)"""
        lo, hi = (int(self.codes.min()), int(self.codes.max())) if self.codes.numel() else (0, 0)
        if lo < 0 or hi > 255:
            raise OverflowError(f"codes span [{lo}, {hi}]: not representable in uint8")
        return self.codes.to(torch.uint8)

    def int4_packed(self) -> torch.Tensor:
        """two 4-bit codes per byte, flat in memory order: element 2i in the low nibble, 2i + 1 in the high one, two's
        complement for signed codes in [-8, 7] and plain for unsigned ones in [0, 15] (an odd count pads with 0).  Raises
        when a code needs more than four bits -- without ``saturate`` a 4-bit tensor's largest element is code +8 (quirk B1)."""
        q = self.codes.reshape(-1)
        lo, hi = (int(q.min()), int(q.max())) if q.numel() else (0, 0)
        if not ((lo >= -8 and hi <= 7) or (lo >= 0 and hi <= 15)):
            raise OverflowError(f"codes span [{lo}, {hi}]: not representable in 4 bits (see the `saturate` option)")
        if q.numel() % 2:
            q = torch.cat([q, q.new_zeros(1)])
        nib = (q & 0xF).to(torch.uint8).view(-1, 2)
        return nib[:, 0] | (nib[:, 1] << 4)

    @staticmethod
    def unpack_int4(packed: torch.Tensor, numel: int, signed: bool = True) -> torch.Tensor:
        """inverse of ``int4_packed``: int32 codes, flat"""
        b = packed.to(torch.int32)
        q = torch.stack([b & 0xF, (b >> 4) & 0xF], dim=1).view(-1)[:numel]
        return torch.where(q >= 8, q - 16, q) if signed else q


@dataclass
class LayerExport:
    """``weight`` / ``bias``: quantized parameters of a wrapped layer (None when that parameter is not quantized);
    for an activation operator ``activation`` holds the parameters (codes are computed per input at run time)."""
    path: str
    module: str
    weight: Optional[QuantizedTensor] = None
    bias: Optional[QuantizedTensor] = None
    activation: Optional[dict] = None


def _operator_input(layer: nn.Module, op: nn.Module, attr: str):
    """the tensor ``op`` receives when ``layer.<attr>`` is read (the previous imitation's output, e.g. the pruned
    weight in front of the quantizer), captured with a forward pre-hook; the read happens in evaluation mode"""
    seen = []
    handle = op.register_forward_pre_hook(lambda m, args: seen.append(args[0]))
    try:
        getattr(layer, attr)
    finally:
        handle.remove()
    return seen[-1].detach() if seen else None


def _export_through(q: QuantizeLayer, x: torch.Tensor) -> Optional[QuantizedTensor]:
    if not q.initted or q.timeout <= 0 or int(q._n_updates.item()) < q.timeout:
        return None                          # never quantized: nothing to export
    if not q._quantized:
        # counters say "past the timeout" but the layer has not quantized in THIS process: a checkpoint loaded without
        # `load_extra_state_dict` (the reference's state_dict forgets `_quantized`, quirk B7).  In evaluation mode such a
        # layer passes its input through unquantized (reference quantize.py:505-508), so there is no integer form of what
        # it computes -- exporting codes here would break `dequantize() == effective weight`
        logging.warn(f"export_integer: {q.name or 'a QuantizeLayer'} is past its timeout but `_quantized` is False (checkpoint "
                     "loaded without qs.load_extra_state_dict?): it evaluates unquantized and is skipped")
        return None
    out = q.callback.export(x, q.bits, q.weight.detach(), q.channelwise)
    kind = out.pop("kind")
    return QuantizedTensor(kind=kind, bits=q.bits, channel_index=q.channelwise if q.weight.shape[0] > 1 else -1, **out)


def export_integer(model: nn.Module) -> Dict[str, LayerExport]:
    """integer codes and scales of every quantized parameter, and the parameters of every activation operator, of a
    network built with ``quantize()`` / ``prune()`` / ``convert()``.  The network is read in evaluation mode (restored
    afterwards); a quantizer that never reached its ``timeout`` is skipped."""
    out: Dict[str, LayerExport] = {}
    modes = {m: m.training for m in model.modules()}
    model.eval()
    try:
        with torch.no_grad():
            inside = set()
            for path, layer in model.named_modules():
                q, qb, p = (layer.__dict__.get("_modules", {}).get(k) for k in ("quantize", "quantize_bias", "prune"))
                if not isinstance(q, QuantizeLayer):
                    continue
                inside.update(id(m) for m in (q, qb, p) if m is not None)
                rec = LayerExport(path=path, module=type(layer).__name__)
                w_in = _operator_input(layer, q, "weight")
                if w_in is not None:
                    rec.weight = _export_through(q, w_in)
                    if rec.weight is not None and isinstance(p, PruneLayer) and p.initted:
                        rec.weight.mask = p.mask.detach().clone()
                if isinstance(qb, QuantizeLayer) and layer._parameters.get("bias") is not None:
                    b_in = _operator_input(layer, qb, "bias")
                    if b_in is not None:
                        rec.bias = _export_through(qb, b_in)
                if rec.weight is not None or rec.bias is not None:
                    out[path] = rec
            for path, m in model.named_modules():
                if id(m) in inside:
                    continue
                if (isinstance(m, QuantizeLayer) and m.initted and m.timeout > 0 and int(m._n_updates.item()) >= m.timeout
                        and m._quantized):
                    act = dict(operator="quantize", bits=m.bits, channelwise=m.channelwise,
                               quantizer=type(m.callback).__name__, weight=m.weight.detach().clone())
                    out[path] = LayerExport(path=path, module="QuantizeLayer", activation=act)
                elif isinstance(m, PruneLayer) and m.initted and m.mask.numel() > 1:
                    out[path] = LayerExport(path=path, module="PruneLayer",
                                            activation=dict(operator="prune", mask=m.mask.detach().clone(),
                                                            sparsity=float(m._cur_sparsity.item())))
    finally:
        for m, was in modes.items():
            m.training = was
    return out
