"""Small shared helpers (mirrors the role of the reference's qsparse/common.py:7-19)."""
from typing import Union

import torch

TensorOrInt = Union[int, torch.Tensor]
TensorOrFloat = Union[float, torch.Tensor]


def ensure_tensor(v) -> torch.Tensor:
    """pass tensors through, wrap anything else with ``torch.tensor`` (reference common.py:14-19)."""
    return v if isinstance(v, torch.Tensor) else torch.tensor(v)


class HostMirror:
    """Host-side copy of a one-element state tensor (``_n_updates``, ``t`` ...).

    The reference reads such counters back with ``.item()`` several times per forward
    (qsparse/quantize.py:495, qsparse/sparse.py:213,251-269,56,88,107), i.e. one device->host sync each.
    Here the value is read once and then tracked on the host; the tensor is still updated (with an
    asynchronous in-place op) so ``state_dict()`` stays exact.  A write by anyone else -- e.g.
    ``load_state_dict`` (bumps ``_version``) or ``preload_qsparse_state_dict`` (replaces the Parameter)
    -- is detected and triggers one re-read.
    """

    __slots__ = ("_obj", "_version", "_value")

    def __init__(self):
        self._obj, self._version, self._value = None, -1, None

    def read(self, p: torch.Tensor):
        if p is not self._obj or p._version != self._version:
            self._value = p.item()
            self._obj, self._version = p, p._version
        return self._value

    def add(self, p: torch.Tensor, delta: int = 1):
        cur = self.read(p)
        with torch.no_grad():
            p.add_(delta)
        self._value, self._version = cur + delta, p._version

    def note_device_add(self, p: torch.Tensor, delta: int = 1):
        """the tensor was (or will be, in stream order) incremented by a kernel through its raw pointer:
        only the host copy needs updating."""
        self._value = self.read(p) + delta

    def write(self, p: torch.Tensor, value):
        with torch.no_grad():
            p.fill_(value)
        self._obj, self._version = p, p._version
        self._value = f32_round(float(value)) if p.dtype == torch.float32 else value

    def invalidate(self):
        """forget the host copy (the tensor was advanced behind our back, e.g. by hipGraph replays)"""
        self._obj, self._version, self._value = None, -1, None

    def __deepcopy__(self, memo):
        return HostMirror()  # a copied layer re-reads its own (copied) tensor on first use


def f32_round(v: float) -> float:
    """value of ``v`` after a round trip through an fp32 tensor (what ``t[0] = v; t.item()`` yields)."""
    import struct
    return struct.unpack("f", struct.pack("f", v))[0]
