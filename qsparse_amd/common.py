"""Small shared helpers (mirrors the role of the reference's qsparse/common.py:7-19)."""
from typing import Union

import torch

TensorOrInt = Union[int, torch.Tensor]
TensorOrFloat = Union[float, torch.Tensor]


def ensure_tensor(v) -> torch.Tensor:
    """pass tensors through, wrap anything else with ``torch.tensor`` (reference common.py:14-19)."""
    return v if isinstance(v, torch.Tensor) else torch.tensor(v)


_TENSOR_DATA = torch.Tensor.data   # the C-level getset descriptor behind ``tensor.data``


class StateParameter(torch.nn.Parameter):
    """``nn.Parameter`` for the one-element state tensors that are mirrored on the host (``_n_updates``,
    ``_cur_sparsity``, ``callback.t``).

    The reference reads these with ``.item()`` on every forward (qsparse/quantize.py:495, qsparse/sparse.py:251-269), so
    a write by ANY route -- including its own idiom ``layer._n_updates.data[:] = v`` -- takes effect on the next
    forward.  A host mirror needs to notice such writes without reading device memory.  In-place operations on the
    parameter or on a ``detach()`` alias bump ``Tensor._version``; the one route that does not is ``.data``, whose
    alias has a version counter of its own.  This subclass closes it: ``.data`` returns ``detach()`` (same storage,
    same values, *shared* version counter), and assigning ``.data`` (``Module.to`` / ``.cuda()``) advances an epoch
    the mirror compares as well.  Nothing else changes: ``isinstance(p, nn.Parameter)`` holds, ``state_dict`` /
    ``load_state_dict`` / ``deepcopy`` behave as for a plain Parameter.  (In-place writes through ``.data`` are
    therefore visible to autograd's version checks; these tensors are integer / ``requires_grad=False`` state that no
    graph saves.)
    """

    @property
    def data(self):
        return self.detach()

    @data.setter
    def data(self, value):
        _TENSOR_DATA.__set__(self, value)
        self.__dict__["_qs_epoch"] = self.__dict__.get("_qs_epoch", 0) + 1

    def __repr__(self):
        return "Parameter containing:\n" + repr(self.detach())


STATE_KEYS = ("_n_updates", "_cur_sparsity", "t")   # parameter names that are mirrored on the host


def state_parameter(t: torch.Tensor) -> "StateParameter":
    return StateParameter(t, requires_grad=False)


def adopt_state_parameters(module: torch.nn.Module):
    """make the mirrored counters of ``module`` (not its children) tracked Parameters, in place"""
    for key in STATE_KEYS:
        p = module._parameters.get(key)
        if p is not None:
            _adopt(p)


def _adopt(p):
    """plain Parameters (created by ``preload_qsparse_state_dict``, unpickling, user code) become tracked ones"""
    if type(p) is torch.nn.Parameter:
        p.__class__ = StateParameter
    return p


class HostMirror:
    """Host-side copy of a one-element state tensor (``_n_updates``, ``t`` ...).

    The reference reads such counters back with ``.item()`` several times per forward
    (qsparse/quantize.py:495, qsparse/sparse.py:213,251-269,56,88,107), i.e. one device->host sync each.
    Here a GPU tensor is read once and then tracked on the host; the tensor is still updated (with an
    asynchronous in-place op or by the kernels) so ``state_dict()`` stays exact.  A write by anyone else is
    detected and triggers one re-read: ``load_state_dict`` / in-place ops / ``.data`` writes through
    ``Tensor._version`` (see ``StateParameter``), ``preload_qsparse_state_dict`` and re-assignment through the
    Parameter's identity, ``.to()`` / ``.cuda()`` through the epoch of ``StateParameter``.  CPU tensors are read
    directly every time -- there is no sync to save.
    """

    __slots__ = ("_obj", "_version", "_epoch", "_value")
    track_cpu = False    # tests set this to run the tracking logic (normally GPU-only) on CPU tensors

    def __init__(self):
        self._obj, self._version, self._epoch, self._value = None, -1, -1, None

    def _stamp(self, p):
        if p.is_inference():             # no version counter to watch (created under torch.inference_mode()): never trusted
            self._obj = None
            return
        self._obj, self._version, self._epoch = p, p._version, p.__dict__.get("_qs_epoch", 0)

    def read(self, p: torch.Tensor):
        if (not p.is_cuda and not HostMirror.track_cpu) or p.is_inference():
            return p.item()
        if p is not self._obj or p._version != self._version or p.__dict__.get("_qs_epoch", 0) != self._epoch:
            self._value = _adopt(p).item()
            self._stamp(p)
        return self._value

    def add(self, p: torch.Tensor, delta: int = 1):
        cur = self.read(p)
        with torch.no_grad():
            p.add_(delta)
        self._value = cur + delta
        self._stamp(p)

    def note_device_add(self, p: torch.Tensor, delta: int = 1):
        """the tensor was (or will be, in stream order) incremented by a kernel through its raw pointer:
        only the host copy needs updating."""
        self._value = self.read(p) + delta

    def write(self, p: torch.Tensor, value):
        with torch.no_grad():
            p.fill_(value)
        _adopt(p)
        self._stamp(p)
        self._value = f32_round(float(value)) if p.dtype == torch.float32 else value

    def invalidate(self):
        """forget the host copy (the tensor was advanced behind our back, e.g. by hipGraph replays)"""
        self._obj, self._version, self._epoch, self._value = None, -1, -1, None

    def __deepcopy__(self, memo):
        return HostMirror()  # a copied layer re-reads its own (copied) tensor on first use


def f32_round(v: float) -> float:
    """value of ``v`` after a round trip through an fp32 tensor (what ``t[0] = v; t.item()`` yields)."""
    import struct
    return struct.unpack("f", struct.pack("f", v))[0]
