"""ctypes binding of ``libqsparse_hip.so`` (C ABI: ``include/qsparse_hip.h``).

PyTorch is used here only as the owner of device memory and streams: every call passes raw
``data_ptr()`` values and the current HIP stream.  There is no fallback: a CUDA/HIP tensor reaching
this module without the library present raises ``QsparseHipError``.
"""
import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
MEAN_ABS, MEAN_L0, MEAN_RELU = 1, 2, 4
WS_KTH_VALUE = 1
MAX_DIMS = 6

ABI_VERSION = 26          # QS_ABI_VERSION of include/qsparse_hip.h this binding was written against: the version it NEEDS
_LIB_NAME = "libqsparse_hip.so"
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


class QsparseHipError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("QSPARSE_HIP_LIB", os.path.join(_PKG_DIR, _LIB_NAME))


# name -> (restype, argtypes); mirrors include/qsparse_hip.h one to one
_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
SIGNATURES = {
    "qs_version": (c_int, []),
    "qs_abi_floor": (c_int, []),
    "qs_quant_fwd_v": (c_int, [_P]),
    "qs_pq_select_v": (c_int, [_P]),
    "qs_site_fwd_v": (c_int, [_P, _P]),
    "qs_quantize_step_v": (c_int, [_P]),
    "qs_status_string": (c_char_p, [_I]),
    "qs_workspace_bytes": (c_size_t, [_I, _L]),
    "qs_quant_scaler_fwd": (c_int, [_P, _P, _P, _P, _L, _F, _P, _L, _L, _L, _I, _I, _I, _I, c_int32, c_int32, _I, _I, _P, _P, _I, _P, _P]),
    "qs_quant_decimal_fwd": (c_int, [_P, _P, _P, _P, _L, _F, _P, _L, _L, _L, _I, _I, _I, _I, c_int32, c_int32, _I, _I, _P, _P, _I, _P, _P]),
    "qs_quant_image_ok": (c_int, [_L, _L, _L, _I, _I, _I, _I]),
    "qs_quant_ste_relu_bwd": (c_int, [_P, _P, _P, _P, _P, _L, _F, _I, _F, _F, _P, _L, _L, _L, _I, _I, _I, _I, _P, _I, _P]),
    "qs_activation": (c_int, [_I, _F, _F]),
    "qs_quant_line_fwd": (c_int, [_P, _P, _P, _P, _L, _I, _I, _L, _L, _L, _I, _I, _P]),
    "qs_quant_ste_bwd": (c_int, [_P, _P, _P, _L, _F, _I, _F, _F, _I, _P, _L, _L, _L, _I, _I, _I, _P]),
    "qs_absmax": (c_int, [_P, _P, _I, _L, _L, _L, _I, _I, _I, _I, _P, c_size_t, _P]),
    "qs_minmax": (c_int, [_P, _P, _P, _I, _L, _L, _L, _I, _I, _P, c_size_t, _P]),
    "qs_scale_update": (c_int, [_P, _I, _P, _L, _L, _P, _I, _I, _I, _P, _I, _P]),
    "qs_lines_update": (c_int, [_P, _P, _P, _L, _L, _P, _I, _I, _P]),
    "qs_decimal_from_scale": (c_int, [_P, _P, _L, _P]),
    "qs_mean_dim": (c_int, [_P, _P, _L, _L, _L, _I, _I, _I, _P, _P, _L, _L, _L, _P]),
    "qs_l0_flag": (c_int, [_P, _L, _I, _P, _P, _P]),
    "qs_running_mean": (c_int, [_P, _P, _I, _L, _L, _P, _P]),
    "qs_kth_value": (c_int, [_P, _L, _L, _P, _P, c_size_t, _P]),
    "qs_mask_ge": (c_int, [_P, _P, _P, _L, _P]),
    "qs_mask_apply": (c_int, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "qs_pq_select": (c_int, [_P, _P, _I, _L, _I, _L, _I, _L, _P, _P, _L, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    "qs_mean_last2": (c_int, [_P, _P, _L, _L, _L, _I, _I, _P, _P, _L, _P, _P]),
    "qs_mean_dim_cl": (c_int, [_P, _P, _L, _L, _L, _I, _I, _I, _P, _P, _P]),
    "qs_token_stats": (c_int, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _P]),
    "qs_mean_cl_w": (c_int, [_P, _P, _L, _L, _L, _L, _I, _I, _I, _P, _P]),
    "qs_mean_dim_split": (c_int, [_P, _P, _L, _L, _L, _L, _I, _I, _I, _P, _P]),
    "qs_mailbox_bytes": (c_size_t, [_I, _L]),
    "qs_mailbox_alloc": (c_int, [c_size_t, _P]),
    "qs_mailbox_free": (c_int, [_P]),
    "qs_mailbox_export": (c_int, [_P, _P]),
    "qs_mailbox_open": (c_int, [_P, _P]),
    "qs_mailbox_close": (c_int, [_P]),
    "qs_mailbox_publish": (c_int, [_P, _L, _P, _I, _I, ctypes.c_uint32, _P]),
    "qs_mailbox_wait": (c_int, [_P, _I, _L, ctypes.c_uint32, _P, ctypes.c_uint32, _P, _P]),
    "qs_records_max": (c_int, [_P, _I, _L, _P, _P]),
    "qs_mean_strided": (c_int, [_P, _P, _L, _L, _I, _P, _P, _P, _I, _I, _L, _I, _I, _I, _P, _P]),
    "qs_multi_plan": (c_int, [_P, _I, _P, _P, _P, _P]),
    "qs_multi_absmax": (c_int, [_P, _I, _I, _P]),
    "qs_multi_scale_update": (c_int, [_P, _I, _I, _P]),
    "qs_multi_quant_fwd": (c_int, [_P, _I, _I, _P, _I, _P]),
    "qs_multi_magnitude": (c_int, [_P, _I, _I, _P]),
    "qs_multi_mask_refresh": (c_int, [_P, _I, _I, _I, _P]),
    "qs_multi_stage_plan": (c_int, [_P, _I, _P]),
    "qs_multi_stage_mean": (c_int, [_P, _I, _I, _P]),
    "qs_multi_ste_bwd": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "qs_quantize_step": (c_int, [_P, _P, _P, _P, _I, _P, _L, _I, _I, _I, _L, _P, _P, _I, _I, _I, c_int32, c_int32, _P, _P, _I, _P]),
    "qs_site_fwd": (c_int, [_P, _P, _P, _P, _I, _L, _L, _L, _P, _I, _P, _I, _P, _P, _P]),
    "qs_site_stats": (c_int, [_P, _P, _I, _P, _P]),
    "qs_site_bwd": (c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _P, _I, _P, _P]),
    "qs_site_bwd_v": (c_int, [_P, _P]),
    "qs_quant_ste_relu_bwd_v": (c_int, [_P]),
    "qs_stats_pack": (c_int, [_P, _I, _P, _L, _L, _P, _P]),
    "qs_stats_combine": (c_int, [_P, _I, _L, _P, _P, _L, _P]),
}



class SitePlanStruct(ctypes.Structure):
    """`qs_site_plan` of include/qsparse_hip.h, field for field"""
    _fields_ = [("N", c_int64), ("C", c_int64), ("H", c_int64), ("W", c_int64),
                ("layout", c_int32), ("xdt", c_int32), ("ydt", c_int32), ("bits", c_int32),
                ("magnitude", c_void_p), ("mask", c_void_p), ("scale", c_void_p), ("chan_absmax", c_void_p),
                ("absmax_stride", c_int64), ("stage", c_void_p), ("amax_part", c_void_p), ("stage_mean", c_void_p),
                ("prune_n_updates", c_void_p), ("quant_n_updates", c_void_p), ("callback_t", c_void_p),
                ("quantizer_t_dev", c_void_p), ("callback_t_from_device", c_int32),
                ("saturate", c_int32), ("code_lo", c_int32), ("code_hi", c_int32), ("act", c_int32),
                ("elide_mask", c_void_p), ("absmax_dense", c_void_p), ("reduce_ws", c_void_p), ("reduce_ws_bytes", c_int64)]


class QuantFwdArgs(ctypes.Structure):
    """`qs_quant_fwd_args` of include/qsparse_hip.h"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("kind", c_int32), ("x", c_void_p), ("y", c_void_p), ("codes", c_void_p),
                ("param", c_void_p), ("nparam", c_int64), ("param_host", c_float), ("xdt", c_int32), ("ydt", c_int32), ("qdt", c_int32),
                ("chan_mask", c_void_p), ("outer", c_int64), ("C", c_int64), ("inner", c_int64),
                ("saturate", c_int32), ("code_lo", c_int32), ("code_hi", c_int32), ("pre_relu", c_int32), ("elide_masked", c_int32),
                ("imgdt", c_int32), ("gate_out", c_void_p), ("image_out", c_void_p), ("xback_out", c_void_p), ("stream", c_void_p)]


class PqSelectArgs(ctypes.Structure):
    """`qs_pq_select_args` of include/qsparse_hip.h"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("sdt", c_int32), ("stat_dt", c_int32), ("bits", c_int32), ("world", c_int32),
                ("update_magnitude", c_int32), ("refresh_mask", c_int32), ("update_scale", c_int32),
                ("magnitude", c_void_p), ("stage_mean", c_void_p), ("C", c_int64), ("t_mag", c_int64), ("k", c_int64), ("t_q", c_int64),
                ("mask", c_void_p), ("chan_absmax", c_void_p), ("chan_absmax_stride", c_int64), ("scale", c_void_p),
                ("bump_i32_a", c_void_p), ("bump_i32_b", c_void_p), ("bump_i64_a", c_void_p), ("bump_i64_b", c_void_p),
                ("t_mag_dev", c_void_p), ("t_q_dev", c_void_p), ("gathered", c_void_p), ("elide_mask_out", c_void_p), ("stream", c_void_p)]


class SiteFwdArgs(ctypes.Structure):
    """`qs_site_fwd_args` of include/qsparse_hip.h"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("flags", c_int32), ("imgdt", c_int32), ("world", c_int32),
                ("x", c_void_p), ("y", c_void_p), ("gate_out", c_void_p), ("t_mag", c_int64), ("k", c_int64), ("t_q", c_int64),
                ("image_out", c_void_p), ("gathered", c_void_p), ("xback_out", c_void_p), ("decimal", c_void_p), ("stream", c_void_p)]


class QuantizeStepArgs(ctypes.Structure):
    """`qs_quantize_step_args` of include/qsparse_hip.h"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("lines", c_int32), ("xdt", c_int32), ("ydt", c_int32), ("bits", c_int32),
                ("pre_relu", c_int32), ("update", c_int32), ("saturate", c_int32), ("code_lo", c_int32), ("code_hi", c_int32),
                ("imgdt", c_int32), ("x", c_void_p), ("y", c_void_p), ("gate_out", c_void_p), ("amax_lines", c_void_p), ("scale", c_void_p),
                ("numel", c_int64), ("t", c_int64), ("t_dev", c_void_p), ("n_updates", c_void_p), ("xback_out", c_void_p),
                ("image_out", c_void_p), ("stream", c_void_p)]


class SiteBwdArgs(ctypes.Structure):
    """`qs_site_bwd_args` of include/qsparse_hip.h (size-prefixed: `struct_size` is what THIS binding knows)"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("flags", c_int32), ("gdt", c_int32), ("g2dt", c_int32),
                ("g", c_void_p), ("gate", c_void_p), ("gx", c_void_p), ("lo_mul", c_float), ("hi_mul", c_float),
                ("g2", c_void_p), ("decimal", c_void_p), ("stream", c_void_p),
                ("g3", c_void_p), ("gx_image", c_void_p), ("gx_image_dt", c_int32), ("reserved0", c_int32),
                ("act_x", c_void_p), ("act_x_kind", c_int32), ("reserved1", c_int32)]         # v26


class SteReluBwdArgs(ctypes.Structure):
    """`qs_ste_relu_bwd_args` of include/qsparse_hip.h"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("gdt", c_int32), ("xdt", c_int32), ("g2dt", c_int32),
                ("g", c_void_p), ("x", c_void_p), ("gate", c_void_p), ("gx", c_void_p), ("step", c_void_p), ("nstep", c_int64),
                ("step_host", c_float), ("step_is_decimal", c_int32), ("lo_mul", c_float), ("hi_mul", c_float),
                ("chan_mask", c_void_p), ("outer", c_int64), ("C", c_int64), ("inner", c_int64),
                ("elide_masked", c_int32), ("act", c_int32), ("g2", c_void_p), ("stream", c_void_p),
                ("g3", c_void_p), ("gx_image", c_void_p), ("gx_image_dt", c_int32), ("reserved0", c_int32),
                ("act_x", c_void_p), ("act_x_kind", c_int32), ("reserved1", c_int32)]         # v26


class MultiRow(ctypes.Structure):
    """`qs_multi_row` of include/qsparse_hip.h, field for field"""
    _fields_ = [("x", c_void_p), ("scale", c_void_p), ("amax", c_void_p), ("decimal", c_void_p), ("backup", c_void_p),
                ("t_dev", c_void_p), ("bump", c_void_p), ("numel", c_int64), ("y_off", c_int64), ("outer", c_int64),
                ("inner", c_int64), ("C", c_int32), ("train", c_int32), ("is_decimal", c_int32), ("t_offset", c_int32),
                ("code_lo", c_int32), ("code_hi", c_int32), ("denom", c_float),
                ("mask", c_void_p), ("mask_inner", c_int64), ("mask_C", c_int32), ("kind", c_int32),
                ("prune_n_updates", c_void_p), ("prune_t", c_void_p), ("magnitude", c_void_p), ("mag_backup", c_void_p),
                ("refresh", c_int32), ("select_k", ctypes.c_uint32), ("importance", c_void_p), ("select_state", c_void_p),
                ("mask_backup", c_void_p),
                ("row_splits", c_int32),
                ("absmax_block0", c_int32), ("absmax_blocks", c_int32), ("quant_block0", c_int32), ("chan0", c_int32),
                ("hist_block0", c_int32), ("hist_blocks", c_int32), ("reserved1", c_int32)]


SITE_LIVE, SITE_REFRESH, SITE_PRE_RELU, SITE_ELIDE, SITE_NO_MASK, SITE_STATS_DONE, SITE_SCALE_ONLY, SITE_NO_QUANT = 1, 2, 4, 8, 16, 32, 64, 128
QSTEP_APPLY, QSTEP_ALL, QSTEP_ABSMAX, QSTEP_FINISH = 0, 1, 2, 3

_lib = None

# Mask-aware traffic elision (set through set_qsparse_options(elide_pruned=...), see qs_elementwise.h):
#   "forward" (default)  the quantizer forward of a prune -> quantize site skips the loads of pruned channels where that saves
#                        traffic AND is exact for every input: an NCHW activation (a pruned channel is a whole row), a forward
#                        that records no ReLU gate, on a step whose statistics pass has seen the input -- the select then
#                        writes an ELISION MASK (1 kept, 0 pruned and finite, 2 pruned with a NaN / Inf in the channel) and
#                        only the channels marked 0 are skipped: x * 0 is a zero there whatever x is, and a NaN / Inf on a
#                        pruned channel still comes out as the reference's f32(INT_MIN) * s (quirk B15).  Everything else --
#                        steps without statistics (evaluation, idle steps), channels_last (a pruned channel is a 2-byte
#                        column: nothing to skip), gate-recording forwards (they load every element anyway), backwards, mask
#                        applies -- runs the loading arithmetic
#   "all"                every kernel that carries a channel mask, the backward and mask-apply kernels and the forwards of
#                        steps without statistics included: +0.0 where the reference has -0.0, f32(0)*s where it has
#                        f32(INT_MIN)*s for a NaN / Inf on a pruned channel (opt-in)
#   "off"                every element is loaded
elide_mode = "forward"


def _elide_fwd(channels_last: bool = False, records_gate: bool = False, exact: bool = False) -> int:
    """`exact`: the kernel is handed the select's elision mask of THIS step instead of the mask"""
    if elide_mode == "forward":
        return int(exact and not (channels_last or records_gate))
    return int(elide_mode == "all")


def _elide_all() -> int:
    return int(elide_mode == "all")


# ---- optional per-kernel timing with HIP events (bench.py) -----------------------------------------
# torch.cuda.Event records on torch's current stream, which is the stream every kernel here is
# launched on, so (start, end) pairs bracket exactly one launch.
_event_log = None
_event_filter = None


def start_event_log(only=None):
    """only: optional collection of kernel names to time (every event pair costs a few us of stream time)."""
    global _event_log, _event_filter
    _event_log = {}
    _event_filter = set(only) if only else None


def take_event_pairs():
    """hand over the recorded (start, end) event pairs without synchronising; stops logging."""
    global _event_log
    log, _event_log = _event_log, None
    return {k: [(a, b) for a, b, _ in v] for k, v in (log or {}).items()}


def stop_event_log(with_bytes: bool = False):
    """returns {kernel name: [ms, ...]} (synchronises); `with_bytes`: {kernel name: [(ms, algorithmic bytes), ...]} --
    the bytes each launch has to move by the reference's dtype contract (see `_timed`), for roofline accounting."""
    global _event_log
    log, _event_log = _event_log, None
    torch.cuda.synchronize()
    if with_bytes:
        return {k: [(a.elapsed_time(b), nb) for a, b, nb in v] for k, v in (log or {}).items()}
    return {k: [a.elapsed_time(b) for a, b, _ in v] for k, v in (log or {}).items()}


class _timed:
    """`operands`: the data tensors the launch reads or writes (or a byte count): its algorithmic bytes are every one of
    them moved ONCE in its own dtype (dense: the mask-aware elision of NCHW launches is not subtracted); nothing for the
    C-sized launches, which are latency, not bandwidth.  Only evaluated while an event log is being recorded."""
    __slots__ = ("name", "a", "operands")

    def __init__(self, name, *operands):
        self.name, self.operands = name, operands

    def __enter__(self):
        self.a = None
        if _event_log is not None and (_event_filter is None or self.name in _event_filter):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            nbytes = sum(o if isinstance(o, int) else o.numel() * o.element_size() for o in self.operands if o is not None)
            _event_log.setdefault(self.name, []).append((self.a, b, nbytes))
        return False


def load(path: Optional[str] = None):
    """dlopen the library and declare every prototype (no GPU needed for this)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or lib_path()
    if not os.path.exists(p):
        raise QsparseHipError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). GPU tensors cannot be processed without it.")
    lib = ctypes.CDLL(p)
    # Up to v24 symbol names kept their names while their argument lists changed: a stale library would be called with shifted
    # arguments (ints in pointer slots), so the version is part of loading, not of the tests.  From v25 on the positional
    # prototypes are frozen and new operands arrive through size-prefixed descriptors (include/qsparse_hip.h, "ABI
    # compatibility"): ANY library that still honours the prototypes of the version this binding was written against and is at
    # least that new will do -- qs_abi_floor() <= ABI_VERSION <= qs_version().
    lib.qs_version.restype, lib.qs_version.argtypes = c_int, []
    found = lib.qs_version()
    floor = found
    if hasattr(lib, "qs_abi_floor"):
        lib.qs_abi_floor.restype, lib.qs_abi_floor.argtypes = c_int, []
        floor = lib.qs_abi_floor()
    if not abi_compatible(found, floor):
        raise QsparseHipError(
            f"{p} is ABI v{found} (prototypes frozen since v{floor}), this package needs v{ABI_VERSION}: rebuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'`")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == a symbol this binding needs is missing
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def abi_compatible(found: int, floor: int, needed: Optional[int] = None) -> bool:
    """whether a library reporting `qs_version() == found` and `qs_abi_floor() == floor` serves a binding written against ABI
    `needed`: new enough to have every symbol and field the binding uses, and still honouring that version's prototypes"""
    needed = ABI_VERSION if needed is None else needed
    return floor <= needed <= found


def available() -> bool:
    try:
        load()
        return True
    except (QsparseHipError, OSError, AttributeError):
        return False


def _check(status: int, what: str):
    if status != 0:
        msg = load().qs_status_string(status)
        raise QsparseHipError(f"{what}: {msg.decode() if msg else status} (status {status})")


HIP_DTYPES = frozenset((torch.float32, torch.bfloat16, torch.float16))


def on_hip(t: torch.Tensor) -> bool:
    """whether the operators on this tensor run as HIP kernels: a GPU tensor of a dtype the kernels are written for.  A GPU
    tensor of another dtype (float64, integers -- the reference is dtype-agnostic ATen, quantize.py:109-117, sparse.py:116)
    evaluates the package's own ATen expression ON THE DEVICE: the code the CPU path runs, which restates the reference line
    by line (type promotion included: a float64 input is divided in float64 and still comes out as float32).  Never the
    oracle, never a host round trip."""
    return t.is_cuda and t.dtype in HIP_DTYPES


def dense_any_order(x: torch.Tensor) -> bool:
    """non-overlapping and dense: some permutation of the dims is contiguous"""
    expect = 1
    for st, size in sorted((st, size) for st, size in zip(x.stride(), x.shape) if size != 1):
        if st != expect:
            return False
        expect *= size
    return True


class _Relayout(torch.autograd.Function):
    """the same values in another dense layout (strides given); the gradient passes through as it comes"""

    @staticmethod
    def forward(ctx, y, strides):
        out = torch.empty_strided(y.shape, strides, dtype=y.dtype, device=y.device)
        out.copy_(y)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def _same_layout(a, b, shape) -> bool:
    return all(sa == sb or n == 1 for sa, sb, n in zip(a, b, shape))        # (the stride of an extent-1 dim means nothing)


def laid_out_like(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """`y` in the layout the reference's element-wise chain gives the result for the input `x` (`x * mask`, sparse.py:116;
    `(x / s).round() * s`, quantize.py:109-117): TensorIterator lays an output out densely in its input's stride order.  The kernels
    address contiguous and channels_last tensors where they lie and return that layout by construction; a tensor in any other
    order (a transpose, a permute, a channels_last tensor whose channel index is not 1) went through its contiguous copy, and the
    result is copied once more into the order the reference would have returned -- which matters to whatever reduces it next."""
    if _same_layout(y.stride(), x.stride(), x.shape):
        return y
    want = x.stride() if dense_any_order(x) else torch.empty_like(x).stride()
    if _same_layout(y.stride(), want, x.shape):
        return y
    return _Relayout.apply(y, tuple(want))


def keeps_layout(forward):
    """decorator of the operators' `forward(self, x, ...)`: the result of a GPU tensor that is not contiguous comes back in the
    layout the reference returns for it (`laid_out_like`); one `is_contiguous()` for every other input"""
    import functools

    @functools.wraps(forward)
    def wrapped(self, x, *args, **kwargs):
        y = forward(self, x, *args, **kwargs)
        if (isinstance(x, torch.Tensor) and not x.is_contiguous() and x.is_cuda and isinstance(y, torch.Tensor) and y is not x
                and y.stride() != x.stride() and y.shape == x.shape):
            return laid_out_like(y, x)
        return y
    return wrapped


def true_div(a: torch.Tensor, n) -> torch.Tensor:
    """``a / n`` for a Python number ``n`` as ATen's CPU kernels evaluate it: a correctly rounded division.  On the device ATen
    turns a division by a host scalar into a multiplication by its reciprocal (BinaryDivTrueKernel.cu: "compute a *
    reciprocal(b)"), one more rounding -- which the GPU tensors that take the ATen expression (`on_hip`) must not see: there the
    divisor is made a device tensor, for which ATen divides."""
    if not a.is_cuda:
        return a / n
    return a / torch.full((), n, dtype=a.dtype if a.is_floating_point() else torch.get_default_dtype(), device=a.device)


def refuse_capture(t: torch.Tensor, what: str):
    """the ATen-on-device route of a training step keeps its running-mean counts on the host, like the CPU path: captured into
    a hipGraph they would be replayed as constants -- refuse loudly instead"""
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        raise QsparseHipError(f"{what} of a {t.dtype} GPU tensor evaluates the package's ATen expression with host-side counters: it "
                              "cannot be captured into a hipGraph (float32 / bfloat16 / float16 tensors take the graph-safe kernels)")


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise QsparseHipError(f"dtype {t.dtype} is not supported by the HIP path (float32/bfloat16/float16 only)")


def _ptr(t: Optional[torch.Tensor]):
    """raw device address; a host tensor here would fault the GPU, so refuse it loudly."""
    if t is None:
        return None
    if not t.is_cuda:
        raise QsparseHipError(f"tensor of shape {tuple(t.shape)} lives on {t.device}, expected a GPU tensor: move the "
                              "module / argument to the input's device")
    return t.data_ptr()


try:   # raw hipStream_t of torch's current stream without building a Stream object (20x cheaper per call)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:   # pragma: no cover
    _raw_stream = None


def _stream(t: torch.Tensor):
    """hipStream_t to launch on.  Kernels are launched in the CURRENT device context, so the tensor must live on
    the current device (one process per GPU is the supported topology); anything else fails loudly here instead
    of faulting on the device."""
    cur = torch.cuda.current_device()
    idx = t.device.index if t.device.index is not None else cur
    if idx != cur:
        raise QsparseHipError(f"tensor on cuda:{idx} but the current device is cuda:{cur}: call "
                              "torch.cuda.set_device (one process per GPU) or wrap the call in torch.cuda.device(...)")
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(t.device).cuda_stream


def dense(t: torch.Tensor) -> torch.Tensor:
    """contiguous and 16-byte aligned (the ABI's requirement for data tensors)."""
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def mem_view(x: torch.Tensor, channel_index: int):
    """(xm, channel index in xm, like): the tensor the kernels address.  A dense channels_last activation (4-d NHWC or
    5-d NDHWC in memory) whose channel index is 1 (or absent) is used in place through its memory-order view -- channel
    dim innermost, no copy -- and outputs are allocated ``empty_like(like)``, i.e. channels_last again, as the reference's
    element-wise torch ops would return them.  Everything else goes through a contiguous copy, as before."""
    if x.dim() in (4, 5) and channel_index in (-1, 1) and not x.is_contiguous() and x.data_ptr() % 16 == 0:
        fmt = torch.channels_last if x.dim() == 4 else torch.channels_last_3d
        if x.is_contiguous(memory_format=fmt):
            perm = (0, 2, 3, 1) if x.dim() == 4 else (0, 2, 3, 4, 1)
            return x.permute(perm), (-1 if channel_index < 0 else x.dim() - 1), x
    xd = dense(x)
    return xd, channel_index, xd


def split3(shape, channel_index: int):
    """[outer, C, inner] factorisation around `channel_index` (negative: tensor-wise)."""
    numel = 1
    for s in shape:
        numel *= s
    if channel_index < 0:
        return 1, 1, max(numel, 1), numel
    outer = 1
    for s in shape[:channel_index]:
        outer *= s
    inner = 1
    for s in shape[channel_index + 1:]:
        inner *= s
    return outer, shape[channel_index], inner, numel


def _chan_mask_bytes(chan_mask: Optional[torch.Tensor], C: int):
    """flat uint8 view of a per-channel bool mask; a channel count that does not match the tensor raises the
    RuntimeError the reference's ``x * mask`` broadcast raises (PruneLayer.forward docstring, sparse.py:222)"""
    if chan_mask is None:
        return None
    cm = chan_mask.detach().contiguous().view(torch.uint8).view(-1)
    if cm.numel() != C:
        raise RuntimeError(f"The size of tensor a ({C}) must match the size of tensor b ({cm.numel()}) at non-singleton "
                           "dimension (channel mask vs input)")
    return cm


def _f32param(p, device):
    """device fp32 array or (None, host float) for scalars given as Python numbers."""
    if isinstance(p, torch.Tensor):
        q = p.detach().to(device=device, dtype=torch.float32).contiguous().view(-1)
        return q, q.numel(), 0.0
    return None, 1, float(p)


# ----------------------------------------------------------------------------------------------
# quantizers
# ----------------------------------------------------------------------------------------------
_gate_sink = threading.local()


def note_gate(bits: torch.Tensor):
    """a forward just recorded a ReLU gate: whoever asked to be told (fused.py::_with_owned_relu) gets the bitmap"""
    cell = getattr(_gate_sink, "cell", None)
    if cell is not None:
        cell["bits"] = bits


# development aid for the byte accounting of bench.py / tools/profile_config.py: the event log (above) takes the fine-grained
# entry points, which make no autocast image; with this dict set, the composite site step adds what the image route moves
# DIFFERENTLY from the fine-grained one (forward: the image written; backward: the 2-byte gradient read, the float32 one not)
image_byte_delta = None         # None, or {"apply_fwd": bytes, "apply_bwd": bytes}

ACT_RELU, ACT_HARDTANH, ACT_LEAKY = 1, 2, 3
_act_handles = {}
_act_specs = {}          # handle -> (kind, a, b)


def activation(kind: int, a: float = 0.0, b: float = 0.0) -> int:
    """handle of a folded activation (qs_activation): what the `pre_relu` arguments below take besides False / True"""
    key = (kind, float(a), float(b))
    h = _act_handles.get(key)
    if h == 0:
        raise QsparseHipError(f"qs_activation{key}: the library could not intern this descriptor earlier in this process")
    if h is None:
        h = load().qs_activation(kind, float(a), float(b))
        if h < 0:
            _check(h, "qs_activation")
        _act_handles[key] = h
        _act_specs[h] = key
    return h


def try_activation(kind: int, a: float = 0.0, b: float = 0.0) -> int:
    """`activation`, or 0 -- "not foldable" -- where the library cannot intern the descriptor (its 256-entry table is full,
    NaN parameters): the site then runs the activation module by itself, q(p(act(x))), instead of failing the forward"""
    if _act_handles.get((kind, float(a), float(b))) == 0:
        return 0
    try:
        return activation(kind, a, b)
    except QsparseHipError:
        _act_handles[(kind, float(a), float(b))] = 0
        return 0


def act_spec(handle: int):
    """(kind, a, b) of a handle made by `activation` (1: nn.ReLU)"""
    if handle == 1:
        return (ACT_RELU, 0.0, 0.0)
    return _act_specs[handle]


class _ActivationFromOutput(torch.autograd.Function):
    """an out-of-place activation whose backward keeps its OUTPUT instead of its input (threshold_backward /
    hardtanh_backward / leaky_relu_backward all decide on the side of the rectified value just as well): for an input whose
    storage is about to be overwritten, see `act_torch`"""

    @staticmethod
    def forward(ctx, h, act):
        ctx.act = act
        y = _act_aten(act, h)
        # a clamp whose bounds the dtype cannot represent saturates at the ROUNDED bound, which may lie inside (a, b): the output
        # no longer tells a clamped element from one that sat there -- keep the gate of the INPUT (one byte per element) instead
        ctx.from_input = act_spec(act)[0] == ACT_HARDTANH and not bounds_representable(act, h.dtype)
        ctx.save_for_backward(act_gate_of(act, h) if ctx.from_input else y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        kind, slope, _ = act_spec(ctx.act)
        closed = g * slope if kind == ACT_LEAKY else torch.zeros((), dtype=g.dtype, device=g.device)
        return torch.where(y if ctx.from_input else act_gate_of(ctx.act, y), g, closed), None


_representable = {}


def bounds_representable(act: int, dtype: torch.dtype) -> bool:
    """whether both bounds of a folded clamp (nn.Hardtanh / nn.ReLU6) are values of `dtype`: ATen clamps to the bounds ROUNDED to
    the tensor's dtype (0.1 -> 0.10009765625 in bfloat16) while hardtanh_backward compares the float bounds -- with such bounds
    the gate cannot be read off the rectified tensor (see `act_gate_of`)"""
    kind, a, b = act_spec(act)
    if kind != ACT_HARDTANH or dtype == torch.float32:
        return True
    key = (a, b, dtype)
    r = _representable.get(key)
    if r is None:
        t = torch.tensor([a, b], dtype=torch.float64).to(dtype).to(torch.float64)
        r = _representable[key] = bool(t[0].item() == a and t[1].item() == b)
    return r


def _act_aten(act: int, x: torch.Tensor) -> torch.Tensor:
    kind, a, b = act_spec(act)
    if kind == ACT_RELU:
        return torch.relu(x)
    if kind == ACT_HARDTANH:
        return torch.nn.functional.hardtanh(x, a, b)
    return torch.nn.functional.leaky_relu(x, a)


def act_torch(pre_relu, x: torch.Tensor) -> torch.Tensor:
    """the folded activation as an ATen expression (out of place): the routes that have to materialise it after all"""
    cell = owned_relu_cell()
    if (cell is not None and cell.get("x") is not None and torch.is_grad_enabled() and x.requires_grad
            and x.data_ptr() == cell["x"].data_ptr() and x.shape == cell["x"].shape):
        # an owned in-place activation is still pending on this very storage (fused.py::_with_owned_relu deferred it to the site's
        # apply kernel; the ATen in-place pass runs after the site when no kernel did it).  ATen's own out-of-place activation
        # would keep this raw alias for its backward, and the pending pass would then overwrite it ("modified by an inplace
        # operation"): record a node that keeps its output instead.
        return _ActivationFromOutput.apply(x, _act(pre_relu))
    return _act_aten(_act(pre_relu), x)


def act_torch_(pre_relu, x: torch.Tensor) -> torch.Tensor:
    """... and in place"""
    kind, a, b = act_spec(_act(pre_relu))
    if kind == ACT_RELU:
        return x.relu_()
    if kind == ACT_HARDTANH:
        return torch.nn.functional.hardtanh_(x, a, b)
    return torch.nn.functional.leaky_relu_(x, a)


def act_gate_of(pre_relu, h: torch.Tensor) -> torch.Tensor:
    """the backward's gate (True: the gradient passes unchanged) from the activation's INPUT or -- the rectifiers and a leaky
    ReLU with a positive slope map both to the same side -- its output (a clamp: only with bounds its dtype represents exactly,
    `bounds_representable`; the callers check)"""
    kind, a, b = act_spec(_act(pre_relu))
    if kind == ACT_HARDTANH:
        return (h > a) & (h < b)
    if kind == ACT_LEAKY:
        return h > 0
    return ~(h <= 0)


def mean_flags(take_abs: bool, pre_relu=False, l0: bool = False) -> int:
    """flags of qs_mean_dim / qs_mean_dim_cl; a folded activation other than nn.ReLU rides as QS_MEAN_ACT(handle)"""
    h = _act(pre_relu)
    return (MEAN_ABS if take_abs else 0) | (MEAN_L0 if l0 else 0) | ((MEAN_RELU | ((h << 8) if h > 1 else 0)) if h else 0)


def _act(pre_relu) -> int:
    """`pre_relu` as the ABI takes it: 0 (no folded activation), 1 (nn.ReLU; also True) or a qs_activation() handle"""
    return int(pre_relu) if pre_relu else 0


def owned_relu_cell():
    """the cell of the owned in-place ReLU whose site is being evaluated on this thread (fused.py::_with_owned_relu), or None.
    cell["defer"]: the ReLU has NOT been applied to x yet -- the site's apply kernel may write relu(x) back itself (xback_out of
    qs_quant_scaler_fwd) and set cell["done"]; whoever does not, leaves it to the caller's ATen pass."""
    cell = getattr(_gate_sink, "cell", None)
    return cell if (cell is not None and cell.get("defer") and not cell.get("done")) else None


def unpack_gate(bits: torch.Tensor, shape, strides) -> torch.Tensor:
    """the recorded gate as a bool tensor laid out like the (dense) activation it belongs to: True where the ReLU lets the
    gradient through, !(x <= 0) (a torch composition for the rare route that needs the gate outside the backward kernels)"""
    numel = 1
    for d in shape:
        numel *= d
    shifts = torch.arange(8, device=bits.device, dtype=torch.uint8)
    flat = ((bits.view(-1, 1) >> shifts) & 1).bool().view(-1)[:numel]
    return flat.as_strided(tuple(shape), tuple(strides))


class ReluGate:
    """the gate of a folded ReLU as the forward kernel recorded it: one bit per element in the MEMORY order the kernel
    addressed (`channels_last`: the activation was used in place through its NHWC view), plus what the backward needs to
    know about the ReLU's input without keeping it: shape, dtype, layout."""
    __slots__ = ("bits", "shape", "dtype", "channels_last")

    def __init__(self, bits: torch.Tensor, like: torch.Tensor, channels_last: bool):
        self.bits, self.shape, self.dtype, self.channels_last = bits, like.shape, like.dtype, channels_last
        note_gate(bits)

    @classmethod
    def from_saved(cls, bits: torch.Tensor, shape, dtype, channels_last: bool) -> "ReluGate":
        """rebuild the gate in a backward from the bitmap autograd saved and the description kept on ctx"""
        g = cls.__new__(cls)
        g.bits, g.shape, g.dtype, g.channels_last = bits, shape, dtype, channels_last
        return g


def quant_fwd(kind: str, x: torch.Tensor, param, channel_index: int, qdtype: torch.dtype,
              chan_mask: Optional[torch.Tensor] = None, mask_channel_index: Optional[int] = None,
              want_codes: bool = False, out_dtype: torch.dtype = torch.float32, saturate=None, pre_relu: bool = False,
              want_gate: bool = False, xback: Optional[dict] = None, elision_mask: bool = False):
    """kind in {'scaler','decimal'}; returns (y, codes|None).  pre_relu: quantise max(x, 0) (folded nn.ReLU).
    elision_mask: `chan_mask` is the uint8 elision mask `pq_select(..., elide_mask=)` wrote from THIS x's statistics (1 kept,
    0 pruned and finite, 2 pruned with a NaN / Inf) -- skipping the loads of its 0 channels is exact (see `elide_mode`).
    want_gate (with pre_relu): returns (y, codes|None, ReluGate) -- the ReLU's gate as one bit per element, recorded by
    the same pass, for `ste_relu_bwd(gate=...)`.
    xback (with want_gate): the cell of an owned in-place ReLU (fused.py::_with_owned_relu) -- when this launch is one that can
    write relu(x) back into x's own storage (qs_quant_image_ok) it does, and says so in xback["done"]."""
    lib = load()
    pt, n, host = _f32param(param, x.device)
    ci = channel_index if n > 1 else (mask_channel_index if chan_mask is not None else -1)
    x, ci, like = mem_view(x, ci if ci is not None else -1)
    outer, C, inner, numel = split3(x.shape, ci)
    gate = None
    if want_gate:
        if not pre_relu:
            raise ValueError("want_gate records the gate of a folded ReLU: pre_relu must be set")
        gate = ReluGate(torch.empty((numel + 7) // 8, dtype=torch.uint8, device=x.device), like, x is not like)
    if numel == 0:
        y = torch.empty_like(like, dtype=out_dtype)
        return (y, None, gate) if want_gate else (y, None)
    y = torch.empty_like(like, dtype=out_dtype)
    codes = torch.empty_like(like, dtype=torch.int32) if want_codes else None
    cm = _chan_mask_bytes(chan_mask, C)
    sat, lo, hi = (0, 0, 0) if saturate is None else (1, int(saturate[0]), int(saturate[1]))
    fn = lib.qs_quant_scaler_fwd if kind == "scaler" else lib.qs_quant_decimal_fwd
    xb = None
    if (xback is not None and gate is not None and codes is None and out_dtype == torch.float32 and not xback.get("done")
            and lib.qs_quant_image_ok(outer, C, inner, int(n > 1), int(cm is not None), int(cm is None or cm.data_ptr() % 8 == 0), dt(x))):
        xb = x                 # relu(x) lands in x's own storage (x IS the in-place ReLU's tensor, addressed in memory order)
    with _timed(f"quant_{kind}_fwd" + ("+mask" if cm is not None else ""), x, y, codes, gate.bits if gate is not None else None, xb):
        st = fn(_ptr(x), _ptr(y), _ptr(codes), _ptr(pt), n, host, _ptr(cm), outer, C, inner, dt(x), _DT[out_dtype],
                _DT[qdtype], sat, lo, hi, _act(pre_relu), _elide_fwd(x is not like, gate is not None, elision_mask) if cm is not None else 0,
                _ptr(gate.bits) if gate is not None else None, None, 0, _ptr(xb), _stream(x))
    _check(st, f"qs_quant_{kind}_fwd")
    if xb is not None:
        xback["done"] = True
    return (y, codes, gate) if want_gate else (y, codes)


def quant_line_fwd(x: torch.Tensor, lines: torch.Tensor, bits: int, channel_index: int, float_zero_point: bool,
                   want_codes: bool = False):
    """returns y, or (y, codes) with `want_codes`: the int32 level index of every element (see qs_quant_line_fwd)"""
    lib = load()
    ln = lines.detach().to(device=x.device, dtype=torch.float32).contiguous().view(-1, 2)
    n = ln.shape[0]
    x, ci, like = mem_view(x, channel_index if n > 1 else -1)
    outer, C, inner, numel = split3(x.shape, ci)
    y = torch.empty_like(like, dtype=torch.float32)
    codes = torch.empty_like(like, dtype=torch.int32) if want_codes else None
    if numel == 0:
        return (y, codes) if want_codes else y
    with _timed("quant_line_fwd", x, y, codes):
        st = lib.qs_quant_line_fwd(_ptr(x), _ptr(y), _ptr(codes), _ptr(ln), n, int(bits), int(bool(float_zero_point)), outer, C,
                                   inner, dt(x), F32, _stream(x))
    _check(st, "qs_quant_line_fwd")
    return (y, codes) if want_codes else y


def ste_bwd(g: torch.Tensor, step, step_is_decimal: bool, channel_index: int, lo_mul: float, hi_mul: float,
            passthrough: bool, out_dtype: torch.dtype, chan_mask: Optional[torch.Tensor] = None,
            mask_channel_index: Optional[int] = None):
    lib = load()
    pt, n, host = _f32param(step, g.device)
    ci = channel_index if n > 1 else (mask_channel_index if chan_mask is not None else -1)
    g, ci, like = mem_view(g, ci if ci is not None else -1)
    outer, C, inner, numel = split3(g.shape, ci)
    gx = torch.empty_like(like, dtype=out_dtype)
    if numel == 0:
        return gx
    cm = _chan_mask_bytes(chan_mask, C)
    with _timed("quant_ste_bwd" + ("+mask" if cm is not None else ""), g, gx):
        st = lib.qs_quant_ste_bwd(_ptr(g), _ptr(gx), _ptr(pt), n, host, int(bool(step_is_decimal)), float(lo_mul),
                                  float(hi_mul), int(bool(passthrough)), _ptr(cm), outer, C, inner, dt(g),
                                  _DT[out_dtype], _elide_all() if cm is not None else 0, _stream(g))
    _check(st, "qs_quant_ste_bwd")
    return gx


def _like_grad(gx: torch.Tensor, ref: torch.Tensor, act) -> torch.Tensor:
    """A site that folds the IDENTITY (fused.identity_fold_handle: nothing foldable in front, e.g. an nn.GELU) hands its gradient on in
    GRAD_OUTPUT's layout -- what the reference's chain (`grad_output.clamp_()` in place, quantize.py:72-76, 126-130; `* mask`,
    sparse.py:263) and this package's ungated route (`ste_bwd`) return: the op above such a site is an arbitrary ATen backward, and
    ATen's fp16 element-wise backwards are not bit-stable across operand layouts (gelu_backward: tools/probes/probe_gelu_tail.py).
    The gate-reading kernels compute in the layout the forward addressed; where the incoming gradient is dense in another one (a
    contiguous gradient for a channels_last activation) the result is copied into that layout.  Sites behind a FOLDED activation keep
    the forward's layout: what is above them is the layer that produced x, which wants its gradient laid out like x (a contiguous
    gradient from the classifier's pooling would otherwise turn the whole last stage of a channels_last ResNet-50 to NCHW gradients:
    +1.0 ms per step, measured)."""
    h = _act(act)
    if h <= 1 or _act_specs.get(h) != (ACT_LEAKY, 1.0, 0.0):
        return gx
    if gx.shape != ref.shape or gx.stride() == ref.stride() or not dense_any_order(ref):
        return gx
    out = torch.empty_strided(ref.shape, ref.stride(), dtype=gx.dtype, device=gx.device)
    out.copy_(gx)
    return out


def ste_relu_bwd(g: Optional[torch.Tensor], x: Optional[torch.Tensor], step, step_is_decimal: bool, lo_mul: float, hi_mul: float,
                 chan_mask: Optional[torch.Tensor], mask_channel_index: int = 1, gate: Optional[ReluGate] = None,
                 g2: Optional[torch.Tensor] = None, act=1, g3: Optional[torch.Tensor] = None,
                 gx_image_dtype: Optional[torch.dtype] = None):
    """gx = (x <= 0 ? 0 : clamp(g) * mask) in x's dtype: STE backward + channel mask + folded-ReLU gate.  With `gate` (the
    bitmap `quant_fwd(want_gate=True)` recorded) x is not needed.  `g2` (with `gate`; bf16 / fp16): a second gradient that
    is added to the float32 `g` in float32 before the clamp; `g` may then be None.
    The riders of the all-float32 form (with `gate`, a float32 x, through qs_quant_ste_relu_bwd_v): `g3`, a third stream of g2's
    dtype added between g and g2; `gx_image_dtype`: also write RNE(gx) in that dtype -- the call then returns (gx, image)."""
    lib = load()
    ref = g if g is not None else g2
    pt, n, host = _f32param(step, ref.device)
    ci = mask_channel_index if chan_mask is not None else -1
    if gate is not None:
        assert tuple(ref.shape) == tuple(gate.shape) and (g2 is None or g is None or g.dtype == torch.float32)

        def as_mem(t):
            if t is None:
                return None, ci
            if gate.channels_last:         # the bitmap follows the NHWC memory order the forward addressed
                fmt = torch.channels_last if t.dim() == 4 else torch.channels_last_3d
                tcl = t.contiguous(memory_format=fmt)
                if tcl.data_ptr() % 16:
                    tcl = tcl.clone(memory_format=torch.preserve_format)
                perm = (0, 2, 3, 1) if t.dim() == 4 else (0, 2, 3, 4, 1)
                return tcl.permute(perm), (-1 if ci < 0 else perm.index(ci))
            return dense(t), ci

        gm, ci_mem = as_mem(g)
        g2m, ci2 = as_mem(g2)
        g3m, _ = as_mem(g3)
        refm = gm if gm is not None else g2m
        ci_mem = ci_mem if gm is not None else ci2
        if gate.channels_last:
            gx = torch.empty(gate.shape, dtype=gate.dtype, device=ref.device,
                             memory_format=torch.channels_last if ref.dim() == 4 else torch.channels_last_3d)
        else:
            gx = torch.empty(gate.shape, dtype=gate.dtype, device=ref.device)
        outer, C, inner, numel = split3(refm.shape, ci_mem)
        if numel == 0:
            return gx
        cm = _chan_mask_bytes(chan_mask, C)
        if g3m is not None or gx_image_dtype is not None:
            assert gate.dtype == torch.float32 and (gm is None or gm.dtype == torch.float32), "the riders belong to the all-float32 form"
            gimg = torch.empty_like(gx, dtype=gx_image_dtype) if gx_image_dtype is not None else None
            a = SteReluBwdArgs()
            a.struct_size = ctypes.sizeof(SteReluBwdArgs)
            a.gdt, a.xdt, a.g2dt = F32, F32, (0 if g2m is None else dt(g2m))
            a.g, a.gate, a.gx, a.step, a.nstep, a.step_host = _ptr(gm), _ptr(gate.bits), _ptr(gx), _ptr(pt), n, host
            a.step_is_decimal, a.lo_mul, a.hi_mul, a.chan_mask = int(bool(step_is_decimal)), float(lo_mul), float(hi_mul), _ptr(cm)
            a.outer, a.C, a.inner, a.elide_masked, a.act = outer, C, inner, 0, _act(act) or 1
            a.g2, a.stream, a.g3 = _ptr(g2m), _stream(refm), _ptr(g3m)
            if gimg is not None:
                a.gx_image, a.gx_image_dt = _ptr(gimg), _DT[gx_image_dtype]
            with _timed("quant_ste_relu_bwd", gm, g2m, g3m, gate.bits, gx, gimg):
                st = lib.qs_quant_ste_relu_bwd_v(ctypes.byref(a))
            _check(st, "qs_quant_ste_relu_bwd_v")
            return (_like_grad(gx, ref, act), _like_grad(gimg, ref, act)) if gx_image_dtype is not None else _like_grad(gx, ref, act)
        with _timed("quant_ste_relu_bwd", gm, g2m, gate.bits, gx):
            st = lib.qs_quant_ste_relu_bwd(_ptr(gm), None, _ptr(gate.bits), _ptr(gx), _ptr(pt), n, host,
                                           int(bool(step_is_decimal)), float(lo_mul), float(hi_mul), _ptr(cm), outer, C, inner,
                                           F32 if gm is None else dt(gm), _DT[gate.dtype],
                                           _elide_all() if (cm is not None and g2m is None) else 0, _act(act) or 1, _ptr(g2m),
                                           0 if g2m is None else dt(g2m), _stream(refm))
        _check(st, "qs_quant_ste_relu_bwd")
        return _like_grad(gx, ref, act)
    assert g.shape == x.shape
    g_in = g
    xm, ci_mem, like = mem_view(x, ci)
    gm = None
    if like is x and xm is not x:        # x is addressed in place as channels_last: the gradient must share that layout
        gcl = g.contiguous(memory_format=torch.channels_last if x.dim() == 4 else torch.channels_last_3d)
        gv, _, glike = mem_view(gcl, ci)
        if glike is gcl and gv is not gcl:
            gm = gv
    if gm is None:                        # the ordinary case, and the fallback for a misaligned gradient
        xm = dense(x)
        ci_mem, like, gm = ci, xm, dense(g)
    g, x, ci = gm, xm, ci_mem
    outer, C, inner, numel = split3(g.shape, ci)
    gx = torch.empty_like(like, dtype=like.dtype)
    if numel == 0:
        return gx
    cm = _chan_mask_bytes(chan_mask, C)
    with _timed("quant_ste_relu_bwd", g, x, gx):
        st = lib.qs_quant_ste_relu_bwd(_ptr(g), _ptr(x), None, _ptr(gx), _ptr(pt), n, host, int(bool(step_is_decimal)),
                                       float(lo_mul), float(hi_mul), _ptr(cm), outer, C, inner, dt(g), dt(x),
                                       _elide_all() if cm is not None else 0, _act(act) or 1, None, 0, _stream(g))
    _check(st, "qs_quant_ste_relu_bwd")
    return _like_grad(gx, g_in, act)


# ----------------------------------------------------------------------------------------------
# statistics
# ----------------------------------------------------------------------------------------------
WS_REDUCE = 2


def reduce_workspace_bytes(channel_index: int, outer: int, C: int, inner: int) -> int:
    """bytes of scratch the two-stage per-channel reduction over few columns wants for this geometry (0: it never takes that route)"""
    if channel_index < 0 or inner >= 64 or outer < 256:
        return 0
    return int(load().qs_workspace_bytes(WS_REDUCE, C * inner))


def _reduce_workspace(x: torch.Tensor, channel_index: int, outer: int, C: int, inner: int):
    """scratch for the two-stage per-channel reduction over few columns (channels_last activations, 2-d inputs)"""
    nbytes = reduce_workspace_bytes(channel_index, outer, C, inner)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=x.device), nbytes


TENSOR_AMAX_LINES = 16   # partial accumulators of a tensor-wise abs-max, one 128-byte line each


def tensor_amax_accumulator(device) -> torch.Tensor:
    """zeroed accumulator of a tensor-wise abs-max: [16, 32] floats, partial maxima in [:, 0] (qs_absmax out_lines);
    qs_scale_update folds and re-zeroes it."""
    return torch.zeros(TENSOR_AMAX_LINES, AMAX_LINE_STRIDE, dtype=torch.float32, device=device)


def absmax(x: torch.Tensor, channel_index: int, accumulate_into: Optional[torch.Tensor] = None,
           pre_relu: bool = False) -> torch.Tensor:
    """max|x| per channel / over the tensor.  `accumulate_into`: zeroed persistent fp32 buffer that is
    max-accumulated instead of allocating + initialising a fresh one (one launch instead of two); for a tensor-wise
    maximum it may be a `tensor_amax_accumulator` ([lines, 32]: partial maxima, see there).
    `pre_relu`: the statistic of max(x, 0) (folded nn.ReLU)."""
    lib = load()
    x, channel_index, _ = mem_view(x, channel_index)
    outer, C, inner, numel = split3(x.shape, channel_index)
    n = C if channel_index >= 0 else 1
    out = accumulate_into if accumulate_into is not None else torch.empty(n, dtype=torch.float32, device=x.device)
    lines = 1
    if channel_index < 0 and out.dim() == 2:
        assert out.shape[1] == AMAX_LINE_STRIDE and out.is_contiguous() and accumulate_into is not None
        lines = out.shape[0]
    else:
        assert out.numel() == n
    assert out.dtype == torch.float32
    ws, ws_bytes = _reduce_workspace(x, channel_index, outer, C, inner)
    with _timed("absmax", x):
        st = lib.qs_absmax(_ptr(x), _ptr(out), int(channel_index >= 0), outer, C, inner, dt(x),
                           int(accumulate_into is not None), _act(pre_relu), lines, _ptr(ws), ws_bytes, _stream(x))
    _check(st, "qs_absmax")
    return out


def minmax_key_buffers(n: int, device):
    """persistent accumulators of a min/max statistic as order-preserving uint32 keys (held in int32 tensors), neutral:
    min keys 0xffffffff, max keys 0 -- `minmax(accumulate_into=...)` accumulates into them, `lines_update(from_keys=True)`
    consumes and resets them."""
    return (torch.full((n,), -1, dtype=torch.int32, device=device), torch.zeros(n, dtype=torch.int32, device=device))


def minmax(x: torch.Tensor, channel_index: int, accumulate_into=None):
    """(min, max) of x over the tensor / per channel as float32 tensors; with `accumulate_into` (a `minmax_key_buffers`
    pair) ONE launch that leaves keys in those buffers (returned as they are)."""
    lib = load()
    x, channel_index, _ = mem_view(x, channel_index)
    outer, C, inner, numel = split3(x.shape, channel_index)
    n = C if channel_index >= 0 else 1
    if accumulate_into is not None:
        mn, mx = accumulate_into
        assert mn.numel() == n and mx.numel() == n and mn.dtype == torch.int32 and mx.dtype == torch.int32
    else:
        mn = torch.empty(n, dtype=torch.float32, device=x.device)
        mx = torch.empty(n, dtype=torch.float32, device=x.device)
    ws, ws_bytes = _reduce_workspace(x, channel_index, outer, C, inner)
    with _timed("minmax", x):
        st = lib.qs_minmax(_ptr(x), _ptr(mn), _ptr(mx), int(channel_index >= 0), outer, C, inner, dt(x),
                           int(accumulate_into is not None), _ptr(ws), ws_bytes, _stream(x))
    _check(st, "qs_minmax")
    return mn, mx


def scale_update(absmax_t: torch.Tensor, weight: torch.Tensor, t: int, bits: int, t_dev: Optional[torch.Tensor] = None,
                 clear_absmax: bool = False, bump: Optional[torch.Tensor] = None, stat_dtype: torch.dtype = torch.float32,
                 advance_t_dev: bool = False, lines: Optional[int] = None):
    """in place on `weight` (fp32, contiguous); `t_dev`: optional device int64 counter read instead of `t` (and
    incremented by the kernel with `advance_t_dev`);
    `clear_absmax`: zero the statistics buffer after use; `bump`: int32 one-element counter to increment;
    `stat_dtype`: dtype of the tensor the abs-max came from (the reference divides in that dtype);
    `lines`: number of partial accumulators when `absmax_t` is a `tensor_amax_accumulator` (tensor-wise only)."""
    assert weight.dtype == torch.float32 and weight.is_contiguous()
    assert bump is None or bump.dtype == torch.int32
    if lines is None:     # callers that own a `tensor_amax_accumulator` say so; a bare [n] / [1] statistic is one line
        lines = 1
    assert lines == 1 or (weight.numel() == 1 and absmax_t.numel() == lines * AMAX_LINE_STRIDE)
    with _timed("scale_update"):
        st = load().qs_scale_update(_ptr(absmax_t), lines, _ptr(weight), weight.numel(), int(t), _ptr(t_dev), int(advance_t_dev),
                                int(bits), int(clear_absmax), _ptr(bump), _DT.get(stat_dtype, F32), _stream(weight))
    _check(st, "qs_scale_update")


def lines_update(mn: torch.Tensor, mx: torch.Tensor, lines: torch.Tensor, t_after: int,
                 t_dev: Optional[torch.Tensor] = None, advance_t_dev: bool = False, from_keys: bool = False):
    """`from_keys`: mn / mx are the key buffers `minmax(accumulate_into=...)` filled; consumed and reset here."""
    assert lines.dtype == torch.float32 and lines.is_contiguous()
    assert mn.dtype == mx.dtype == (torch.int32 if from_keys else torch.float32)
    with _timed("lines_update"):
        st = load().qs_lines_update(_ptr(mn), _ptr(mx), _ptr(lines), mn.numel(), int(t_after), _ptr(t_dev), int(advance_t_dev),
                                int(from_keys), _stream(lines))
    _check(st, "qs_lines_update")


def decimal_from_scale(scale: torch.Tensor) -> torch.Tensor:
    s = scale.detach().to(torch.float32).contiguous()
    d = torch.empty_like(s)
    with _timed("decimal_from_scale"):
        st = load().qs_decimal_from_scale(_ptr(s), _ptr(d), s.numel(), _stream(s))
    _check(st, "qs_decimal_from_scale")
    return d


AMAX_LINE_STRIDE = 32   # QS_AMAX_LINE_STRIDE: floats per 128-byte line


def amax_accumulator(C: int, device) -> torch.Tensor:
    """zeroed per-channel abs-max accumulator with one 128-byte line per channel ([C, 32]; channel c lives in
    [c, 0]): atomics into one line serialise on MI355X, see qs_mean_dim in the header."""
    return torch.zeros(C, AMAX_LINE_STRIDE, dtype=torch.float32, device=device)


def amax_stride(t: Optional[torch.Tensor]) -> int:
    """element stride between channels of an abs-max tensor: [C] (dense) or [C, stride] (padded)."""
    if t is None or t.dim() < 2:
        return 1
    assert t.is_contiguous()
    return t.shape[1]


def amax_values(t: torch.Tensor) -> torch.Tensor:
    """the [C] view of a dense or padded abs-max tensor."""
    return t[:, 0] if t.dim() == 2 else t.view(-1)


def mean_dim(x: torch.Tensor, pre: int, n: int, post: int, out_dtype: torch.dtype, flags: int = 0,
             l0_flag: Optional[torch.Tensor] = None, absmax_out: Optional[torch.Tensor] = None, chan_div: int = 1,
             C: int = 1, mr_cols: Optional[int] = None) -> torch.Tensor:
    """x: contiguous storage viewed as [pre, n, post]; returns a flat [pre*post] tensor.  `mr_cols`: the cascade prefix of a
    permuted tensor's memory view (qs_mean_dim_split) instead of ATen's rule for a contiguous tensor."""
    x = dense(x)
    out = torch.empty(pre * post, dtype=out_dtype, device=x.device)
    if mr_cols is not None:
        assert absmax_out is None
        with _timed("mean_dim", x, out):
            st = load().qs_mean_dim_split(_ptr(x), _ptr(out), pre, n, post, int(mr_cols), dt(x), _DT[out_dtype], int(flags),
                                          _ptr(l0_flag), _stream(x))
        _check(st, "qs_mean_dim_split")
        return out
    with _timed("mean_dim" + ("+absmax" if absmax_out is not None else ""), x, out):
        st = load().qs_mean_dim(_ptr(x), _ptr(out), pre, n, post, dt(x), _DT[out_dtype], int(flags), _ptr(l0_flag),
                                _ptr(absmax_out), amax_stride(absmax_out), int(chan_div), int(C), _stream(x))
    _check(st, "qs_mean_dim")
    return out


LAST2_MAX_TILE_BYTES = 63 * 1024 - 64     # qs_mean_last2 holds an [H*W + W + 8] float tile in LDS (64 KiB per workgroup)


def mean_last2(x: torch.Tensor, pre: int, H: int, W: int, out_dtype: torch.dtype, amax_part: Optional[torch.Tensor] = None,
               absmax_out: Optional[torch.Tensor] = None, record: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x viewed as [pre, H, W] -> [pre]: mean over H then over W, each rounded like ``Tensor.mean``.  `amax_part`
    ([pre, H, W] from mean_dim_cl) is reduced per slice into `absmax_out` on the way.  `record` (float32 [2 * pre]):
    receives the rank's exchange record (means | `absmax_out`'s values) -- what `stats_pack` would write."""
    x = dense(x)
    out = torch.empty(pre, dtype=out_dtype, device=x.device)
    assert record is None or (record.dtype == torch.float32 and record.numel() == 2 * pre and record.is_contiguous())
    with _timed("mean_last2", x, amax_part):
        st = load().qs_mean_last2(_ptr(x), _ptr(out), pre, H, W, dt(x), _DT[out_dtype], _ptr(amax_part), _ptr(absmax_out),
                                  amax_stride(absmax_out), _ptr(record), _stream(x))
    _check(st, "qs_mean_last2")
    return out


def mean_dim_cl(x_nhwc: torch.Tensor, out_dtype: torch.dtype, flags: int, want_amax: bool,
                l0_flag: Optional[torch.Tensor] = None):
    """first squeeze stage of a channels_last activation given as its contiguous [N, H, W, C] view: mean over N ->
    ([C*H*W] in NCHW order, per-element maxima or None).  Any C; `l0_flag` with MEAN_L0 in `flags`."""
    N, C = x_nhwc.shape[0], x_nhwc.shape[-1]
    hw = x_nhwc.numel() // (N * C)
    out = torch.empty(C * hw, dtype=out_dtype, device=x_nhwc.device)
    part = torch.empty(C * hw, dtype=torch.float32, device=x_nhwc.device) if want_amax else None
    with _timed("mean_dim" + ("+absmax" if want_amax else ""), x_nhwc, out, part):
        st = load().qs_mean_dim_cl(_ptr(x_nhwc), _ptr(out), N, hw, C, dt(x_nhwc), _DT[out_dtype], int(flags), _ptr(l0_flag),
                                   _ptr(part), _stream(x_nhwc))
    _check(st, "qs_mean_dim_cl")
    return out, part


def mean_cl_w(x_nhwc: torch.Tensor, out_dtype: torch.dtype, flags: int, l0_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """the one squeeze stage of a channels_last activation, given as its contiguous [N, H, W, C] view, whose first reduced dim
    is W: mean over W -> [N, C, H] (NCHW order) in ATen's order for that layout (qs_mean_cl_w)"""
    N, H, W, C = x_nhwc.shape
    out = torch.empty(N * C * H, dtype=out_dtype, device=x_nhwc.device)
    with _timed("mean_dim", x_nhwc, out):
        st = load().qs_mean_cl_w(_ptr(x_nhwc), _ptr(out), N, H, W, C, dt(x_nhwc), _DT[out_dtype], int(flags), _ptr(l0_flag),
                                 _stream(x_nhwc))
    _check(st, "qs_mean_cl_w")
    return out


STRIDED_MAX_KEPT = 6     # kStridedMaxKept of qs_reduce.h


def mean_strided(x: torch.Tensor, plan, out_dtype: torch.dtype, flags: int, l0_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """first squeeze stage of a dense tensor in any dim order, read in place: `plan` = (n, stride, kept, order, split_dim, split)
    from util.aten_reduce_plan, kept = [(size, input stride, output stride)] with the lane dim first -> flat contiguous result
    (qs_mean_strided)"""
    n, stride, kept, order, split_dim, split = plan
    numel = 1
    for size, _, _ in kept:
        numel *= size
    out = torch.empty(numel, dtype=out_dtype, device=x.device)
    k = len(kept)
    arr = (ctypes.c_int64 * (3 * max(k, 1)))(*([s for s, _, _ in kept] + [i for _, i, _ in kept] + [o for _, _, o in kept]))
    base = ctypes.addressof(arr)
    with _timed("mean_dim", x, out):
        st = load().qs_mean_strided(_ptr(x), _ptr(out), n, stride, k, base, base + 8 * k, base + 16 * k, order, split_dim, split,
                                    dt(x), _DT[out_dtype], int(flags), _ptr(l0_flag), _stream(x))
    _check(st, "qs_mean_strided")
    return out


def l0_flag(x: torch.Tensor) -> torch.Tensor:
    x = dense(x)
    flag = torch.empty(1, dtype=torch.int32, device=x.device)
    scratch = torch.empty(2, dtype=torch.float32, device=x.device)
    with _timed("l0_flag", x):
        st = load().qs_l0_flag(_ptr(x), x.numel(), dt(x), _ptr(flag), _ptr(scratch), _stream(x))
    _check(st, "qs_l0_flag")
    return flag


def running_mean(state: torch.Tensor, new: torch.Tensor, t: int, t_dev: Optional[torch.Tensor] = None):
    assert state.dtype == torch.float32 and new.numel() == state.numel()
    if state.is_contiguous():
        new = new.contiguous()
    else:
        # a dense state in another memory order (a full-shape magnitude laid out channels_last like its weight): element-wise all
        # the same, once `new` sits in memory the same way
        assert state.dim() == 4 and state.is_contiguous(memory_format=torch.channels_last) and new.shape == state.shape
        new = new.contiguous(memory_format=torch.channels_last)
    with _timed("running_mean"):
        st = load().qs_running_mean(_ptr(state), _ptr(new), dt(new), state.numel(), int(t), _ptr(t_dev), _stream(state))
    _check(st, "qs_running_mean")


# ----------------------------------------------------------------------------------------------
# masks
# ----------------------------------------------------------------------------------------------
def kth_value(imp: torch.Tensor, k: int) -> torch.Tensor:
    lib = load()
    imp = imp.detach().to(torch.float32).contiguous().view(-1)
    n = imp.numel()
    thr = torch.empty(1, dtype=torch.float32, device=imp.device)
    nbytes = lib.qs_workspace_bytes(WS_KTH_VALUE, n)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=imp.device)
    with _timed("kth_value", imp):
        st = lib.qs_kth_value(_ptr(imp), n, int(k), _ptr(thr), _ptr(ws), ws.numel(), _stream(imp))
    _check(st, "qs_kth_value")
    return thr


def mask_ge(imp: torch.Tensor, thr: torch.Tensor, out_mask: torch.Tensor):
    """out_mask (bool, same shape or same numel) <- imp >= thr, in place."""
    imp = imp.detach().to(torch.float32).contiguous()
    assert out_mask.dtype == torch.bool and out_mask.numel() == imp.numel()
    # a mask that is not contiguous (a full-shape 4-d mask after `model.to(memory_format=torch.channels_last)`): the kernel
    # writes the logical order `imp.contiguous()` has, the copy puts it into the mask's own layout
    target = out_mask if out_mask.is_contiguous() else torch.empty(out_mask.shape, dtype=torch.bool, device=out_mask.device)
    with _timed("mask_ge", imp, out_mask):
        st = load().qs_mask_ge(_ptr(imp), _ptr(thr), _ptr(target), imp.numel(), _stream(imp))
    _check(st, "qs_mask_ge")
    if target is not out_mask:
        out_mask.copy_(target)


def mask_apply(x: torch.Tensor, mask: torch.Tensor, pre_relu: bool = False, want_gate: bool = False):
    """x * mask for a bool mask broadcastable to x (same rank, extents 1 or equal); `pre_relu`: max(x, 0) * mask
    (channel-type masks only).  want_gate (with pre_relu): returns (y, ReluGate) -- see quant_fwd."""
    lib = load()
    if mask.dim() != x.dim():
        raise RuntimeError(f"mask rank {mask.dim()} does not match input rank {x.dim()}")
    for sm, sx in zip(mask.shape, x.shape):
        if sm != 1 and sm != sx:
            raise RuntimeError(
                f"The size of tensor a ({sx}) must match the size of tensor b ({sm}) at non-singleton dimension")
    m = mask.detach().to(x.device).contiguous()
    xm, _, like = mem_view(x, 1 if x.dim() in (4, 5) else -1)
    if xm is not like:                    # dense channels_last x, addressed in memory order; the mask dims follow
        m = m.permute((0, 2, 3, 1) if x.dim() == 4 else (0, 2, 3, 4, 1))
    x = xm
    y = torch.empty_like(like)
    gate = None
    if want_gate:
        if not pre_relu:
            raise ValueError("want_gate records the gate of a folded ReLU: pre_relu must be set")
        gate = ReluGate(torch.empty((x.numel() + 7) // 8, dtype=torch.uint8, device=x.device), like, xm is not like)
    if x.numel() == 0:
        return (y, gate) if want_gate else y
    nd = x.dim()
    sizes = (c_int64 * nd)(*x.shape)
    mstr = (c_int64 * nd)(*[0 if m.shape[d] == 1 else m.stride(d) for d in range(nd)])
    with _timed("mask_apply", x, y, gate.bits if gate is not None else None):
        st = lib.qs_mask_apply(_ptr(x), _ptr(m), _ptr(y), nd, sizes, mstr, dt(x), _act(pre_relu), _elide_all(),
                               _ptr(gate.bits) if gate is not None else None, _stream(x))
    _check(st, "qs_mask_apply")
    return (y, gate) if want_gate else y


def pq_select(magnitude: torch.Tensor, stage_mean: Optional[torch.Tensor], update_magnitude: bool, t_mag: int,
              refresh_mask: bool, k: int, mask: torch.Tensor, chan_absmax: Optional[torch.Tensor], update_scale: bool,
              t_q: int, bits: int, scale: Optional[torch.Tensor], bump_a: Optional[torch.Tensor] = None,
              bump_b: Optional[torch.Tensor] = None, bump_c: Optional[torch.Tensor] = None,
              bump_d: Optional[torch.Tensor] = None, t_mag_dev: Optional[torch.Tensor] = None,
              t_q_dev: Optional[torch.Tensor] = None, stat_dtype: torch.dtype = torch.float32,
              gathered: Optional[torch.Tensor] = None, world: int = 1, elide_mask: Optional[torch.Tensor] = None):
    """bump_a / bump_b: int32 one-element counters; bump_c / bump_d: int64 one-element counters; t_*_dev: device
    int64 counters read instead of the by-value t_mag / t_q (each optional).  `gathered`: the all-gathered
    [world, 2C] float32 records of `stats_pack`, combined in rank order by the kernel itself."""
    C = magnitude.numel()
    sdt = dt(stage_mean) if stage_mean is not None else F32
    assert gathered is None or (gathered.dtype == torch.float32 and gathered.numel() == world * 2 * C)
    assert elide_mask is None or (elide_mask.dtype == torch.uint8 and elide_mask.numel() == C and elide_mask.is_contiguous())
    for b32 in (bump_a, bump_b):
        assert b32 is None or b32.dtype == torch.int32
    for b64 in (bump_c, bump_d, t_mag_dev, t_q_dev):
        assert b64 is None or b64.dtype == torch.int64
    with _timed("pq_select"):
        st = load().qs_pq_select(_ptr(magnitude), _ptr(stage_mean), sdt, C, int(update_magnitude), int(t_mag),
                                 int(refresh_mask), int(k), _ptr(mask), _ptr(chan_absmax), amax_stride(chan_absmax),
                                 int(update_scale), int(t_q),
                                 int(bits), _ptr(scale), _ptr(bump_a), _ptr(bump_b), _ptr(bump_c), _ptr(bump_d),
                                 _ptr(t_mag_dev), _ptr(t_q_dev), _DT.get(stat_dtype, F32), _ptr(gathered), int(world),
                                 _ptr(elide_mask), _stream(magnitude))
    _check(st, "qs_pq_select")


def stats_pack(stage: Optional[torch.Tensor], chan_absmax: Optional[torch.Tensor], C: int, device) -> torch.Tensor:
    rec = torch.empty(2 * C, dtype=torch.float32, device=device)
    with _timed("stats_pack"):
        st = load().qs_stats_pack(_ptr(stage), dt(stage) if stage is not None else F32, _ptr(chan_absmax),
                              amax_stride(chan_absmax), C, _ptr(rec), _stream(rec))
    _check(st, "qs_stats_pack")
    return rec


def stats_combine(gathered: torch.Tensor, world: int, C: int, want_stage: bool, absmax_out: Optional[torch.Tensor]):
    stage = torch.empty(C, dtype=torch.float32, device=gathered.device) if want_stage else None
    with _timed("stats_combine"):
        st = load().qs_stats_combine(_ptr(gathered), int(world), C, _ptr(stage), _ptr(absmax_out),
                                 amax_stride(absmax_out), _stream(gathered))
    _check(st, "qs_stats_combine")
    return stage


# ----------------------------------------------------------------------------------------------
# one activation site per call (qs_site_fwd / qs_site_bwd: the launches above in sequence, one FFI transition)
# ----------------------------------------------------------------------------------------------
def logging_events() -> bool:
    """an event log is being recorded: callers keep to the fine-grained entry points so that every launch is bracketed"""
    return _event_log is not None


def quantize_step(x: torch.Tensor, y: torch.Tensor, gate_bits: Optional[torch.Tensor], amax_lines: Optional[torch.Tensor],
                  scale: torch.Tensor, bits: int, t: int, t_dev: Optional[torch.Tensor], n_updates: Optional[torch.Tensor],
                  pre_relu: bool, update, saturate=None, xback: bool = False, image: Optional[torch.Tensor] = None):
    """x: dense (any memory order: the quantizer is tensor-wise), 16-byte aligned; y: same layout; saturate: None or the
    (code_lo, code_hi) pair of the opt-in saturation; update: False / True or one of QSTEP_* (ABSMAX: the abs-max launch alone,
    y may be None; FINISH: running scale from the -- meanwhile all-reduced -- accumulator lines, then quantize)"""
    sat, lo, hi = (0, 0, 0) if saturate is None else (1, int(saturate[0]), int(saturate[1]))
    st = load().qs_quantize_step(x.data_ptr(), None if y is None else y.data_ptr(), None if gate_bits is None else gate_bits.data_ptr(),
                                 None if amax_lines is None else amax_lines.data_ptr(), TENSOR_AMAX_LINES, scale.data_ptr(),
                                 x.numel(), _DT[x.dtype], _DT[(y if y is not None else x).dtype], int(bits), int(t),
                                 None if t_dev is None else t_dev.data_ptr(), None if n_updates is None else n_updates.data_ptr(),
                                 _act(pre_relu), int(update), sat, lo, hi, x.data_ptr() if xback else None,
                                 None if image is None else image.data_ptr(), 0 if image is None else _DT[image.dtype], _stream(x))
    if st:
        _check(st, "qs_quantize_step")


def site_fwd(plan_ref, x: torch.Tensor, y: torch.Tensor, gate_bits: Optional[torch.Tensor], flags: int, t_mag: int, k: int,
             t_q: int, image: Optional[torch.Tensor] = None, gathered: Optional[torch.Tensor] = None, world: int = 1,
             xback: bool = False, decimal: Optional[torch.Tensor] = None):
    """image: optional bf16 / fp16 tensor of y's shape and layout that receives RNE(y) from the same pass (see qs_quant_image_ok);
    gathered (with SITE_STATS_DONE in flags): the all-gathered [world, 2C] records of `site_stats`; xback: the apply kernel also
    writes relu(x) back into x (an owned nn.ReLU(inplace=True), see qs_quant_scaler_fwd xback_out); decimal: float32 [1] of this
    call when the site's quantizer is a DecimalQuantizer (receives the power-of-two step; hand it to `site_bwd`)"""
    st = load().qs_site_fwd(plan_ref, x.data_ptr(), y.data_ptr(), None if gate_bits is None else gate_bits.data_ptr(), flags,
                            t_mag, k, t_q, None if image is None else image.data_ptr(), 0 if image is None else _DT[image.dtype],
                            None if gathered is None else gathered.data_ptr(), world, x.data_ptr() if xback else None,
                            None if decimal is None else decimal.data_ptr(), _stream(x))
    if st:
        _check(st, "qs_site_fwd")


def site_stats(plan_ref, x: torch.Tensor, flags: int, record: torch.Tensor):
    """the statistics launches of a live site step; `record` (float32 [2C]) receives the rank's exchange record"""
    st = load().qs_site_stats(plan_ref, x.data_ptr(), flags, record.data_ptr(), _stream(x))
    if st:
        _check(st, "qs_site_stats")


DACT_GELU = 1             # qs_dact_kind: the caller's activation whose backward a site's backward evaluates (v26)


def site_bwd(plan_ref, g: Optional[torch.Tensor], gate_bits: Optional[torch.Tensor], gx: torch.Tensor, flags: int, lo_mul: float,
             hi_mul: float, g2: Optional[torch.Tensor] = None, decimal: Optional[torch.Tensor] = None,
             g3: Optional[torch.Tensor] = None, gx_image: Optional[torch.Tensor] = None, act_x: Optional[torch.Tensor] = None):
    """g2: a second, 2-byte gradient added to g in float32 (g may then be None), see qs_quant_ste_relu_bwd; g3 / gx_image: the
    riders of the all-float32 backward (qs_site_bwd_v): a third gradient stream of g2's dtype added between g and g2, and a 2-byte
    tensor of gx's shape and layout that receives RNE(gx) from the same pass; act_x (v26): the input of the nn.GELU the caller
    applied in front of the site -- gx comes out as gelu_backward(the site's gradient, act_x); the gate is not read"""
    if g3 is not None or gx_image is not None or act_x is not None:
        a = SiteBwdArgs()
        a.struct_size = ctypes.sizeof(SiteBwdArgs)
        a.flags, a.gdt, a.g2dt = flags, (F32 if g is None else _DT[g.dtype]), (0 if g2 is None else _DT[g2.dtype])
        a.g, a.gate, a.gx = (None if g is None else g.data_ptr()), (None if gate_bits is None else gate_bits.data_ptr()), gx.data_ptr()
        a.lo_mul, a.hi_mul = lo_mul, hi_mul
        a.g2, a.decimal, a.stream = (None if g2 is None else g2.data_ptr()), (None if decimal is None else decimal.data_ptr()), _stream(gx)
        a.g3 = None if g3 is None else g3.data_ptr()
        if gx_image is not None:
            a.gx_image, a.gx_image_dt = gx_image.data_ptr(), _DT[gx_image.dtype]
        if act_x is not None:
            assert act_x.dtype == gx.dtype and act_x.shape == gx.shape and act_x.stride() == gx.stride()
            a.act_x, a.act_x_kind = act_x.data_ptr(), DACT_GELU
        st = load().qs_site_bwd_v(plan_ref, ctypes.byref(a))
        if st:
            _check(st, "qs_site_bwd_v")
        return
    st = load().qs_site_bwd(plan_ref, None if g is None else g.data_ptr(), None if gate_bits is None else gate_bits.data_ptr(),
                            gx.data_ptr(), F32 if g is None else _DT[g.dtype], flags, lo_mul, hi_mul,
                            None if g2 is None else g2.data_ptr(), 0 if g2 is None else _DT[g2.dtype],
                            None if decimal is None else decimal.data_ptr(), _stream(gx))
    if st:
        _check(st, "qs_site_bwd")


def ste_act_bwd(g: Optional[torch.Tensor], act_x: torch.Tensor, step, step_is_decimal: bool, lo_mul: float, hi_mul: float,
                chan_mask: Optional[torch.Tensor] = None, mask_channel_index: int = 1, g2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gelu_backward(RNE(clamp(g [+ float(g2)]) * mask), act_x) from one pass (qs_quant_ste_relu_bwd_v with act_x, ABI v26): the STE
    backward of a site whose input was produced by an nn.GELU the caller applied to `act_x`.  Contiguous operands (the fine-grained
    form of what `site_bwd(act_x=...)` does for every layout of a site plan)."""
    ref = g if g is not None else g2
    assert act_x.is_contiguous() and ref.is_contiguous() and tuple(ref.shape) == tuple(act_x.shape)
    assert (g is None or g.dtype in (torch.float32, act_x.dtype)) and (g2 is None or (g2.is_contiguous() and (g is None or g.dtype == torch.float32)))
    pt, n, host = _f32param(step, ref.device)
    outer, C, inner, numel = split3(ref.shape, mask_channel_index if chan_mask is not None else -1)
    gx = torch.empty_like(act_x)
    if numel == 0:
        return gx
    cm = _chan_mask_bytes(chan_mask, C)
    a = SteReluBwdArgs()
    a.struct_size = ctypes.sizeof(SteReluBwdArgs)
    a.gdt, a.xdt, a.g2dt = (F32 if g is None else dt(g)), dt(act_x), (0 if g2 is None else dt(g2))
    a.g, a.gx, a.step, a.nstep, a.step_host = _ptr(g), _ptr(gx), _ptr(pt), n, host
    a.step_is_decimal, a.lo_mul, a.hi_mul, a.chan_mask = int(bool(step_is_decimal)), float(lo_mul), float(hi_mul), _ptr(cm)
    a.outer, a.C, a.inner, a.g2, a.stream = outer, C, inner, _ptr(g2), _stream(ref)
    a.act_x, a.act_x_kind = _ptr(act_x), DACT_GELU
    with _timed("quant_ste_act_bwd", g, g2, act_x, gx):
        st = load().qs_quant_ste_relu_bwd_v(ctypes.byref(a))
    _check(st, "qs_quant_ste_relu_bwd_v")
    return gx


# ----------------------------------------------------------------------------------------------
# multi-tensor weight path (host arrays of device pointers; see qs_multi_* in the header)
# ----------------------------------------------------------------------------------------------
def ptr_array(tensors):
    """ctypes array of device addresses (None -> NULL)"""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else _ptr(t) for t in tensors])


def i64_array(values):
    return (c_int64 * len(values))(*[int(v) for v in values])


def _device_stream(device):
    """hipStream_t for a multi-tensor launch: same rule as ``_stream`` -- the tensors' device must be the current one"""
    cur = torch.cuda.current_device()
    idx = device.index if device.index is not None else cur
    if idx != cur:
        raise QsparseHipError(f"tensors on cuda:{idx} but the current device is cuda:{cur}: call "
                              "torch.cuda.set_device (one process per GPU) or wrap the call in torch.cuda.device(...)")
    return _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream


def _upload_table(host, n, device) -> torch.Tensor:
    """device copy of a launch table (a few KB).  A blocking copy from pageable memory: it happens once per launch plan (a plan
    is built when a set of layers, their mode or their due operators change), never in a steady-state step.  It cannot be part of
    a hipGraph capture -- a plan that has to be built while the stream is capturing is refused with a clear message instead of
    the capture error of the copy itself (run one eager step in that mode first: graphs.GraphedStep does)."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise QsparseHipError("the weight path has to build a launch table, which cannot happen inside a hipGraph capture: run one "
                              "eager step in this mode (training / evaluation, same set of due operators) before capturing")
    raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8) if n else torch.zeros(0, dtype=torch.uint8)
    return raw.to(device)


class MultiTable:
    """a `qs_multi_row` table: the host copy (a ctypes array the caller fills), the derived launch totals and the device
    copy the kernels read.  Built once per set of layers; `set_train` rewrites the one per-step field and re-uploads (a few KB)
    only when it changed."""

    def __init__(self, rows, device):
        self.n = len(rows)
        self.host = (MultiRow * self.n)(*rows)
        ab, qb, ch, hb = c_int(0), c_int(0), c_int(0), c_int(0)
        _check(load().qs_multi_plan(self.host, self.n, ctypes.byref(ab), ctypes.byref(qb), ctypes.byref(ch), ctypes.byref(hb)),
               "qs_multi_plan")
        self.absmax_blocks, self.quant_blocks, self.channels, self.hist_blocks = ab.value, qb.value, ch.value, hb.value
        self.device = device
        self.dev = None
        self.upload()

    def upload(self):
        self.dev = _upload_table(self.host, self.n, self.device)

    def __deepcopy__(self, memo):
        raise TypeError("a launch table holds raw device pointers: rebuild it, never copy it")


class MultiStage(ctypes.Structure):
    """`qs_multi_stage` of include/qsparse_hip.h, field for field"""
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("pre", c_int64), ("n", c_int64), ("post", c_int64),
                ("take_abs", c_int32), ("layout", c_int32), ("block0", c_int32), ("reserved", c_int32)]


class StageTable:
    """one level of the staged means of a list of tensors (`qs_multi_stage_mean`): host array -> plan -> device copy"""

    def __init__(self, stages, device):
        self.n = len(stages)
        self.host = (MultiStage * self.n)(*stages)
        blocks = c_int(0)
        _check(load().qs_multi_stage_plan(self.host, self.n, ctypes.byref(blocks)), "qs_multi_stage_plan")
        self.blocks, self.device = blocks.value, device
        self.dev = _upload_table(self.host, self.n, device)

    def __deepcopy__(self, memo):
        raise TypeError("a launch table holds raw device pointers: rebuild it, never copy it")


def multi_stage_mean(table: StageTable, nbytes: int = 0):
    with _timed("multi_stage_mean", int(nbytes)):
        st = load().qs_multi_stage_mean(table.dev.data_ptr(), table.n, table.blocks, _device_stream(table.device))
    _check(st, "qs_multi_stage_mean")


def multi_absmax(table: MultiTable, nbytes: int = 0):
    with _timed("multi_absmax", int(nbytes)):
        st = load().qs_multi_absmax(table.dev.data_ptr(), table.n, table.absmax_blocks, _device_stream(table.device))
    _check(st, "qs_multi_absmax")


def multi_scale_update(table: MultiTable):
    with _timed("multi_scale_update"):
        st = load().qs_multi_scale_update(table.dev.data_ptr(), table.n, table.channels, _device_stream(table.device))
    _check(st, "qs_multi_scale_update")


def multi_quant_fwd(table: MultiTable, ybase: torch.Tensor, advance: bool, nbytes: int = 0):
    with _timed("multi_quant_fwd", int(nbytes)):
        st = load().qs_multi_quant_fwd(table.dev.data_ptr(), table.n, table.quant_blocks, ybase.data_ptr(), int(bool(advance)),
                                       _device_stream(table.device))
    _check(st, "qs_multi_quant_fwd")


def i32_array(values):
    return (c_int32 * len(values))(*[int(v) for v in values])


def f32_array(values):
    return (c_float * len(values))(*[float(v) for v in values])


def multi_magnitude(table: MultiTable, nbytes: int = 0):
    """the running magnitudes of the table's pruned weights (rows with `magnitude`), before `multi_quant_fwd` advances their count"""
    with _timed("multi_magnitude", int(nbytes)):
        st = load().qs_multi_magnitude(table.dev.data_ptr(), table.n, table.quant_blocks, _device_stream(table.device))
    _check(st, "qs_multi_magnitude")


def multi_mask_refresh(table: MultiTable, nbytes: int = 0):
    """the mask rebuild of the table's pruned weights that are due (rows with `refresh`): radix select + mask, nine launches"""
    with _timed("multi_mask_refresh", int(nbytes)):
        st = load().qs_multi_mask_refresh(table.dev.data_ptr(), table.n, table.hist_blocks, table.quant_blocks,
                                          _device_stream(table.device))
    _check(st, "qs_multi_mask_refresh")


def multi_ste_bwd(n: int, g_ptrs, gx_ptrs, step_ptrs, numels, lo_muls, hi_muls, decimal: bool, device, nbytes: int = 0,
                  channels=None, inners=None, masks=None, mask_channels=None, mask_inners=None):
    """channels / inners (`i32_array` / `i64_array`, both or neither): per-channel steps -- gradient i is the contiguous
    [*, channels[i], inners[i]] view; None: one step per tensor.  masks (`ptr_array`, entries may be NULL) with mask_channels /
    mask_inners: the pruned weights' masks, gx = clamp(g) * mask"""
    with _timed("multi_ste_bwd", int(nbytes)):
        st = load().qs_multi_ste_bwd(n, g_ptrs, gx_ptrs, step_ptrs, numels, channels, inners, lo_muls, hi_muls, int(bool(decimal)),
                                     masks, mask_channels, mask_inners, _device_stream(device))
    _check(st, "qs_multi_ste_bwd")
