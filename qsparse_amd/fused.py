"""Fused channel-prune -> tensor-wise-quantize pair (kernels K9/K10 of SURVEY.md section 2).

``convert`` builds, for every activation site that is first pruned and then quantized,
``Sequential(Sequential(act, PruneLayer), QuantizeLayer)`` (reference qsparse/convert.py:214-218).  Run
module by module that is, per training step, 2 statistics reads + 2 element-wise passes forward and
2 passes backward.  ``FusedPruneQuantize`` keeps the very same children (and therefore the same
``state_dict`` keys, names and ``str()``) but executes the pair as

    stats   one read of x:  staged mean|x| per channel  +  per-channel max|x|          (qs_mean_dim)
    select  C-sized:        running magnitude, k-th value, mask, max over kept channels,
                            running scale                                               (qs_pq_select)
    apply   one read of x, one write of y:  y = dequant(quant(x * mask[c]))             (qs_quant_*_fwd)
    bwd     one read of g, one write of gx: gx = clamp(g) * mask[c]                      (qs_quant_ste_bwd)

The state machines of both layers (schedules, counters, refresh policy: reference sparse.py:215-273,
99-122 and quantize.py:473-518, 327-349) advance exactly as if the layers had run one after the other;
``tests/test_gpu_parity.py`` checks the trajectories against the unfused path and the oracle.

The same idea one operator at a time: ``FusedActQuantize`` and ``FusedActPrune`` fold a plain ``nn.ReLU`` into a lone
quantize / prune site (``Sequential(act, op)``), see the classes below.
"""
import ctypes
import os

import torch
import torch.nn as nn
from torch.nn.modules import module as _m

from qsparse_amd import _hip
from qsparse_amd.quantize import DecimalQuantizer, QuantizeLayer, ScalerQuantizer, _out_dtype
from qsparse_amd.sparse import MagnitudePruningCallback, PruneLayer
from qsparse_amd.util import _options_epoch, _reduction_plan, _staged_mean_hip, get_option, logging, threshold_rank
from qsparse_amd import distributed as qdist


def _chan_dim(p: PruneLayer, h: torch.Tensor):
    """the channel dim of a site the pair kernels serve -- 1 for (N, C, ...) activations with `dimensions={1}`, the LAST dim of a 3-d
    token-major (B, T, C) activation with `dimensions={2}` -- or None (reference sparse.py:231-239 builds the mask for any set)"""
    if p.dimensions == {1}:
        return 1
    if h.dim() == 3 and p.dimensions == {2}:
        return 2
    return None


def _eligible(p: PruneLayer, q: QuantizeLayer, h: torch.Tensor) -> bool:
    if not h.is_cuda or h.dim() < 2 or h.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return False
    cb, qc = p.callback, q.callback
    if type(cb) is not MagnitudePruningCallback or cb.use_gradient or not cb.running_average or cb.l0:
        return False
    cd = _chan_dim(p, h)
    if cd is None or h.shape[cd] > 65536 or h.shape[cd] < 2:
        return False
    if type(qc) not in (ScalerQuantizer, DecimalQuantizer) or qc.group_num > 0 or q.channelwise != -1:
        return False
    if qc.backward_passthrough:
        return False
    return True


# ----------------------------------------------------------------------------------------------------------------------
# Zero-on-entry accumulators.  The per-channel abs-max rides in the statistics launch as an atomic max into a persistent
# buffer that the select launch re-zeroes; nothing initialises it per step.  If a step dies BETWEEN the two -- a failed
# collective, KeyboardInterrupt, an exception in a hook -- the buffer stays dirty and the next step would silently
# max-accumulate a stale value into the scale.  Every route that opens such a window arms a flag on the owning layer before
# the first launch and clears it after the last; a step that finds the flag set re-zeroes the buffers first.
# ----------------------------------------------------------------------------------------------------------------------
_ARMED = "_qs_accumulators_armed"
# which of the optional routes ran (benchmarks and tests read it; `collections.Counter`, never reset by the library):
#   image / second_image   an autocast consumer took a site's first / second image
#   grad_image             a promoting add's backward used the gradient image the consuming site's backward kernel wrote
#   grad_image_cast        ... or had to cast (another consumer of the sum contributed, or the site took a route without the rider)
#   image_disarmed         a site stopped making images because nobody took the last one
#   image_rearmed          ... and offered one again (after REARM_EVERY steps, or when an option changed)
ROUTES = __import__("collections").Counter()


def _arm_accumulators(q: QuantizeLayer):
    d = q.__dict__
    if d.get(_ARMED):
        for key in ("_chan_absmax", "_chan_absmax_dense"):
            buf = d.get(key)
            if buf is not None:
                buf.zero_()
    d[_ARMED] = True


def _disarm_accumulators(q: QuantizeLayer):
    q.__dict__[_ARMED] = False


def _elision_mask(q, C: int, device) -> torch.Tensor:
    """[C] bytes the select fills for the apply kernel of the same step (fine-grained route; the composite's is in its plan)"""
    buf = q.__dict__.get("_qs_elision")
    if buf is None or buf.numel() != C or buf.device != device:
        buf = torch.empty(C, dtype=torch.uint8, device=device)
        q.__dict__["_qs_elision"] = buf
    return buf


def _absmax_accumulator_dense(q: QuantizeLayer, C: int, device) -> torch.Tensor:
    """persistent dense [C] accumulator for the abs-max passes that run on their own (no statistics stage to ride on):
    zero on entry because the select re-zeroes it, so the reduction needs no initialisation launch."""
    buf = getattr(q, "_chan_absmax_dense", None)
    if buf is None or buf.shape[0] != C or buf.device != device:
        buf = torch.zeros(C, dtype=torch.float32, device=device)
        q._chan_absmax_dense = buf
    return buf


def _absmax_accumulator(q: QuantizeLayer, C: int, device) -> torch.Tensor:
    """persistent per-layer scratch for the per-channel abs-max (not a parameter, not in state_dict)."""
    buf = getattr(q, "_chan_absmax", None)
    if buf is None or buf.shape[0] != C or buf.device != device:
        buf = _hip.amax_accumulator(C, device)   # [C, 32]: one 128-byte line per channel
        q._chan_absmax = buf
    return buf


class _FusedApply(torch.autograd.Function):
    """y = Q(relu?(x) * mask) forward, gx = gate * clamp(g) * mask backward, each one pass over the tensor."""

    @staticmethod
    def forward(ctx, h, mask_c, scale, kind, bits, notch, quant_on, pre_relu=False, saturate=None, elision=None, cd=1):
        # elision: the elision mask the select wrote from THIS input's statistics (see _hip.elide_mode), or None
        # cd: the dim of h the channel mask runs along (`_chan_dim`)
        ctx.kind, ctx.bits, ctx.notch, ctx.quant_on, ctx.x_dtype = kind, bits, notch, quant_on, h.dtype
        ctx.cd = cd
        ctx.pre_relu = pre_relu
        ctx.has_mask = mask_c is not None
        ctx.gate_meta = None
        # a folded ReLU's backward needs nothing of x but the sign test: the forward records it as one bit per element
        # (qs_quant_*_fwd gate_out) and x is not kept -- the backward reads g and the bitmap instead of g and x
        want_gate = bool(pre_relu and quant_on and ctx.needs_input_grad[0] and get_option("relu_gate"))
        empty = h.new_empty(0)
        if not quant_on:
            ctx.save_for_backward(mask_c if mask_c is not None else empty, scale, h if pre_relu else empty)
            return _hip.mask_apply(h, mask_c.view([-1 if i == cd else 1 for i in range(h.dim())]))
        out_dtype = _out_dtype(h)
        # what the backward clamps with is what the reference's Function saved: the scale PARAMETER itself for a
        # ScalerQuantizer (quantize.py:108 -- a later statistics update is seen by an earlier forward's backward), the
        # decimal computed at THIS forward for a DecimalQuantizer (a fresh tensor, quantize.py:312-325 -> :41)
        param = scale if kind == "scaler" else _hip.decimal_from_scale(scale)
        res = _hip.quant_fwd(kind, h, param, -1, torch.float32, chan_mask=mask_c if elision is None else elision,
                             mask_channel_index=cd, out_dtype=out_dtype, elision_mask=elision is not None,
                             pre_relu=pre_relu, want_gate=want_gate, saturate=saturate,
                             xback=_hip.owned_relu_cell() if (want_gate and h.data_ptr() % 16 == 0) else None)
        # the bitmap travels through save_for_backward like any saved activation (released with the graph, visible to
        # saved-tensor hooks); only its description -- shape, dtype, layout -- stays on ctx
        if want_gate:
            gate = res[2]
            ctx.gate_meta = (gate.shape, gate.dtype, gate.channels_last)
            ctx.save_for_backward(mask_c if mask_c is not None else empty, param, gate.bits)
        else:
            ctx.save_for_backward(mask_c if mask_c is not None else empty, param, h if pre_relu else empty)
        return res[0]

    @staticmethod
    def backward(ctx, g):
        mask_c, step, x = ctx.saved_tensors
        mask_c = mask_c if ctx.has_mask else None
        if not ctx.quant_on:
            return (_hip.mask_apply(g, mask_c.view([-1 if i == ctx.cd else 1 for i in range(g.dim())])),) + (None,) * 10
        limit = 2.0 ** (ctx.bits - 1)
        if ctx.pre_relu:
            gate = _hip.ReluGate.from_saved(x, *ctx.gate_meta) if ctx.gate_meta is not None else None
            gx = _hip.ste_relu_bwd(g, None if gate is not None else x, step, ctx.kind == "decimal", -limit + ctx.notch,
                                   limit - 1 + ctx.notch, mask_c, mask_channel_index=ctx.cd, gate=gate, act=ctx.pre_relu)
            return (gx,) + (None,) * 10
        out_dtype = ctx.x_dtype if g.dtype == torch.float32 else g.dtype
        gx = _hip.ste_bwd(g, step, ctx.kind == "decimal", -1, -limit + ctx.notch, limit - 1 + ctx.notch, False,
                          out_dtype, chan_mask=mask_c, mask_channel_index=ctx.cd)
        return (gx,) + (None,) * 10


# ----------------------------------------------------------------------------------------------------------------------
# one FFI call per site and direction (qs_site_fwd / qs_site_bwd, include/qsparse_hip.h): the launches of a live training
# step -- statistics, last two stages, select, apply -- enqueued from ONE ctypes call with a cached plan instead of four
# calls with their argument marshalling (the fine-grained entry points stay the per-call-site binding of INTEGRATION.md)
# ----------------------------------------------------------------------------------------------------------------------
class _SitePlan:
    """`qs_site_plan` of one site and input signature plus what keeps its pointers alive"""
    __slots__ = ("key", "c", "ref", "keep", "out_dtype", "channels_last", "xdt", "image_ok", "image_made", "image_used", "image_fused",
                 "xbuf", "widen_nomask", "decimal", "cd", "interleaved", "idle", "epoch")

    def __init__(self):
        self.key = None
        self.xbuf = None
        self.image_ok, self.image_made, self.image_used, self.image_fused = True, False, False, False
        self.idle, self.epoch = 0, 0

    def __deepcopy__(self, memo):        # a cache of raw pointers never travels: copies and pickles rebuild their own
        return _SitePlan()

    def __reduce__(self):
        return (_SitePlan, ())


def _site_plan(p: PruneLayer, q: QuantizeLayer, h: torch.Tensor, act: int = 1):
    """the cached plan of this site for inputs like `h`, or None when the site is not one the composite call covers
    (4-d NCHW / channels_last, 2-d [N, C] or contiguous 3-d token-major [B, T, C] activation -- mask on the last dim -- with a
    batch of at least two, tensor-wise Scaler / Decimal quantizer, state on h's device)"""
    cb, qc = p.callback, q.callback
    token = h.dim() == 3 and p.dimensions == {2}
    if (h.dim() not in (2, 4) and not token) or h.shape[0] < 2 or type(qc) not in (ScalerQuantizer, DecimalQuantizer) or not hasattr(cb, "magnitude"):
        return None
    flat = h.dim() == 2
    if token:                            # qs_site_plan layout 3: N = B, H = T, W = 1 (two staged means: over B, then over T)
        N, C, H, W = h.shape[0], h.shape[2], h.shape[1], 1
        if H < 2 or not h.is_contiguous():
            return None                  # (T == 1 has one stage less: the fine-grained route)
    else:
        N, C, H, W = (h.shape[0], h.shape[1], 1, 1) if flat else h.shape
    if not flat and not token and (H < 2 or W < 2):
        return None                      # (an extent of 1 is not reduced -- and turns its neighbour into an inner reduction)
    cl = not h.is_contiguous()
    if cl and (flat or not h.is_contiguous(memory_format=torch.channels_last)):
        return None
    if h.data_ptr() % 16 or (not token and (H * W + W) * 4 > _hip.LAST2_MAX_TILE_BYTES):
        return None
    if cl and C % 8:
        return None                      # (any channel count works through the fine-grained path's generic kernel)
    state = (cb.magnitude, p.mask, q.weight, p._n_updates, q._n_updates, cb.t)
    if any((not t.is_cuda) or t.device != h.device for t in state):
        return None
    if p.mask.numel() != C or cb.magnitude.numel() != C or q.weight.numel() != 1:
        return None                      # (a channel count that does not match the mask raises on the fine-grained route)
    graph_safe = bool(get_option("graph_safe"))
    t_q_dev = qc.device_t(h.device) if graph_safe else None
    out_dtype = _out_dtype(h)
    sat = qc.code_range(q.bits)
    key = (N, C, H, W, flat, token, h.dtype, cl, h.device, out_dtype, graph_safe, q.bits, sat, act, type(qc)) + tuple(t.data_ptr() for t in state) + \
        ((t_q_dev.data_ptr(),) if t_q_dev is not None else ())
    plan = q.__dict__.get("_qs_site_plan")
    if plan is not None and plan.key == key:
        return plan
    plan = _SitePlan()
    plan.key, plan.out_dtype, plan.channels_last, plan.xdt = key, out_dtype, cl, h.dtype
    plan.decimal = type(qc) is DecimalQuantizer
    plan.cd = 2 if token else 1          # the dim of h the mask runs along
    plan.interleaved = cl or token       # channels are the fastest-running index in memory: eliding pruned ones saves no traffic
    acc = _absmax_accumulator(q, C, h.device)
    stage = None if flat else torch.empty(C * H * W, dtype=h.dtype, device=h.device)
    # per-element abs-max keys of the first statistics stage: channels_last (reduced per channel by qs_mean_last2) and token-major
    # (qs_token_stats: per column, folded per channel by its third launch)
    part = torch.empty(C * H * W, dtype=torch.float32, device=h.device) if (cl or token) else None
    stage_mean = torch.empty(C, dtype=h.dtype, device=h.device)
    # a frozen-mask step (QS_SITE_SCALE_ONLY): dense [C] abs-max accumulator + the reduction's scratch for this geometry
    dense = _absmax_accumulator_dense(q, C, h.device)
    so, si = (N * H * W, 1) if (cl or flat or token) else (N, H * W)
    ws_bytes = _hip.reduce_workspace_bytes(1, so, C, si)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=h.device) if ws_bytes else None
    elide = torch.empty(C, dtype=torch.uint8, device=h.device)     # the select's elision mask (see _hip.elide_mode)
    plan.keep = (acc, stage, part, stage_mean, t_q_dev, dense, ws, elide) + state
    c = _hip.SitePlanStruct()
    c.N, c.C, c.H, c.W = N, C, H, W
    c.layout, c.xdt, c.ydt, c.bits = (3 if token else 2 if flat else int(cl)), _hip.dt(h), _hip._DT[out_dtype], int(q.bits)
    c.magnitude, c.mask, c.scale = cb.magnitude.data_ptr(), p.mask.data_ptr(), q.weight.data_ptr()
    c.chan_absmax, c.absmax_stride = acc.data_ptr(), _hip.amax_stride(acc)
    c.stage, c.amax_part, c.stage_mean = (None if flat else stage.data_ptr()), (part.data_ptr() if part is not None else None), stage_mean.data_ptr()
    c.prune_n_updates, c.quant_n_updates, c.callback_t = p._n_updates.data_ptr(), q._n_updates.data_ptr(), cb.t.data_ptr()
    c.quantizer_t_dev = t_q_dev.data_ptr() if t_q_dev is not None else None
    c.callback_t_from_device = int(graph_safe)
    c.saturate, c.code_lo, c.code_hi = (0, 0, 0) if sat is None else (1, sat[0], sat[1])
    c.act = int(act) or 1                # the activation QS_SITE_PRE_RELU folds at this site (1: nn.ReLU)
    c.elide_mask = elide.data_ptr()
    c.absmax_dense, c.reduce_ws, c.reduce_ws_bytes = dense.data_ptr(), (ws.data_ptr() if ws is not None else None), ws_bytes
    # whether the forward kernel of this geometry can write the autocast image itself (else it is a cast of y)
    outer, inner = (N * H * W, 1) if (cl or token) else (N, H * W)
    plan.image_fused = bool(_hip.load().qs_quant_image_ok(outer, C, inner, 0, 1, int(p.mask.data_ptr() % 8 == 0), _hip.dt(h)))
    # (the same kernels write an owned in-place ReLU's result back into x; without the mask the launch is tensor-wise)
    plan.widen_nomask = bool(_hip.load().qs_quant_image_ok(1, 1, outer * C * inner, 0, 0, 1, _hip.dt(h)))
    plan.c, plan.ref = c, ctypes.byref(c)
    q.__dict__["_qs_site_plan"] = plan
    return plan


# ----------------------------------------------------------------------------------------------------------------------
# Autocast image.  The reference's quantizers return float32 for a bf16 input (quantize.py:109-117), so under autocast every
# convolution that consumes a site's output casts it back: an `fp32 -> bf16` pass in the forward and its `bf16 -> fp32` mirror
# image in the backward, 6.1 ms of a 56.5 ms ResNet-50 step.  With the option `autocast_image` the site
# hands its FIRST autocast consumer the bf16 image itself -- the very values ATen's cast would produce -- and receives that
# consumer's bf16 gradient directly: its backward kernel adds it, in float32, to whatever float32 gradient the output's
# other consumers delivered (qs_quant_ste_relu_bwd g2).  Every value is the reference's: the image is RNE(y), the gradient
# sum is autograd's own accumulation (the first consumer's backward is the last to arrive, so even the grouping of more
# than two terms is kept).  What changes is the TYPE of the site's output: a `torch.Tensor` subclass that carries the image
# until a convolution / linear / matmul under autocast takes it; every other operation sees a plain float32 tensor.
# The DEFAULT under autocast (option `autocast_image`, on since round 5).  The image consumer's gradient reaches the site's backward
# directly, NOT through the output tensor; what can observe that is handled: `register_hook` / `retain_grad` on the output before
# the consumer ran cancel the image (the whole gradient then flows through the tensor as usual); registered AFTER the consumer took
# it they run as pre-hooks of the site's backward node, which holds every stream (`_late_hook`); `torch.autograd.grad(loss, y)`
# differentiates with respect to the images as well and adds the shares (`_grad_through_duals`).
#
# Second image (round 6).  A block output that feeds the next block's first convolution AND its down-sampling convolution has two
# autocast consumers; the second used to cast for itself (6 B/elem forward, 6 B/elem for the cast's backward).  The site's forward
# returns the image TWICE -- two tensors on one storage, two gradient slots of the backward node -- and the second consumer gets
# the second one, provided NOTHING touched the output between the two (`_qs_image_b` is armed by the first take and dropped by
# any other consumer).  Autograd delivers the consumers' shares in reverse order of the consumers' creation, so the reference's
# float32 accumulation at the output is ((float32 consumers, created last) + f32(second)) + f32(first): exactly what the backward
# kernel evaluates from the three streams (qs_site_bwd_v g / g3 / g2) for a float32 site; elsewhere the second share is added the
# way autograd would (a cast and an add), before the kernel.
#
# Promoting add (round 6).  `bn(conv(h)) + identity` under autocast is bf16 + float32 -> float32 (the identity is a site's
# output); ATen's AddBackward0 hands the bf16 operand its gradient as a CAST of the sum's gradient, a 6 B/elem pass per residual
# block.  The sum's only consumer is the next site, whose backward kernel produces that gradient: when the add involves a site's
# output the library runs the same ATen add inside a node of its own (`_PromotingAdd`), tags the sum, and the consuming site's
# backward writes RNE(gx) next to gx in the same pass (qs_site_bwd_v gx_image, +2 B/elem) -- which the add's backward hands to the
# bf16 operand when the gradient it receives IS that gx (no other consumer of the sum contributed), and casts itself otherwise.
# ----------------------------------------------------------------------------------------------------------------------
def _image_consumers():
    import torch.nn.functional as F
    fns = [torch.conv1d, torch.conv2d, torch.conv3d, torch.conv_transpose1d, torch.conv_transpose2d, torch.conv_transpose3d,
           F.linear, torch.matmul, torch.mm, torch.bmm, F.conv1d, F.conv2d, F.conv3d]
    return frozenset(fns)


_IMAGE_CONSUMERS = _image_consumers()
_GRADIENT_OBSERVERS = frozenset([torch.Tensor.register_hook, torch.Tensor.retain_grad])
# calls that read a tensor's metadata (or alias it without a gradient path): they do not make the tensor's gradient a sum of one
# more term.  EVERYTHING else that touches a site's output before its image consumer does cancels the image -- the image consumer is
# then always the FIRST consumer, whose backward autograd runs last: its share is the last term of the float32 accumulation in both
# routes, so sums of three and more gradient streams keep the reference's grouping (found by tests/fuzz/fuzz_image.py: a residual
# read BEFORE the convolution gave (g_cat + g_res) + g_conv against autograd's (g_cat + g_conv) + g_res, one float32 ulp apart)
_NON_CONSUMING = frozenset(getattr(torch.Tensor, n) for n in (
    "size", "dim", "ndimension", "stride", "numel", "nelement", "is_contiguous", "data_ptr", "storage_offset", "element_size",
    "is_floating_point", "is_complex", "get_device", "dim_order", "detach", "__len__", "__repr__", "__str__", "__format__",
    "is_shared", "is_pinned", "has_names", "untyped_storage") if hasattr(torch.Tensor, n))


# ... and the attribute reads that are metadata (`y.T`, `y.mT`, `y.real` are views with a gradient path: consumers)
_METADATA_ATTRS = frozenset(("shape", "dtype", "device", "requires_grad", "is_cuda", "is_cpu", "grad", "grad_fn", "is_leaf", "ndim", "layout",
                             "names", "_version", "output_nr", "is_quantized", "is_sparse", "is_sparse_csr", "is_meta", "is_mkldnn",
                             "is_xpu", "is_mps", "is_nested", "itemsize", "nbytes", "retains_grad", "_base", "_grad", "_backward_hooks",
                             "is_ipu", "is_xla", "is_mtia", "is_maia", "is_vulkan", "is_ort", "volatile", "name"))


def _non_consuming(func) -> bool:
    if func in _NON_CONSUMING:
        return True
    return getattr(func, "__name__", "") == "__get__" and getattr(getattr(func, "__self__", None), "__name__", None) in _METADATA_ATTRS


def _drop_images(d):
    d.pop("_qs_image", None)
    d.pop("_qs_image_b", None)
    d.pop("_qs_image_b_pending", None)


def _cancel_images(args):
    """drop the images (not yet taken) of every site output among `args` (one level of lists / tuples deep: torch.cat([y, z]))"""
    for a in args:
        if type(a) is AutocastImageTensor:
            _drop_images(a.__dict__)
        elif isinstance(a, (list, tuple)):
            for b in a:
                if type(b) is AutocastImageTensor:
                    _drop_images(b.__dict__)


def _whole(g, g16, g16b=None):
    """autograd's own accumulation of the gradient streams of a site's output: float32 consumers, then the second image's
    consumer, then the first image's (the order in which their backward nodes run: reverse order of creation)"""
    if g16b is not None:
        g = g16b.float() if g is None else g + g16b.float()
    if g16 is None:
        return g
    return g16.float() if g is None else g + g16.float()


def _slots(dual):
    """the backward node's gradient slots of (output, image, second image | None)"""
    sl = dual.__dict__.get("_qs_slots", (0, 1, 2))
    return sl if len(sl) == 3 else (sl[0], sl[1], None)


def _late_hook(dual, fn):
    """`register_hook` on a site's output AFTER its image was taken: the hook must see the WHOLE gradient of the output, of which
    the image consumer's share never passes through the tensor -- so it runs as a pre-hook of the site's backward node, which
    holds both streams.  A hook that only looks (returns None) leaves the fast route untouched; one that returns a
    replacement turns the step into the plain one: (replacement, no second stream)."""
    node = dual.__dict__["_qs_node"]
    i0, i1, i2 = _slots(dual)       # the node's gradient slots of the output, of its image and of the second image

    def pre(grads):
        g, g16 = grads[i0], grads[i1]
        g16b = grads[i2] if (i2 is not None and i2 < len(grads)) else None
        earlier = node.__dict__.get("_qs_override")
        replaced = earlier is not None and i0 in earlier        # an earlier late hook already replaced this gradient: chain on it
        full = earlier[i0] if replaced else _whole(g, g16, g16b)
        if full is None:
            return None
        r = fn(full)
        if r is None:
            return None
        if g is None or replaced:
            # (a pre-hook cannot put a gradient where autograd has none: the replacement rides on the node itself -- for a Python
            #  Function the node IS the ctx -- and the node's backward takes it in place of both streams)
            node.__dict__.setdefault("_qs_override", {})[i0] = r
            return None
        out = list(grads)
        out[i0], out[i1] = r, None
        if g16b is not None:
            out[i2] = None
        return tuple(out)

    return node.register_prehook(pre)


def _late_retain_grad(dual):
    import weakref
    torch.Tensor.retain_grad(dual)
    ref, node = weakref.ref(dual), dual.__dict__["_qs_node"]
    i0, i1, i2 = _slots(dual)

    def pre(grads):
        d = ref()
        g16b = grads[i2] if (i2 is not None and i2 < len(grads)) else None
        if d is not None and (grads[i1] is not None or g16b is not None):
            with torch.no_grad():
                d.grad = _whole(grads[i0], grads[i1], g16b)    # (the tensor's own retain-grad hook has stored the float32 share by now)
        return None

    node.register_prehook(pre)


def _grad_through_duals(args, kwargs):
    """`torch.autograd.grad(outputs, inputs, ...)` with a site's output among `inputs` after its image was taken: differentiate
    with respect to the image as well and add the two shares (float32 accumulation, as autograd itself would)"""
    kwargs = dict(kwargs or {})
    args = list(args)
    inputs = kwargs.pop("inputs") if "inputs" in kwargs else args.pop(1)
    outputs = kwargs.pop("outputs") if "outputs" in kwargs else args.pop(0)
    single = isinstance(inputs, torch.Tensor)
    ins = [inputs] if single else list(inputs)
    # (second image first: its consumer's share is added before the first consumer's, as autograd's own accumulation would)
    extra = [(i, t.__dict__[key]) for i, t in enumerate(ins) if type(t) is AutocastImageTensor
             for key in ("_qs_image_taken_b", "_qs_image_taken") if t.__dict__.get(key) is not None]
    if not extra:
        return torch.autograd.grad(outputs, inputs, *args, **kwargs)
    allow_unused = kwargs.pop("allow_unused", None)
    materialize = kwargs.pop("materialize_grads", False)
    res = list(torch.autograd.grad(outputs, ins + [img for _, img in extra], *args, allow_unused=True, **kwargs))
    shares = res[len(ins):]
    res = res[:len(ins)]
    for (i, _), g16 in zip(extra, shares):
        res[i] = _whole(res[i], g16)
    for i, r in enumerate(res):
        if r is None:
            if materialize:
                res[i] = torch.zeros_like(ins[i].as_subclass(torch.Tensor))
            elif not allow_unused:
                raise RuntimeError("One of the differentiated Tensors appears to not have been used in the graph. Set allow_unused=True "
                                   "if this is the desired behavior.")
    return tuple(res)


class AutocastImageTensor(torch.Tensor):
    """float32 output of a quantize site that also carries its low-precision image for ONE autocast consumer.  Everything that
    can OBSERVE the output's gradient sees the whole gradient: `register_hook` / `retain_grad` before the image is taken cancel
    it; after it was taken they, and `torch.autograd.grad(..., inputs=[output])`, combine the two streams (`_late_hook`)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            if func in _GRADIENT_OBSERVERS and args and type(args[0]) is cls:
                # someone wants to SEE this tensor's gradient: it must be the whole one
                d = args[0].__dict__
                d.pop("_qs_image_b_pending", None)
                if d.pop("_qs_image", None) is None and d.get("_qs_image_taken") is not None and d.get("_qs_node") is not None:
                    # ... and a consumer has already bypassed the tensor: observe at the site's backward node instead
                    if func is torch.Tensor.retain_grad:
                        return _late_retain_grad(args[0])
                    return _late_hook(args[0], *args[1:], **(kwargs or {}))
            elif func in _IMAGE_CONSUMERS and args and (type(args[0]) is cls or (len(args) > 1 and type(args[1]) is cls)):
                # the input (an activation site's output) and / or the weight (the weight path's quantized tensor, batch.py)
                for pos in (0, 1):
                    a = args[pos] if pos < len(args) else None
                    if type(a) is not cls:
                        continue
                    d = a.__dict__
                    held, second = d.pop("_qs_image", None), False
                    if held is None:
                        # the second image: armed when the first consumer took the first one, still there if nothing has touched
                        # the output since (a third autocast consumer casts for itself)
                        held, second = d.pop("_qs_image_b", None), True
                        if held is None:
                            _drop_images(d)
                            continue
                    img, version, plan = held
                    pending = d.pop("_qs_image_b_pending", None)
                    if (a._version == version and torch.is_autocast_enabled("cuda")
                            and torch.get_autocast_dtype("cuda") == img.dtype):
                        if img.requires_grad:
                            d["_qs_image_taken_b" if second else "_qs_image_taken"] = img
                        plan.image_used = True
                        ROUTES["second_image" if second else "image"] += 1
                        args = tuple(args[:pos]) + (img,) + tuple(args[pos + 1:])
                        if pending is not None and not second:
                            d["_qs_image_b"] = (pending, version, plan)
                _cancel_images(args[2:])
            elif func is torch.autograd.grad:
                return _grad_through_duals(args, kwargs)
            elif func in _PROMOTING_ADDS and len(args) == 2 and not kwargs:
                _cancel_images(args)          # (a float32 consumer like any other)
                s = _promoting_add(args[0], args[1])
                if s is not None:
                    return s
            elif not _non_consuming(func):
                _cancel_images(args)          # another consumer comes first: the image's would not be the last term of the sum
            return func(*args, **(kwargs or {}))

    def __repr__(self):
        with torch._C.DisableTorchFunctionSubclass():
            return torch.Tensor.__repr__(self.as_subclass(torch.Tensor))

    # copies and pickles are plain tensors: the image (and the site plan behind it) belongs to THIS forward pass
    def __deepcopy__(self, memo):
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__deepcopy__(memo)

    def __reduce_ex__(self, proto):
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__reduce_ex__(proto)


# ---- the promoting add in front of a site (see "Promoting add" above) ----------------------------------------------------------
class _PromotingAdd(torch.autograd.Function):
    """`a + b` of a 2-byte tensor and a float32 site output: ATen's own add forward; backward hands the 2-byte operand the image
    of the gradient the consuming site's backward kernel wrote (cell["g16"]) when the gradient arriving here IS that site's"""

    @staticmethod
    def forward(ctx, a, b, cell):
        ctx.cell, ctx.dtypes = cell, (a.dtype, b.dtype)
        return torch.add(a, b)

    @staticmethod
    def backward(ctx, g):
        cell = ctx.cell
        gx, g16 = cell.pop("gx", None), cell.pop("g16", None)
        out = []
        for i, dt in enumerate(ctx.dtypes):
            if not ctx.needs_input_grad[i]:
                out.append(None)
            elif dt == g.dtype:
                out.append(g)
            elif g16 is not None and gx is not None and g is gx and g16.dtype == dt:
                out.append(g16)
                ROUTES["grad_image"] += 1
            else:
                out.append(g.to(dt))
                ROUTES["grad_image_cast"] += 1
        return out[0], out[1], None


_PROMOTING_ADDS = frozenset((torch.add, torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__))
_GRAD_IMAGE_CELL = "_qs_grad_image_cell"


def _promoting_add(a, b):
    """the sum through `_PromotingAdd`, tagged for the site that consumes it -- or None when this is not the case the node serves:
    a float32 site output plus a bf16 / fp16 CUDA tensor that requires grad, same shape (no broadcast to undo in the backward)"""
    if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)) or not torch.is_grad_enabled() or not get_option("autocast_image"):
        return None
    lo, hi = (a, b) if a.dtype != torch.float32 else (b, a)
    if (hi.dtype != torch.float32 or lo.dtype not in (torch.bfloat16, torch.float16) or not lo.requires_grad or not lo.is_cuda
            or lo.shape != hi.shape or lo.device != hi.device or type(hi) is not AutocastImageTensor):
        return None
    cell = {"dtype": lo.dtype}
    s = _PromotingAdd.apply(a, b, cell)
    s.__dict__[_GRAD_IMAGE_CELL] = cell
    return s


def grad_image_cell(h):
    """the cell of the promoting add that produced `h` (the consuming site's backward fills it), or None"""
    return h.__dict__.get(_GRAD_IMAGE_CELL) if h.dtype == torch.float32 else None


def _as_dual(y, img, plan, slots=None, img_b=None):
    """the site's float32 output as the subclass that carries its image to the first autocast consumer.  `slots`: the gradient
    slots of (output, image) at the backward node when they are not (0, 1) -- the weight path's hand-out groups.  `img_b`: the
    second image (the same storage, the node's third gradient slot) for a second autocast consumer"""
    plan.image_made = True
    dual = y.as_subclass(AutocastImageTensor)
    dual.__dict__["_qs_image"] = (img, y._version, plan)
    if img_b is not None:
        dual.__dict__["_qs_image_b_pending"] = img_b
    dual.__dict__["_qs_node"] = y.grad_fn        # the backward node: both gradient streams arrive there (None: no grad)
    if slots is not None:
        dual.__dict__["_qs_slots"] = slots
    return dual


class ImageStat:
    """bookkeeping of a maker of images that is not a `_SitePlan` (a lone quantizer behind its activation): whether the last image
    was taken, whether to keep making them"""
    __slots__ = ("image_ok", "image_made", "image_used", "idle", "epoch")

    def __init__(self):
        self.image_ok, self.image_made, self.image_used = True, False, False
        self.idle, self.epoch = 0, 0

    def __deepcopy__(self, memo):
        return ImageStat()

    def __reduce__(self):
        return (ImageStat, ())


REARM_EVERY = 1024      # steps after which a maker whose image nobody took offers one again


def image_bookkeeping(stat):
    """once per step of an image maker (a `_SitePlan` / `ImageStat`), before it decides whether to make one: an image nobody took
    (the consumer is not an autocast matmul / convolution -- or a hook touched the output first, or an evaluation pass fed another
    consumer) stops the making; it is offered again after REARM_EVERY steps or as soon as an option changes (the options epoch), so
    ONE odd forward cannot bring the 2 x 6 B/elem cast passes back for the rest of the process.  `ROUTES` counts both."""
    if stat.image_made and not stat.image_used:
        stat.image_ok, stat.idle, stat.epoch = False, 0, _options_epoch[0]
        ROUTES["image_disarmed"] += 1
    elif not stat.image_ok:
        stat.idle += 1
        if stat.idle >= REARM_EVERY or stat.epoch != _options_epoch[0]:
            stat.image_ok = True
            ROUTES["image_rearmed"] += 1
    stat.image_made = stat.image_used = False


def autocast_image_dtype():
    """dtype of the images this forward should make, or None: the option is on, autocast to bf16 / fp16 is active, and tensors
    have version counters (not inference mode)"""
    if not get_option("autocast_image") or not torch.is_autocast_enabled("cuda") or torch.is_inference_mode_enabled():
        return None
    dt = torch.get_autocast_dtype("cuda")
    return dt if dt in (torch.bfloat16, torch.float16) else None


_IDENTITY = [None]
_IDENTITY_FOLD = os.environ.get("QS_NO_IDENTITY_FOLD", "0") != "1"      # (development switch for A/B measurements)


def identity_fold_handle(h) -> int:
    """The autocast image is written by the kernels that record a folded activation's gate (and its gradient comes back through the
    gated backward), so a site WITHOUT a foldable activation in front -- nn.GELU, nn.SiLU, nn.Identity, a lone QuantizeLayer on an
    activation -- had no image: its float32 output was cast by its consumer, 12 B/elem of cast passes per step (a ViT MLP encoder
    converted with nn.GELU sites: 1.23 x plain, with nn.ReLU sites 1.085).  Such a site folds the IDENTITY instead:
    nn.LeakyReLU(negative_slope=1.0), `x > 0 ? x : x * 1.0`, is x bit for bit in every dtype, its backward `g * 1.0` is g -- the
    kernels run as for a folded nn.LeakyReLU, record a gate nobody needs (1/8 B/elem each way) and write the image.  Returns that
    activation's handle when an image is wanted for `h` right now (autocast on, option on, float32 promotion in place), else 0."""
    if autocast_image_dtype() is None or not get_option("relu_gate") or not isinstance(h, torch.Tensor) or not h.is_cuda or not _IDENTITY_FOLD:
        return 0
    if h.dtype not in (torch.float32, torch.bfloat16, torch.float16) or _out_dtype(h) != torch.float32 or h.data_ptr() % 16:
        return 0
    if _IDENTITY[0] is None:
        _IDENTITY[0] = _hip.try_activation(_hip.ACT_LEAKY, 1.0)
    return _IDENTITY[0] or 0


def _image_dtype(plan, training_needs_gate: bool):
    """dtype of the image this step should produce, or None"""
    if not get_option("autocast_image") or not plan.image_ok or plan.out_dtype != torch.float32:
        return None
    if not torch.is_autocast_enabled("cuda") or torch.is_inference_mode_enabled():
        return None      # (an inference tensor has no version counter: a write to the site's output could not be seen)
    dt = torch.get_autocast_dtype("cuda")
    if dt not in (torch.bfloat16, torch.float16) or (training_needs_gate and not get_option("relu_gate")):
        return None
    return dt


def _exchange_buffers(plan: _SitePlan, C: int, world: int, device):
    """(this rank's record [2C], the gathered records [world * 2C]) of a data-parallel site step: persistent -- the collective
    is ordered on the stream like the kernels around it, so one pair per site serves every step"""
    buf = plan.xbuf
    if buf is None or buf[1].numel() != world * 2 * C:
        buf = plan.xbuf = (torch.empty(2 * C, dtype=torch.float32, device=device),
                           torch.empty(world * 2 * C, dtype=torch.float32, device=device))
    return buf


class _SiteStep(torch.autograd.Function):
    """the whole site through qs_site_fwd / qs_site_bwd; same results as the statistics + select + `_FusedApply` route"""

    @staticmethod
    def forward(ctx, h, plan, flags, t_mag, k, t_q, bits, notch, mask_c, scale, image_dtype=None, gathered=None, world=1, add_cell=None,
                act_out=None):
        # add_cell: of the promoting add that produced h (`grad_image_cell`), or None
        # act_out: the caller applied an nn.GELU under no_grad -- act_out = gelu(h) is what the site reads, h (the first, differentiable
        # argument) is the GELU's INPUT, kept for the backward, whose kernel multiplies by the GELU's derivative (`act_backward`)
        act_in = None
        if act_out is not None:
            act_in, h = h, act_out
        want_gate = bool((flags & _hip.SITE_PRE_RELU) and ctx.needs_input_grad[0] and get_option("relu_gate"))
        y = torch.empty_like(h, dtype=plan.out_dtype)
        make_image = image_dtype is not None and (want_gate or not ctx.needs_input_grad[0])
        # the image comes out of the forward kernel itself when the gate-recording kernels serve this geometry with the mask
        # (forward-only calls -- evaluation, serving -- let them record a bitmap nobody reads: 1/8 B/elem for a 6 B/elem pass)
        fused_image = bool(make_image and (flags & _hip.SITE_PRE_RELU) and plan.image_fused and not (flags & _hip.SITE_NO_MASK))
        # an owned nn.ReLU(inplace=True) in front of this site has not touched x yet: the apply kernel writes relu(x) back into
        # x's own storage when it is one of the gate-recording widening kernels (forward-only calls record a bitmap nobody reads)
        cell = _hip.owned_relu_cell() if (flags & _hip.SITE_PRE_RELU) else None
        xback = bool(cell is not None and plan.out_dtype == torch.float32
                     and (plan.widen_nomask if (flags & _hip.SITE_NO_MASK) else plan.image_fused))
        bits_t = torch.empty((h.numel() + 7) // 8, dtype=torch.uint8, device=h.device) if (want_gate or fused_image or xback) else None
        if want_gate:
            _hip.note_gate(bits_t)
        img = torch.empty_like(h, dtype=image_dtype) if fused_image else None
        if (flags & _hip.SITE_ELIDE) and not _hip._elide_fwd(plan.interleaved, bits_t is not None, bool(flags & _hip.SITE_LIVE)):
            flags &= ~_hip.SITE_ELIDE          # elision only where it saves traffic and is exact (see _hip.elide_mode)
        # a DecimalQuantizer's power-of-two step of THIS call (the backward clamps with it; two forwards may precede a backward)
        dec = torch.empty(1, dtype=torch.float32, device=h.device) if plan.decimal else None
        _hip.site_fwd(plan.ref, h, y, bits_t, flags, t_mag, k, t_q, image=img, gathered=gathered, world=world, xback=xback,
                      decimal=dec)
        if xback:
            cell["done"] = True
        if _hip.image_byte_delta is not None and img is not None:
            _hip.image_byte_delta["apply_fwd"] += img.numel() * img.element_size()
        ctx.plan, ctx.flags, ctx.bits, ctx.notch, ctx.has_gate = plan, flags, bits, notch, want_gate
        ctx.act, ctx.dec, ctx.add_cell = plan.c.act, dec, add_cell
        ctx.x_shape, ctx.x_dtype = h.shape, h.dtype
        keep_x = bool(flags & _hip.SITE_PRE_RELU) and not want_gate
        ctx.has_act_x = act_in is not None
        ctx.save_for_backward(mask_c if mask_c is not None else h.new_empty(0), scale,
                              bits_t if want_gate else (h if keep_x else h.new_empty(0)), act_in if act_in is not None else h.new_empty(0))
        ctx.set_materialize_grads(False)
        if make_image:
            im = img if fused_image else y.to(image_dtype)      # RNE(y): the cast autocast would apply in front of a convolution
            return y, im, im.detach()     # ... twice: the same storage, a gradient slot of its own for a second autocast consumer
        return y

    @staticmethod
    def backward(ctx, g, g16=None, g16b=None):
        plan, flags = ctx.plan, ctx.flags
        n_in = 15
        override = ctx.__dict__.pop("_qs_override", None)
        if override is not None:         # a late hook on the output replaced its whole gradient (fused._late_hook)
            g, g16, g16b = override[0], None, None
        if g16 is None and g16b is not None:
            g16, g16b = g16b, None       # (only the second consumer's share exists: it is the one 2-byte stream)
        if g is None and g16 is None:
            return (None,) * n_in
        mask_c, scale, third, act_x = ctx.saved_tensors
        if not ctx.has_act_x:
            act_x = None
        limit = 2.0 ** (ctx.bits - 1)
        lo_mul, hi_mul = -limit + ctx.notch, limit - 1 + ctx.notch
        pre_relu = bool(flags & _hip.SITE_PRE_RELU)
        fmt = torch.channels_last if plan.channels_last else torch.contiguous_format

        def dense(t):
            return t.is_contiguous(memory_format=fmt) and t.data_ptr() % 16 == 0 and tuple(t.shape) == tuple(ctx.x_shape)

        # the riders of the all-float32 backward kernel (qs_site_bwd_v): the second image consumer's share as a third stream, and
        # the image of gx for the promoting add that produced this site's input
        f32_gated = ctx.has_gate and ctx.x_dtype == torch.float32
        cell = ctx.add_cell if f32_gated else None

        def grad_image(gx):
            if cell is None:
                return None
            gimg = torch.empty(ctx.x_shape, dtype=cell["dtype"], device=gx.device, memory_format=fmt)
            # (the cell holds gx until the add's backward ran: that also keeps the engine from accumulating INTO it)
            cell["gx"], cell["g16"] = gx, gimg
            if _hip.image_byte_delta is not None:
                _hip.image_byte_delta["apply_bwd"] += gimg.numel() * gimg.element_size()
            return gimg

        if g16 is not None:
            # the image's consumer delivered its low-precision gradient: g + float(g16) inside the kernel (qs_site_bwd g2)
            dual_ok = ((ctx.has_gate or act_x is not None) and dense(g16) and (g is None or (g.dtype == torch.float32 and dense(g)))
                       and _hip.elide_mode != "all" and not _hip.logging_events())
            if g16b is not None and not (dual_ok and f32_gated and g16b.dtype == g16.dtype and dense(g16b)):
                g = g16b.float() if g is None else g + g16b.float()    # the second consumer's share as autograd would add it
                g16b = None
                dual_ok = dual_ok and g.dtype == torch.float32 and dense(g)
            if dual_ok:
                gx = torch.empty(ctx.x_shape, dtype=ctx.x_dtype, device=g16.device, memory_format=fmt)
                if _hip.image_byte_delta is not None:
                    _hip.image_byte_delta["apply_bwd"] += g16.numel() * g16.element_size() - (g16.numel() * 4 if g is None else 0)
                    if g16b is not None:
                        _hip.image_byte_delta["apply_bwd"] += g16b.numel() * g16b.element_size()
                if act_x is not None:
                    ROUTES["act_backward"] += 1
                _hip.site_bwd(plan.ref, g, third if ctx.has_gate else None, gx, flags & _hip.SITE_NO_MASK, lo_mul, hi_mul, g2=g16,
                              decimal=ctx.dec, g3=g16b, gx_image=grad_image(gx), act_x=act_x)
                return (gx,) + (None,) * (n_in - 1)
            g = _whole(g, g16, g16b)          # autograd's own accumulation, then the usual routes
        fast = (dense(g) and (ctx.has_gate or not pre_relu) and g.dtype in (torch.float32, ctx.x_dtype)
                and not _hip.logging_events())
        if fast:
            gx = torch.empty(ctx.x_shape, dtype=ctx.x_dtype, device=g.device, memory_format=fmt)
            bflags = (flags & _hip.SITE_NO_MASK) | (_hip.SITE_ELIDE if _hip.elide_mode == "all" else 0)
            gimg = grad_image(gx) if (g.dtype == torch.float32 and not (bflags & _hip.SITE_ELIDE)) else None
            if act_x is not None:
                ROUTES["act_backward"] += 1
            _hip.site_bwd(plan.ref, g, third if ctx.has_gate else None, gx, bflags, lo_mul, hi_mul, decimal=ctx.dec, gx_image=gimg,
                          act_x=act_x)
            return (gx,) + (None,) * (n_in - 1)
        mask = mask_c.detach().view(-1) if mask_c.numel() else None      # (the layers' own parameters were saved, not aliases)
        is_dec = ctx.dec is not None
        scale = ctx.dec if is_dec else scale.detach()
        if pre_relu:
            gate = _hip.ReluGate.from_saved(third, ctx.x_shape, ctx.x_dtype, plan.channels_last) if ctx.has_gate else None
            gx = _hip.ste_relu_bwd(g, None if gate is not None else third, scale, is_dec, lo_mul, hi_mul, mask, mask_channel_index=plan.cd,
                                   gate=gate, act=ctx.act)
        else:
            out_dtype = ctx.x_dtype if g.dtype == torch.float32 else g.dtype
            gx = _hip.ste_bwd(g, scale, is_dec, -1, lo_mul, hi_mul, False, out_dtype, chan_mask=mask, mask_channel_index=plan.cd)
        if act_x is not None:            # the caller's nn.GELU, as autograd's GeluBackward0 would evaluate it on this gradient
            gx = torch.ops.aten.gelu_backward(gx if gx.dtype == act_x.dtype else gx.to(act_x.dtype), act_x)
        return (gx,) + (None,) * (n_in - 1)


class _ActGrad(torch.autograd.Function):
    """h = gelu(x) was computed under no_grad (`act_rider`): this node puts it back into the graph -- its backward is autograd's own
    GeluBackward0, `gelu_backward(g, x)` -- for the routes whose backward kernel does not evaluate the GELU itself"""

    @staticmethod
    def forward(ctx, x, h):
        ctx.save_for_backward(x)
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return torch.ops.aten.gelu_backward(g if g.dtype == x.dtype else g.to(x.dtype), x), None


def act_rider(act, x):
    """`act_backward`: the nn.GELU (erf form) `act` in front of a site keeps ATen's forward, evaluated here under no_grad, and hands its
    backward to the site (`_SiteStep(act_out=)`, qs_site_bwd_args::act_x; `_ActGrad` elsewhere).  Returns h = act(x) detached from the
    graph, or None when the site should run `act` as an ordinary module (not a GELU, option off, nothing to differentiate, float16 --
    see `_FastPair.try_run` --, a layout the kernels do not address in place)."""
    if not (type(act) is nn.GELU and getattr(act, "approximate", "none") == "none" and isinstance(x, torch.Tensor) and x.is_cuda
            and x.requires_grad and torch.is_grad_enabled() and get_option("act_backward") and x.data_ptr() % 16 == 0
            and x.dtype in (torch.float32, torch.bfloat16) and _hip.dense_any_order(x)):
        return None
    with torch.no_grad():
        h = act(x)           # ATen's own forward; the graph sees one node, x -> y, whose backward knows the GELU
    if h.stride() == x.stride() and h.dtype == x.dtype and h.data_ptr() % 16 == 0:
        return h
    return None


def fused_prune_quantize(p: PruneLayer, q: QuantizeLayer, h: torch.Tensor, pre_relu=False, act_in=None) -> torch.Tensor:
    """one training / evaluation step of ``q(p(h))`` on a GPU tensor -- or of ``q(p(act(h)))`` with ``pre_relu``: True for a
    folded nn.ReLU, or the handle of another folded activation (``_fold_handle``; the caller guarantees that the quantizer is
    active this step, so the activation is applied inside the kernels).  ``act_in``: h is `act_rider(gelu, act_in)` -- detached;
    the result's gradient flows to act_in through the GELU's backward."""
    cb, qc = p.callback, q.callback
    cd = _chan_dim(p, h) or 1            # the dim the channel mask runs along (the caller checked `_eligible`)
    C = h.shape[cd]

    # ---- PruneLayer.forward bookkeeping (reference sparse.py:228-273) ----
    if not p.initted:
        p._lazy_init(h)
    n = p.advance_schedule()
    prune_on = True
    update_mag = refresh = False
    t_mag = 0
    k = 0
    if p.training and p.mask.numel() != 1:
        if n >= p.start:
            if n == p.start and get_option("log_during_train"):
                logging.warning(f"Start pruning at {p.name} @ {n}")
            sparsity = p.current_sparsity()
            t_mag = cb.begin_step(p.mask)
            update_mag = t_mag < cb.stop_mask_refresh
            refresh = cb.refresh_due(t_mag, sparsity)
            if refresh:
                k = threshold_rank(sparsity, C)
                if k >= C:
                    raise IndexError(f"index {k} is out of bounds for dimension 0 with size {C}")
        else:
            prune_on = False

    # ---- QuantizeLayer.forward bookkeeping (reference quantize.py:482-517) ----
    if not q.initted:
        q._lazy_init(h)
    quant_on = update_scale = False
    t_q = 0
    if q.timeout > 0:
        tq = q._steps.read(q._n_updates)
        if tq >= q.timeout:
            if q.training:
                if tq == q.timeout and get_option("log_during_train"):
                    logging.warn(f"quantizing {q.name} with {q.bits} bits")
                update_scale = True
                t_q = qc.t
                q._quantized = True
            quant_on = q._quantized

    # ---- statistics: one read of h ----
    p_counts = p.training and p.mask.numel() != 1
    q_counts = q.timeout > 0 and q.training
    bump_p = bump_q = bump_t = None
    select_bumped_tq = False
    elision = None          # the select's elision mask of THIS step, when it wrote one (fine-grained route)
    # one FFI call for the whole site (qs_site_fwd) when this is a live steady-state step -- magnitude, mask policy and
    # scale all updated from this input -- or a pure apply step (evaluation, frozen statistics) of a plain 4-d site
    site = None
    live = update_mag and update_scale and prune_on and p_counts and q_counts and n >= p.start
    idle = not (update_mag or refresh or update_scale) and quant_on
    world = qdist.stats_world_size()
    exchange = qdist.exchange_active(world)      # (only a live step has statistics to exchange)
    # the mask is frozen (the callback's t passed stop_mask_refresh: the steady state of the reference's layerwise recipe,
    # sparse.py:343-359) and the scale still follows the data: abs-max per channel, the select's scale half, apply
    frozen = (update_scale and not (update_mag or refresh) and prune_on and p_counts and q_counts and n >= p.start
              and not exchange)
    if (live or idle or frozen) and not _hip.logging_events():
        site = _site_plan(p, q, h, int(pre_relu) or 1)
    if site is not None and (live or frozen):   # the counters ride in the select launch, as on the fine-grained route (the
        bump_p = bump_q = bump_t = True          # flags below only ask WHETHER they did)
        select_bumped_tq = bool(get_option("graph_safe"))
    site_gathered = None
    with torch.no_grad():
        hd = h.detach() if site is None else None
        stage = chan_absmax = record = None
        if site is not None:
            if live or frozen:
                _arm_accumulators(q)
                if exchange and live:
                    # data-parallel step, two calls around ONE collective: the statistics launches (the last of them writes
                    # this rank's record), the all-gather, and -- in `_SiteStep` below -- select + apply on the gathered records
                    rec, site_gathered = _exchange_buffers(site, C, world, h.device)
                    _hip.site_stats(site.ref, h, _hip.SITE_PRE_RELU if pre_relu else 0, rec)
                    if qdist.mailbox_enabled():      # (prototype: peers' records arrive in a mapped mailbox, no host collective)
                        site_gathered = qdist.mailbox_exchange(q, rec)
                    else:
                        qdist.all_gather_records(site_gathered, rec)
        elif update_scale and not prune_on:
            # pruning not started yet: the scale follows max|h| of the whole tensor (quantize.py:329-348)
            am = qdist.allreduce_max_(_hip.absmax(hd, -1, pre_relu=pre_relu), world)
            _hip.scale_update(am, q.weight.data.view(-1), t_q, q.bits,
                              t_dev=qc.device_t(h.device) if get_option("graph_safe") else None, stat_dtype=h.dtype)
        else:
            # step counters that live on this GPU ride along in the select launch (callback.t stays on the CPU
            # when the module was never moved with .to(device): that one is then bumped on the host)
            def on_dev(t):
                return t.data if (t.is_cuda and t.device == h.device) else None

            if update_mag or refresh or update_scale:
                bump_p = on_dev(p._n_updates) if p_counts else None
                bump_q = on_dev(q._n_updates) if q_counts else None
                bump_t = on_dev(cb.t) if (p_counts and n >= p.start) else None
            if update_mag or update_scale:
                _arm_accumulators(q)
            if update_mag:
                dims = _reduction_plan(hd.shape, p.mask.shape)
                # the per-channel abs-max rides along in the first statistics stage when that stage reduces a dim
                # in front of the channel dim (the batch); with a batch of one there is no such stage and the
                # abs-max is a pass of its own
                rides = update_scale and bool(dims) and dims[0] < cd
                if rides and not hd.is_contiguous() and not (hd.dim() == 4 and hd.is_contiguous(memory_format=torch.channels_last)):
                    # channels_last_3d, and every other dense layout (a transposed / permuted activation): the in-place first
                    # stage -- ATen's order for that layout, qs_mean_dim_cl / qs_mean_strided -- carries no abs-max
                    rides = False
                if rides:
                    chan_absmax = _absmax_accumulator(q, C, h.device)   # zero on entry, re-zeroed by the select
                elif update_scale:
                    chan_absmax = _hip.absmax(hd, cd, pre_relu=pre_relu, accumulate_into=_absmax_accumulator_dense(q, C, h.device))
                if qdist.exchange_active(world) and h.is_cuda:   # the last statistics launch writes the exchange record
                    record = {"buf": torch.empty(2 * C, dtype=torch.float32, device=h.device), "filled": False}
                stage = _staged_mean_hip(hd, dims, take_abs=True, absmax_out=chan_absmax if rides else None,
                                         absmax_channel_dim=cd, pre_relu=pre_relu, record=record,
                                         record_absmax=chan_absmax).contiguous().view(-1)
            elif update_scale:
                chan_absmax = _hip.absmax(hd, cd, pre_relu=pre_relu, accumulate_into=_absmax_accumulator_dense(q, C, h.device))
            gathered = None
            if qdist.exchange_active(world) and (stage is not None or chan_absmax is not None):
                if h.is_cuda:     # one collective; the select kernel combines the ranks' records in rank order
                    gathered = qdist.gather_pair_statistics(stage, chan_absmax, world, record)
                else:
                    stage, chan_absmax = qdist.sync_pair_statistics(stage, chan_absmax, world)
            if update_mag or refresh or update_scale:
                mag = cb.magnitude.data.view(-1) if hasattr(cb, "magnitude") else torch.zeros(C, device=h.device)
                t_mag_dev = t_q_dev = None
                if get_option("graph_safe"):   # running-mean counters come from device memory (hipGraph replay)
                    t_mag_dev = on_dev(cb.t) if update_mag else None
                    t_q_dev = qc.device_t(h.device) if update_scale else None
                if update_scale and h.dim() == 4 and h.is_contiguous() and _hip.elide_mode != "off":
                    elision = _elision_mask(q, C, h.device)      # (an NCHW forward may skip rows: see _hip.elide_mode)
                _hip.pq_select(mag, stage, update_mag, t_mag, refresh, k, p.mask.data.view(-1), chan_absmax,
                               update_scale, t_q, q.bits, q.weight.data, bump_a=bump_p, bump_b=bump_q, bump_c=bump_t,
                               bump_d=t_q_dev, t_mag_dev=t_mag_dev, t_q_dev=t_q_dev, stat_dtype=h.dtype,
                               gathered=gathered, world=world if gathered is not None else 1, elide_mask=elision)
                _disarm_accumulators(q)
                select_bumped_tq = t_q_dev is not None
        if update_scale:
            if select_bumped_tq:
                qc._advance_t(qc.__dict__["_t_dev"], bumped_by_kernel=True)
            else:
                qc._advance_t(qc.device_t(h.device) if (get_option("graph_safe") and not prune_on) else None)

    # ---- counters (same order as the unfused layers) ----
    if p_counts:
        if n >= p.start:
            if bump_t is not None:
                cb._t_host.note_device_add(cb.t, 1)
                if cb.forward_hook is not None:
                    cb.forward_hook(p.mask, p.name)
            else:
                cb.end_step(p.mask, p.name)
        if bump_p is not None:
            p._steps.note_device_add(p._n_updates, 1)
        else:
            p._steps.add(p._n_updates, 1)
    if q_counts:
        if bump_q is not None:
            q._steps.note_device_add(q._n_updates, 1)
        else:
            q._steps.add(q._n_updates, 1)

    # ---- apply: one read of h, one write ----
    if act_in is not None and site is None:
        h = _ActGrad.apply(act_in, h)          # (only the composite's backward kernel evaluates the GELU itself)
    if not prune_on and not quant_on:
        return _hip.act_torch(pre_relu, h) if pre_relu else h
    if pre_relu and not quant_on:      # cannot happen when the caller checked q.is_active(); stay correct anyway
        h, pre_relu = _hip.act_torch(pre_relu, h), False
    kind = "scaler" if isinstance(qc, ScalerQuantizer) else "decimal"
    if site is not None:
        flags = ((_hip.SITE_LIVE if (live or frozen) else 0) | (_hip.SITE_SCALE_ONLY if frozen else 0)
                 | (_hip.SITE_REFRESH if refresh else 0) | (_hip.SITE_PRE_RELU if pre_relu else 0)
                 | (_hip.SITE_ELIDE if _hip.elide_mode != "off" else 0) | (0 if prune_on else _hip.SITE_NO_MASK))
        image_bookkeeping(site)
        image_dtype = _image_dtype(site, training_needs_gate=torch.is_grad_enabled() and h.requires_grad) if pre_relu else None
        if site_gathered is not None:
            flags |= _hip.SITE_STATS_DONE
        if act_in is None:
            out = _SiteStep.apply(h, site, flags, t_mag, k, t_q, q.bits, 1 if qc.flip_axis else 0,
                                  p.mask if prune_on else None, q.weight, image_dtype, site_gathered, world, grad_image_cell(h))
        else:
            out = _SiteStep.apply(act_in, site, flags, t_mag, k, t_q, q.bits, 1 if qc.flip_axis else 0,
                                  p.mask if prune_on else None, q.weight, image_dtype, site_gathered, world, None, h)
        if live or frozen:
            _disarm_accumulators(q)
        # (what `_FastPair.arm` looks at: a steady-state step of the composite route, no exchange)
        q.__dict__["_qs_last_route"] = ("live" if live else "frozen") if ((live or frozen) and site_gathered is None) else None
        if type(out) is tuple:
            return _as_dual(out[0], out[1], site, img_b=out[2])
        return out
    return _FusedApply.apply(h, p.mask.data.view(-1) if prune_on else None, q.weight.data, kind, q.bits,
                             1 if qc.flip_axis else 0, quant_on, pre_relu, qc.code_range(q.bits),
                             elision if (prune_on and quant_on) else None, cd)


# ----------------------------------------------------------------------------------------------------------------------
# nn.ReLU(inplace=True) in front of a site (what torchvision-style networks carry): the ReLU still has to land in x's own
# storage -- other holders of x see relu(x), as with the plain module -- but its BACKWARD need not be a pass of its own: the
# fused site that consumes the result gates with the bits it recorded (h > 0 <=> x > 0).  Two nodes without kernels of
# their own do the bookkeeping: `_OwnedRelu` (x <- relu(x), marked dirty) and `_Tap` (an alias of the result for the site;
# its backward remembers the gradient the site returned).  When the site is the only consumer of the modified tensor -- the
# rule -- the gradient arriving at `_OwnedRelu` IS that tensor, already gated: it passes through.  A further consumer of the
# modified tensor makes autograd hand over a sum instead; that one is gated here (gating twice changes nothing).
# ----------------------------------------------------------------------------------------------------------------------
class _OwnedRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cell):
        if not cell.get("defer"):        # deferred: the site's apply kernel (or `_with_owned_relu` after it) rectifies x
            _hip.act_torch_(cell["act"], x)
        ctx.mark_dirty(x)
        ctx.cell = cell
        ctx.layout = (x.shape, x.stride())   # (x itself is NOT kept for the backward, unlike ATen's in-place ReLU)
        return x

    @staticmethod
    def backward(ctx, g):
        cell = ctx.cell
        gated = cell.pop("g", None)
        if g is None or g is gated:
            return g, None
        # the rare route: the gate as the site recorded it (or, where the site recorded none, from the tensor itself)
        open_ = (_hip.unpack_gate(cell["bits"], *ctx.layout) if "bits" in cell else _hip.act_gate_of(cell["act"], cell["h"]))
        kind, slope, _ = _hip.act_spec(cell["act"])
        closed = g * slope if kind == _hip.ACT_LEAKY else torch.zeros((), dtype=g.dtype, device=g.device)
        return torch.where(open_, g, closed), None      # (threshold_backward: NaN passes)


class _Tap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, cell):
        ctx.cell = cell
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        ctx.cell["g"] = g                    # (holding it also keeps the engine from accumulating INTO this tensor)
        return g, None


_FOLDABLE = (nn.ReLU, nn.ReLU6, nn.Hardtanh, nn.LeakyReLU)


def _fold_handle(act) -> int:
    """what the kernels' `pre_relu` arguments take for this activation module: 1 for nn.ReLU, a `_hip.activation` handle for
    nn.ReLU6 / nn.Hardtanh (clamp to [a, b]; backward 0 outside (a, b)) and nn.LeakyReLU (slope > 0); 0: not foldable"""
    t = type(act)
    if t is nn.ReLU:
        return 1
    if t in (nn.ReLU6, nn.Hardtanh):       # (nn.ReLU6 is a Hardtanh with min_val = 0, max_val = 6)
        return _hip.try_activation(_hip.ACT_HARDTANH, float(act.min_val), float(act.max_val))
    if t is nn.LeakyReLU and act.negative_slope > 0:
        return _hip.try_activation(_hip.ACT_LEAKY, float(act.negative_slope))
    return 0


def _foldable_relu(act, x):
    """(fold, handle).  fold 0: not an activation the kernels can absorb; 1: out of place; 2: ``inplace=True`` on a tensor the
    library may own (a leaf that requires grad raises in the plain module -- let it; in-place on views stays with ATen's view
    bookkeeping).  handle: see `_fold_handle`."""
    if type(act) not in _FOLDABLE or not get_option("fold_relu") or not isinstance(x, torch.Tensor) or not x.is_cuda:
        return 0, 0
    handle = _fold_handle(act)
    if not handle:
        return 0, 0
    if not act.inplace:
        return 1, handle
    if x._is_view() or (x.is_leaf and x.requires_grad) or not get_option("relu_gate"):
        return 0, 0
    if handle != 1 and not _hip.bounds_representable(handle, x.dtype):
        return 0, 0          # (an owned in-place clamp may have to gate from the RECTIFIED tensor, which such bounds make ambiguous)
    if handle != 1 and type(act) is nn.LeakyReLU and x.data_ptr() % 16:
        return 0, 0          # (not idempotent: it can only be folded when the kernel -- not ATen beforehand -- applies it)
    return 2, handle


def _with_owned_relu(x: torch.Tensor, site, act: int = 1):
    """x <- relu(x) in place, then `site(h)` on the alias the fused site reads (with `pre_relu`: max(h, 0) == h, its gate bits
    are h > 0); the bitmap the site records is also what `_OwnedRelu`'s rare route gates with"""
    # The ReLU is DEFERRED: x keeps its raw values while the site runs -- every one of its kernels takes `pre_relu` and reads
    # max(x, 0) -- and the site's apply kernel, which loads every element of x anyway, stores relu(x) back into x's storage
    # (xback_out of qs_quant_scaler_fwd: +2 / +4 B/elem instead of ATen's 4 / 8 B/elem read + write pass).  A site whose route
    # has no such kernel leaves x raw; the ATen pass below then runs after it -- same tensor contents either way, and in
    # stream order before anything else can read x.
    cell = {"defer": x.data_ptr() % 16 == 0, "act": act, "x": x}
    h = _Tap.apply(_OwnedRelu.apply(x, cell), cell)
    outer = getattr(_hip._gate_sink, "cell", None)
    _hip._gate_sink.cell = cell
    try:
        y = site(h)
    finally:
        _hip._gate_sink.cell = outer
        cell.pop("x", None)                  # (only `_hip.act_torch`, inside the site, looks at it: no cycle through x's own graph)
        if cell.get("defer") and not cell.get("done"):
            with torch.no_grad():
                _hip.act_torch_(act, x)
            cell["done"] = True
    if "bits" not in cell and torch.is_grad_enabled() and h.requires_grad:
        cell["h"] = h.detach()               # a site that recorded no gate this step (quantizer idle ...) kept x itself anyway
    return y


def _hooked(*modules) -> bool:
    """hooks registered on a child (forward, forward-pre, backward, backward-pre) or installed globally for all modules:
    the fused forward never calls the children, so their hooks would not fire -- such a site runs module by module,
    exactly as the plain ``Sequential`` it replaces"""
    if (_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks
            or _m._global_backward_pre_hooks):
        return True
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in modules)


# ----------------------------------------------------------------------------------------------------------------------
# Steady-state fast path of the pair.  `FusedPruneQuantize.forward` spends ~50 of its ~80 us of host time re-deriving decisions
# that do not change from step to step once the schedules have finished: hooks, fold mode, eligibility, the layers' state
# machines, the site plan's key.  After a step that went through the composite call in steady state the site remembers a
# SIGNATURE of everything those decisions depend on -- the modules and their configuration attributes, the options epoch, the
# input's shape / dtype / layout / autograd status, the identity and storage of every state tensor -- and the following steps
# only compare that signature, advance the counters exactly as the full path does and issue the same call.  Anything that does
# not match (a hook registered, an attribute changed, another input, a process group, logging, a counter written from outside:
# the host mirrors notice) takes the full path, which re-arms when it finds the steady state again.
# ----------------------------------------------------------------------------------------------------------------------
_MISS = object()
_FAST_PATH = os.environ.get("QS_NO_FAST_PATH", "0") != "1"      # (development switch for A/B measurements)


def _no_fast():
    return None


def _hook_dicts(modules):
    return tuple(d for m in modules for d in (m._forward_hooks, m._forward_pre_hooks, m._backward_hooks, m._backward_pre_hooks))


def _pair_config(act, p, q, cb, qc):
    """every configuration attribute the pair's decisions read (plain Python attributes: cheap to re-read)"""
    return (type(act), getattr(act, "inplace", None), getattr(act, "min_val", None), getattr(act, "max_val", None),
            getattr(act, "negative_slope", None),
            p.start, p.interval, p.repetition, p.sparsity, p.rampup_interval, tuple(p.schedules), tuple(p.dimensions), p.training,
            type(cb), cb.mask_refresh_interval, cb.stop_mask_refresh, cb.use_gradient, cb.running_average, cb.l0, cb.forward_hook,
            cb.training, q.timeout, q.bits, q.channelwise, q.batch_dimension, q.training, q._quantized,
            type(qc), qc.group_num, qc.backward_passthrough, qc.flip_axis, qc.use_uint, qc.__dict__.get("saturate"), qc.training)


class _FastPair:
    __slots__ = ("epoch", "mods", "hooks", "config", "xsig", "state", "ptrs", "plan", "pre_relu", "fold", "max_schedule", "C",
                 "graph_safe", "notch", "k_of", "autocast", "dact")

    def __deepcopy__(self, memo):        # raw pointers and object identities: a copied network arms its own
        return None

    def __reduce__(self):
        return (_no_fast, ())

    @staticmethod
    def _xsig(x):
        return (x.shape, x.dtype, x.device, x.stride(), x.data_ptr() % 16, x.requires_grad, x.is_leaf, x._is_view())

    @classmethod
    def arm(cls, seq, x, fold, handle):
        """called by the full path after a step; returns a fast path for the following steps, or None"""
        inner, q = seq[0], seq[1]
        act, p = inner[0], inner[1]
        cb, qc = p.callback, q.callback
        if not _FAST_PATH:
            return None
        # fold: 1 / 2 a foldable activation (out of place / owned in place), 0 nn.Identity, 3 any other activation module (applied by
        # ATen in front of the site); for 0 and 3 `handle` is 0 or the identity fold (`identity_fold_handle`)
        if type(act) is nn.Identity:
            fold = 0
        elif not fold:
            return None
        if get_option("log_during_train") or not (seq.training and inner.training):
            return None
        if q.__dict__.get("_qs_last_route") not in ("live", "frozen") or not p.initted or not q.initted or not cb.initted:
            return None
        plan = q.__dict__.get("_qs_site_plan")
        if plan is None or plan.key is None or p.mask.numel() == 1 or not p.schedules or cb.forward_hook is not None:
            return None
        if p._steps.read(p._n_updates) <= max(p.schedules) or q._steps.read(q._n_updates) <= q.timeout:
            return None                  # (the schedules are still running: those steps take decisions of their own)
        hooks = _hook_dicts((inner, act, p, q, cb, qc))
        if any(hooks):
            return None
        f = cls()
        f.epoch = _options_epoch[0]
        f.mods = (inner, act, p, q, cb, qc)
        f.hooks = hooks
        f.config = _pair_config(act, p, q, cb, qc)
        f.xsig = cls._xsig(x)
        f.state = (cb.magnitude, p.mask, q.weight, p._n_updates, q._n_updates, cb.t)
        f.ptrs = tuple(t.data_ptr() for t in f.state)
        f.plan, f.pre_relu, f.fold = plan, handle, fold
        f.max_schedule, f.C = max(p.schedules), plan.c.C
        f.graph_safe, f.notch = bool(get_option("graph_safe")), (1 if qc.flip_axis else 0)
        f.k_of = {}
        f.autocast = cls._autocast()         # (the identity fold is chosen per autocast state)
        # `act_backward`: an nn.GELU (erf form) in front of the site -- its forward stays ATen's pass, its backward rides in the site's
        # backward kernel (qs_site_bwd_args::act_x)
        f.dact = fold == 3 and type(act) is nn.GELU and getattr(act, "approximate", "none") == "none"
        return f

    @staticmethod
    def _autocast():
        on = torch.is_autocast_enabled("cuda")
        return (on, torch.get_autocast_dtype("cuda") if on else None)

    def try_run(self, seq, x):
        """the step, or _MISS (nothing has been touched then)"""
        mods = seq._modules
        inner, q = mods.get("0"), mods.get("1")
        if inner is not self.mods[0] or q is not self.mods[3] or self.epoch != _options_epoch[0]:
            return _MISS
        im = inner._modules
        act, p = im.get("0"), im.get("1")
        if act is not self.mods[1] or p is not self.mods[2]:
            return _MISS
        pm, qm = p._modules, q._modules
        cb, qc = pm.get("callback"), qm.get("callback")
        if cb is not self.mods[4] or qc is not self.mods[5] or not (seq.training and inner.training):
            return _MISS
        if (_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks or _m._global_backward_pre_hooks
                or _hip.logging_events()):
            return _MISS
        for d in self.hooks:
            if d:
                return _MISS
        if not isinstance(x, torch.Tensor) or self._xsig(x) != self.xsig or _pair_config(act, p, q, cb, qc) != self.config:
            return _MISS
        if self.fold in (0, 3) and self._autocast() != self.autocast:
            return _MISS
        pp, qp, cp = p._parameters, q._parameters, cb._parameters
        state = (cp.get("magnitude"), pp.get("mask"), qp.get("weight"), pp.get("_n_updates"), qp.get("_n_updates"), cp.get("t"))
        for a, b, ptr in zip(state, self.state, self.ptrs):
            if a is not b or a.data_ptr() != ptr:
                return _MISS
        if qdist.exchange_active(qdist.stats_world_size()):
            return _MISS
        # ---- the state machines, read-only (the host mirrors re-read a counter somebody else wrote) ----
        n = p._steps.read(state[3])
        tq = q._steps.read(state[4])
        if n <= self.max_schedule or tq <= q.timeout:
            return _MISS
        t_mag = cb._t_host.read(state[5])
        sparsity = p.current_sparsity()
        update_mag = t_mag < cb.stop_mask_refresh
        refresh = cb.refresh_due(t_mag, sparsity)
        live = update_mag
        if not live and refresh:
            return _MISS                 # (the rebuild at t == stop_mask_refresh: the fine-grained route)
        k = 0
        if refresh:
            k = self.k_of.get(sparsity)
            if k is None:
                k = self.k_of[sparsity] = threshold_rank(sparsity, self.C)
            if k >= self.C:
                return _MISS             # (the full path raises the reference's IndexError)
        plan = self.plan
        if q.__dict__.get("_qs_site_plan") is not plan:
            return _MISS
        t_q = qc.t
        if self.graph_safe and qc.__dict__.get("_t_dev_value") != t_q:
            return _MISS                 # (the device copy of the count is stale: `device_t` rebuilds it on the full path)
        # ---- from here on as `fused_prune_quantize` does for a steady-state step of the composite route ----
        pre_relu = self.pre_relu

        def site(h, act_in=None):
            _arm_accumulators(q)
            if self.graph_safe:
                qc._advance_t(qc.__dict__["_t_dev"], bumped_by_kernel=True)
            else:
                qc._advance_t(None)
            cb._t_host.note_device_add(state[5], 1)
            p._steps.note_device_add(state[3], 1)
            q._steps.note_device_add(state[4], 1)
            flags = (_hip.SITE_LIVE | (0 if live else _hip.SITE_SCALE_ONLY) | (_hip.SITE_REFRESH if refresh else 0)
                     | (_hip.SITE_PRE_RELU if pre_relu else 0) | (_hip.SITE_ELIDE if _hip.elide_mode != "off" else 0))
            image_bookkeeping(plan)
            image_dtype = _image_dtype(plan, training_needs_gate=torch.is_grad_enabled() and h.requires_grad) if pre_relu else None
            if act_in is None:
                out = _SiteStep.apply(h, plan, flags, t_mag, k, t_q, q.bits, self.notch, state[1], state[2], image_dtype, None, 1,
                                      grad_image_cell(h))
            else:
                out = _SiteStep.apply(act_in, plan, flags, t_mag, k, t_q, q.bits, self.notch, state[1], state[2], image_dtype, None, 1,
                                      None, h)
            _disarm_accumulators(q)
            if type(out) is tuple:
                return _as_dual(out[0], out[1], plan, img_b=out[2])
            return out

        if self.fold == 2:
            return _with_owned_relu(x, site, pre_relu)
        if self.fold == 3:
            # (not float16: ATen's fp16 gelu / gelu_backward kernels give different bits in a full block of their vectorised kernel
            #  and in its tail block -- 2 of 20,000 (dy, x) pairs, tools/probes/probe_gelu_tail.py; the kernel here equals the full-block
            #  result everywhere, so a tensor whose size is not a multiple of ATen's block would differ from the module-by-module route.
            #  bf16 and float32 are one function of (dy, x) in ATen and here.)
            h = act_rider(act, x) if self.dact else None
            if h is not None:
                return site(h, x)
            return site(act(x))          # an activation the kernels do not fold: ATen applies it, the site follows
        return site(x)


class FusedPruneQuantize(nn.Sequential):
    """``Sequential(Sequential(act, PruneLayer), QuantizeLayer)`` with a fused GPU forward/backward.
    Children, parameter names and ``str()`` are those of the plain ``Sequential`` it replaces."""

    @_hip.keeps_layout
    def forward(self, x):
        fast = self.__dict__.get("_qs_fast")
        if fast is not None:
            out = fast.try_run(self, x)
            if out is not _MISS:
                return out
            self.__dict__["_qs_fast"] = None
        inner, q = self[0], self[1]
        act, p = inner[0], inner[1]
        if _hooked(inner, act, p, q, p.callback, q.callback):
            return q(inner(x))
        # a plain, out-of-place nn.ReLU in front of an active quantizer is folded into the kernels: relu(x) is
        # never materialised (statistics, apply and backward read x itself); the gate of its backward rides in
        # the fused backward kernel
        fold, handle = _foldable_relu(act, x)
        q.__dict__["_qs_last_route"] = None
        if fold and q.is_active() and isinstance(x, torch.Tensor) and _eligible(p, q, x):
            if fold == 2:
                out = _with_owned_relu(x, lambda h: fused_prune_quantize(p, q, h, pre_relu=handle), handle)
            else:
                out = fused_prune_quantize(p, q, x, pre_relu=handle)
            self.__dict__["_qs_fast"] = _FastPair.arm(self, x, fold, handle)
            return out
        h = act_rider(act, x) if q.is_active() else None
        act_in = x if h is not None else None
        if h is None:
            h = act(x)
        if _eligible(p, q, h):
            # (under autocast an active site folds the identity: it then writes its image like a site behind a foldable activation)
            ident = identity_fold_handle(h) if q.is_active() else 0
            out = fused_prune_quantize(p, q, h, pre_relu=ident or False, act_in=act_in)
            if h is x:
                self.__dict__["_qs_fast"] = _FastPair.arm(self, x, 0, ident)
            elif isinstance(h, torch.Tensor) and not getattr(act, "inplace", False):
                self.__dict__["_qs_fast"] = _FastPair.arm(self, x, 3, ident)
            return out
        return q(p(h if act_in is None else _ActGrad.apply(act_in, h)))


FusedPruneQuantize.__name__ = "Sequential"   # keep str(model) identical to the reference's tree


def _quantizer_foldable(q: QuantizeLayer, x) -> bool:
    qc = q.callback
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() >= 1
            and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and type(qc) in (ScalerQuantizer, DecimalQuantizer) and qc.group_num <= 0 and q.channelwise == -1
            and not qc.backward_passthrough and q.batch_dimension == 0)


def fused_relu_quantize(q: QuantizeLayer, x: torch.Tensor, act: int = 1) -> torch.Tensor:
    """one step of ``q(relu(x))`` for an ACTIVE tensor-wise Scaler/Decimal quantizer without materialising
    relu(x): abs-max of max(x, 0) (qs_absmax pre_relu), running scale, y = Q(max(x, 0)), and a backward that
    applies the ReLU gate and the STE clamp in one pass.  Bookkeeping as QuantizeLayer.forward
    (reference quantize.py:482-517)."""
    qc = q.callback
    y = q.single_call_step(x, q._steps.read(q._n_updates), pre_relu=act)
    if y is not None:
        return y
    if q.training:
        t = q._steps.read(q._n_updates)
        if t == q.timeout and get_option("log_during_train"):
            logging.warn(f"quantizing {q.name} with {q.bits} bits")
        qc.__dict__["_bumped_step_counter"] = False
        new_weight = qc.optimize(x.detach(), q.bits, q.weight, batched=True, channel_index=-1, step_counter=q._n_updates,
                                 pre_relu=act)
        if new_weight is not None and new_weight is not q.weight:
            q.weight.data[:] = new_weight
        q._quantized = True
        if qc.__dict__.get("_bumped_step_counter", False):
            q._steps.note_device_add(q._n_updates, 1)
        else:
            q._steps.add(q._n_updates, 1)
    kind = "scaler" if isinstance(qc, ScalerQuantizer) else "decimal"
    return _FusedApply.apply(x, None, q.weight.data, kind, q.bits, 1 if qc.flip_axis else 0, True, act, qc.code_range(q.bits))


class FusedActQuantize(nn.Sequential):
    """``Sequential(act, QuantizeLayer)`` as convert builds it for a quantize-only activation site
    (reference convert.py:214-218): a plain out-of-place nn.ReLU in front of an active tensor-wise quantizer is
    folded into the quantizer's kernels (24 -> 16 B/elem per training step for bf16 activations); anything else runs
    module by module.  Children, parameter names and ``str()`` are those of the plain ``Sequential``."""

    @_hip.keeps_layout
    def forward(self, x):
        act, q = self[0], self[1]
        fold, handle = _foldable_relu(act, x)
        if fold and q.is_active() and _quantizer_foldable(q, x) and not _hooked(act, q, q.callback):
            if fold == 2:
                return _with_owned_relu(x, lambda h: fused_relu_quantize(q, h, handle), handle)
            return fused_relu_quantize(q, x, handle)
        if type(act) is nn.GELU and q.is_active() and _quantizer_foldable(q, x) and not _hooked(act, q, q.callback):
            # `act_backward`: ATen's GELU forward under no_grad, its backward in the quantizer's backward kernel (`_QuantStep(act_out=)`)
            h = act_rider(act, x)
            if h is not None:
                y = q.single_call_step(h, q._steps.read(q._n_updates), act_in=x) if q.initted else None
                return y if y is not None else q(_ActGrad.apply(x, h))
        return q(act(x))


FusedActQuantize.__name__ = "Sequential"


class FusedActPrune(nn.Sequential):
    """``Sequential(act, PruneLayer)`` as convert builds it for a prune-only activation site: a plain out-of-place
    nn.ReLU in front of an active channel-pruning layer is folded into its kernels -- importance of max(x, 0),
    y = max(x, 0) * mask, backward gate(x) * g * mask -- 20 -> 12 B/elem per training step for bf16
    activations.  Children, parameter names and ``str()`` are those of the plain ``Sequential``."""

    @_hip.keeps_layout
    def forward(self, x):
        act, p = self[0], self[1]
        fold, handle = _foldable_relu(act, x)
        if (fold and isinstance(x, torch.Tensor)
                and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and p.is_active()
                and type(p.callback) is MagnitudePruningCallback and not p.callback.l0
                and not p.callback.use_gradient and len(p.dimensions) == 1 and not _hooked(act, p.callback)):
            return _with_owned_relu(x, lambda h: p(h, pre_relu=handle), handle) if fold == 2 else p(x, pre_relu=handle)
        return p(act(x))


FusedActPrune.__name__ = "Sequential"


def _is_act_prune(m: nn.Module) -> bool:
    return len(m) == 2 and type(m[0]) in _FOLDABLE and isinstance(m[1], PruneLayer)


def _is_pair(m: nn.Module) -> bool:
    if len(m) != 2 or not isinstance(m[1], QuantizeLayer):
        return False
    inner = m[0]
    return (type(inner) in (nn.Sequential, FusedActPrune) and len(inner) == 2 and isinstance(inner[1], PruneLayer)
            and not isinstance(inner[0], (PruneLayer, QuantizeLayer)))


def _is_act_quantize(m: nn.Module) -> bool:
    return len(m) == 2 and (type(m[0]) in _FOLDABLE or type(m[0]) is nn.GELU) and isinstance(m[1], QuantizeLayer)


def fuse_prune_quantize_pairs(model: nn.Module) -> nn.Module:
    """re-class every convert-built prune->quantize pair and ReLU->quantize site in ``model`` (in place).
    Idempotent, and safe to call again after a further ``convert`` changed the tree (a site whose structure no
    longer matches goes back to a plain ``Sequential``)."""
    inner_of_pair = set()
    for m in model.modules():
        if type(m) not in (nn.Sequential, FusedPruneQuantize, FusedActQuantize, FusedActPrune):
            continue
        if _is_pair(m):
            m.__class__ = FusedPruneQuantize
            m[0].__class__ = nn.Sequential        # the pair runs its inner (act, prune) itself
            inner_of_pair.add(id(m[0]))
        elif id(m) in inner_of_pair:
            continue
        elif _is_act_quantize(m):
            m.__class__ = FusedActQuantize
        elif _is_act_prune(m):
            m.__class__ = FusedActPrune
        elif type(m) is not nn.Sequential:
            m.__class__ = nn.Sequential
    return model
