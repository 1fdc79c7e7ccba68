"""Options, naming, reductions-to-mask-shape, mask construction, checkpoint preload, logging.

API-compatible with the reference's qsparse/util.py; the two tensor functions
(``squeeze_tensor_to_shape`` util.py:79-99, ``calculate_mask_given_importance`` util.py:103-117) run as
HIP kernels for GPU tensors (``qs_mean_dim``, ``qs_kth_value`` + ``qs_mask_ge``).
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from qsparse_amd import _hip

_options_ = {"log_on_created": True, "log_during_train": True, "sync_statistics": None, "graph_safe": False, "preserve_dtype": False, "fold_relu": True,
             "elide_pruned": "forward", "relu_gate": True, "batch_weights": True, "autocast_image": True,
             "saturate": False, "act_backward": True}


_options_epoch = [0]       # bumped by every set_options call: cached per-site decisions that depend on an option are keyed on it


# The staged-mean kernels (csrc/qs_reduce.h) reproduce the summation ORDER of ATen's CPU SumKernel.cpp with one intra-op thread --
# cascade / 4-way row-sum / vectorised inner sum, chosen per output by TensorIterator's dimension order (`aten_reduce_plan`) -- as
# torch PINNED_TORCH computes it: that is the version the reference was run under when the golden fixtures were recorded
# (tests/golden/*.npz, meta["torch"]) and the one the bit-for-bit claims of DESIGN section 5 hold for.  Another torch may sum in
# another order: results stay within north_star's 1e-6 but masks can differ in a tie; tests/test_aten_contract*.py tell.
PINNED_TORCH = "2.10"
_torch_pin_warned = False


def check_torch_pin(version: Optional[str] = None) -> bool:
    """True when `version` (default: the running torch) is the pinned minor version; warns ONCE otherwise"""
    global _torch_pin_warned
    version = version or torch.__version__
    ok = ".".join(version.split("+")[0].split(".")[:2]) == PINNED_TORCH
    if not ok and not _torch_pin_warned:
        _torch_pin_warned = True
        import warnings
        warnings.warn(f"qsparse_amd: torch {version} is not the version ({PINNED_TORCH}) whose CPU summation order the staged-mean "
                      "kernels reproduce bit for bit; run tests/test_aten_contract.py (CPU) to see whether the order moved", stacklevel=2)
    return ok


def set_options(log_on_created: Optional[bool] = None, log_during_train: Optional[bool] = None,
                sync_statistics: Optional[bool] = None, graph_safe: Optional[bool] = None,
                preserve_dtype: Optional[bool] = None, fold_relu: Optional[bool] = None,
                elide_pruned: Optional[str] = None, relu_gate: Optional[bool] = None, batch_weights: Optional[bool] = None,
                autocast_image: Optional[bool] = None, saturate: Optional[bool] = None, act_backward: Optional[bool] = None):
    """update the global options; ``None`` leaves an option untouched (reference util.py:13-26).
    Exported as ``set_qsparse_options``.  ``sync_statistics`` (extension, default auto) controls the
    cross-rank exchange of mask/scale statistics under ``torch.distributed`` (see distributed.py; ``"always"``: also in a
    one-rank group; ``"mailbox"``: the prototype that moves the pair sites' records through peer-mapped mailboxes instead of
    an all-gather);
    ``graph_safe`` (extension, default False) makes GPU layers feed their running-mean counters to the kernels
    from device memory so that a training step can be captured into a hipGraph and replayed (see graphs.py);
    ``preserve_dtype`` (extension, default False) makes the Scaler/Decimal quantizers return the input's dtype
    instead of the reference's float32 promotion: the value is the float32 result rounded once, i.e. exactly
    what a following autocast convolution would consume, at 4 instead of 6 B/elem and without the cast pass;
    ``fold_relu`` (default True, bit-identical) lets a convert-built ``ReLU -> prune -> quantize`` site apply the
    ReLU inside the fused kernels instead of materialising its output; ``relu_gate`` (default True, bit-identical)
    lets the forward of such a site record the ReLU's gate as one bit per element, so that the backward reads the
    gradient and that bitmap instead of the gradient and the ReLU's input (which is then not kept for the backward);
    ``elide_pruned`` (extension): mask-aware traffic elision in the GPU kernels that carry a channel mask.  A pruned
    channel's input only ever meets ``* 0``, so it need not be loaded: ``"forward"`` (default) elides in the fused
    prune->quantize forward where that saves traffic -- NCHW activations, forwards that record no ReLU gate -- bit-identical
    for every finite input (a NaN / Inf on a PRUNED channel of such a tensor gives ``f32(0)*s`` instead of the reference's
    ``f32(INT_MIN)*s``; channels_last and gate-recording forwards load everything and follow the reference there too);
    ``"all"`` in every masked kernel, the backward and mask-apply ones included, which then write ``+0.0`` where the
    reference's ``g * 0`` / ``x * 0`` has ``-0.0`` (numerically equal);
    ``"off"`` loads everything (a NaN / Inf on a pruned channel then behaves exactly as in the reference);
    ``batch_weights`` (default True, bit-identical): the weight quantizers of a network built by ``convert`` are evaluated
    with three multi-tensor launches at the start of the root's forward instead of three per layer (see batch.py; a
    layer the forward never reaches is rolled back, so the state machines advance exactly as layer by layer);
    ``autocast_image`` (extension, default True since round 5; value-identical): under ``torch.autocast`` a fused ReLU -> prune -> quantize
    site hands the first convolution / linear that consumes its float32 output the bf16 image directly and takes that
    consumer's bf16 gradient as it is -- no ``fp32 <-> bf16`` cast passes around the site (see fused.py, "Autocast image");
    the site's output is then a ``torch.Tensor`` subclass;
    ``saturate`` (extension, default False): the Scaler / Decimal quantizers clamp their integer codes to the range of the
    bit width -- ``[-2^(bits-1) + notch, 2^(bits-1) - 1 + notch]``, or ``[0, 2^bits - 1]`` with ``use_uint`` -- which is what
    the reference's forward spells out but loses (its ``q.float().clamp_(...)`` acts on a temporary, quantize.py:56-62,
    110-116, so its tensor's largest element maps to code ``+2^(bits-1)``, one above the range).  Per quantizer:
    ``ScalerQuantizer(saturate=True)`` / ``quantize_with_scaler(..., saturate=True)``; ``None`` there follows this option.
    With it every code fits its bit width, so ``export_integer(...)`` yields int8 / packed int4 tensors directly;
    ``act_backward`` (default True, bit-identical): an ``nn.GELU()`` (erf form) that ``convert`` put in front of a prune -> quantize
    site keeps ATen's forward pass, but its backward -- ``gelu_backward(g, x)`` -- is evaluated by the site's backward kernel on the
    gradient it has in registers instead of by a pass of its own (8 -> 2 B/elem; ATen's GPU arithmetic reproduced bit for bit)."""
    _options_epoch[0] += 1
    if elide_pruned is not None:
        if elide_pruned not in ("off", "forward", "all"):
            raise ValueError(f"elide_pruned must be 'off', 'forward' or 'all', got {elide_pruned!r}")
        _hip.elide_mode = elide_pruned
        _options_["elide_pruned"] = elide_pruned
    for key, val in (("log_on_created", log_on_created), ("log_during_train", log_during_train),
                     ("sync_statistics", sync_statistics), ("graph_safe", graph_safe),
                     ("preserve_dtype", preserve_dtype), ("fold_relu", fold_relu), ("relu_gate", relu_gate),
                     ("batch_weights", batch_weights), ("autocast_image", autocast_image), ("saturate", saturate),
                     ("act_backward", act_backward)):
        if val is not None:
            _options_[key] = val


def get_option(key: str):
    """exported as ``get_qsparse_option`` (reference util.py:29-40)."""
    assert key in _options_, f"unknown qsparse option {key!r}"
    return _options_[key]


def nn_module(mod: nn.Module) -> nn.Module:
    """unwrap ``nn.DataParallel`` / DDP style wrappers (reference util.py:64-76)."""
    return mod.module if hasattr(mod, "module") else mod


def auto_name_prune_quantize_layers(net: nn.Module) -> nn.Module:
    """name every Prune/Quantize layer after its module path (reference util.py:43-60)."""
    from qsparse_amd.quantize import QuantizeLayer
    from qsparse_amd.sparse import PruneLayer

    for path, mod in net.named_modules():
        if isinstance(mod, (PruneLayer, QuantizeLayer)):
            mod.name = path
    return net


# ----------------------------------------------------------------------------------------------
# staged mean to the mask shape
# ----------------------------------------------------------------------------------------------
def _reduction_plan(xshape: Sequence[int], shape: Sequence[int]) -> List[int]:
    """dims to average, ascending; same error behaviour as the reference (util.py:92-98)."""
    assert len(xshape) == len(shape), "mismatch between the input tensor and mask"
    dims = []
    for i, (sx, sm) in enumerate(zip(xshape, shape)):
        if sx != sm:
            if sm != 1:
                raise ValueError("mismatch between the input tensor and mask")
            dims.append(i)
    return dims


def aten_reduce_plan(shape: Sequence[int], strides: Sequence[int], d: int):
    """How ATen's CPU ``x.mean(d, keepdim=True)`` walks a dense tensor of `shape` / `strides` (elements) -- the host half of
    qs_mean_strided.  TensorIterator (reorder_dimensions, coalesce_dimensions) puts the reduced dim first, sorts the kept dims by
    the stride they have in the contiguous result and merges neighbours that merge in both operands; SumKernel.cpp then picks by
    the strides of the two innermost dims: reduced dim unit-strided and >= 8 long -> vectorised inner sum; next dim unit-strided and
    >= 8 long -> vectorised outer sum (blocks of 32 of its coordinates in cascade order, the rest row-sum); else the scalar inner
    sum (row-sum) when the reduced dim has the smaller stride, the scalar outer sum (blocks of 4 in cascade order, rest row-sum)
    when not.  Returns (n, stride, kept, order, split_dim, split): kept = [(size, input stride, output stride)] sorted by input
    stride (the lane dim first), order / split_dim / split as include/qsparse_hip.h describes them.  Pinned against Tensor.mean on
    the CPU, one intra-op thread, in tests/test_aten_contract.py."""
    nd = len(shape)
    out_strides, acc = [0] * nd, 1
    for i in range(nd - 1, -1, -1):
        out_strides[i] = acc
        acc *= 1 if i == d else shape[i]
    out_strides[d] = 0
    live = [i for i in range(nd) if shape[i] != 1]
    size = [shape[i] for i in live]
    ins = [strides[i] for i in live]
    outs = [out_strides[i] for i in live]
    n = len(live)
    if d not in live:
        raise ValueError("the reduced dim has one element")

    def goes_after(a, b):          # TensorIterator::reorder_dimensions' should_swap(a, b): > 0 when a belongs outside b
        # operand 0, the output: a reduced dim (stride 0) goes inside every kept one; two kept dims by output stride
        if (outs[a] == 0) != (outs[b] == 0):
            return 1 if outs[b] == 0 else -1
        for st in (outs, ins):
            if st[a] == 0 or st[b] == 0:
                continue
            if st[a] != st[b]:
                return 1 if st[a] > st[b] else -1
            if size[a] > size[b]:
                return 1
        return 0

    perm = list(range(n - 1, -1, -1))
    for i in range(1, n):
        hi = i
        for lo in range(i - 1, -1, -1):
            c = goes_after(perm[lo], perm[hi])
            if c > 0:
                perm[lo], perm[hi] = perm[hi], perm[lo]
                hi = lo
            elif c < 0:
                break
    size, ins, outs = ([v[k] for k in perm] for v in (size, ins, outs))
    last = 0
    for k in range(1, n):          # coalesce_dimensions
        if size[last] * ins[last] == ins[k] and size[last] * outs[last] == outs[k]:
            size[last] *= size[k]
        else:
            last += 1
            size[last], ins[last], outs[last] = size[k], ins[k], outs[k]
    n = last + 1
    assert outs[0] == 0 and (n == 1 or outs[1] != 0)
    n0, s0 = size[0], ins[0]
    n1, s1 = (size[1], ins[1]) if n > 1 else (1, 0)
    if s0 == 1 and n0 >= 8:
        order, split = 0, 0
    elif s1 == 1 and n1 >= 8:
        order, split = 2, (n1 // 32) * 32
    elif s0 < s1:
        order, split = 1, 0
    else:
        order, split = 2, (n1 // 4) * 4
    if order == 2 and split == 0:      # no full block: row-sum for every output
        order = 1
    kept = sorted(range(1, n), key=lambda k: ins[k])
    split_dim = kept.index(1) if order == 2 else -1
    return n0, s0, [(size[k], ins[k], outs[k]) for k in kept], order, split_dim, split


_dense_any_order = _hip.dense_any_order


def split_view_plan(shape: Sequence[int], strides: Sequence[int], d: int, plan):
    """For a cascade / row-sum plan of `aten_reduce_plan` (order 2) whose cascade part is a PREFIX of the memory-contiguous columns
    behind the reduced dim -- the split dim is the outermost of the kept dims with a stride below the reduced one's: (order, pre, n,
    post, mr_cols), i.e. the permutation that makes the tensor a contiguous [pre, n, post] one in memory order and the number of
    leading columns of every slice that ATen sums in cascade order (qs_mean_dim_split); None when the plan has no such form.
    Pinned against Tensor.mean on the CPU in tests/test_aten_contract.py."""
    n0, s0, kept, kind, split_dim, split = plan
    if kind != 2 or not kept or kept[0][1] != 1:
        return None
    npost = sum(1 for k in kept if k[1] < s0)
    if split_dim != npost - 1:
        return None
    post = 1
    for k in kept[:npost]:
        post *= k[0]
    order = sorted(range(len(shape)), key=lambda i: (shape[i] != 1, -strides[i]))
    at = order.index(d)
    expect = 1
    for i in reversed(order):           # the memory-order view must be contiguous ...
        if shape[i] != 1:
            if strides[i] != expect:
                return None
            expect *= shape[i]
    # ... with every dim in front of the reduced one above its stride and every dim behind it below
    if any(shape[i] != 1 and strides[i] <= s0 for i in order[:at]) or any(shape[i] != 1 and strides[i] >= s0 for i in order[at + 1:]):
        return None
    numel = 1
    for v in shape:
        numel *= v
    return order, numel // (n0 * post), n0, post, split * (post // kept[split_dim][0])


def _record_for(record, slices: int):
    """the exchange-record buffer if the fused last-two-dims launch can fill it (one slice per channel)"""
    if record is None or record["buf"].numel() != 2 * slices:
        return None
    record["filled"] = True
    return record["buf"]


def _staged_mean_hip(x: torch.Tensor, dims: List[int], take_abs: bool, l0_flag=None, absmax_out=None,
                     absmax_channel_dim: Optional[int] = None, pre_relu: bool = False, record=None,
                     record_absmax=None) -> torch.Tensor:
    """successive keepdim means on the GPU, one ``qs_mean_dim`` launch per reduced dim.

    ``record`` (a dict with a float32 ``buf`` of 2C elements): when the plan ends in the fused last-two-dims launch over
    C slices, that launch also writes the rank's exchange record (means | values of ``record_absmax``, the per-channel
    abs-max accumulator that is complete by then) and ``record["filled"]`` is set."""
    cur = None
    first = True
    if x.dim() == 4 and dims and dims[0] == 0 and x.shape[0] > 1 and (absmax_out is None or absmax_channel_dim == 1):
        # channels_last activation whose batch dim is reduced first: no NCHW copy.  ATen's mean over N of such a tensor
        # returns an NCHW-contiguous result (summed in the order qs_mean_dim_cl reproduces, any channel count), so the
        # remaining stages are the usual NCHW ones.  (A batch of ONE has nothing to sum in that stage: ATen's result is the
        # NCHW copy of x and the later stages run in plain NCHW order -- the dense route below, bit for bit.)
        xm, _, like = _hip.mem_view(x, 1)
        if xm is not like:
            N, C, H, W = x.shape
            flags = _hip.mean_flags(take_abs, pre_relu)
            fusable = (dims == [0, 2, 3] and l0_flag is None and (H * W + W) * 4 <= _hip.LAST2_MAX_TILE_BYTES
                       and (flags & 0xff) in (0, _hip.MEAN_ABS, _hip.MEAN_ABS | _hip.MEAN_RELU)
                       and not (flags == 0 and absmax_out is not None))
            if fusable:       # stages 2 + 3 in one launch, which also folds the per-element maxima per channel
                stage, part = _hip.mean_dim_cl(xm, x.dtype, flags, absmax_out is not None)
                rec = _record_for(record, C)
                # (with a record and no riding abs-max the accumulator is only read, for the record's second half)
                acc = absmax_out if absmax_out is not None else (record_absmax if rec is not None else None)
                return _hip.mean_last2(stage, C, H, W, x.dtype, part, acc, rec).view(1, C, 1, 1)
            if absmax_out is None:
                flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
                stage, _ = _hip.mean_dim_cl(xm, torch.float32 if l0_flag is not None else x.dtype, flags, False, l0_flag=l0_flag)
                cur, dims, first = stage.view(1, C, H, W), dims[1:], False
    elif x.dim() == 5 and dims and dims[0] == 0 and x.shape[0] > 1 and absmax_out is None:
        # channels_last_3d: the same first stage with D*H*W positions per sample (ATen's mean over N of an NDHWC tensor is the
        # batch reduction of the [N, C, 1, D*H*W] channels_last tensor its memory also is; NCDHW-contiguous result)
        xm, _, like = _hip.mem_view(x, 1)
        if xm is not like:
            flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
            stage, _ = _hip.mean_dim_cl(xm, torch.float32 if l0_flag is not None else x.dtype, flags, False, l0_flag=l0_flag)
            cur, dims, first = stage.view((1,) + tuple(x.shape[1:])), dims[1:], False
    if cur is None and x.dim() == 4 and dims and dims[0] == 2 and absmax_out is None:
        # channels_last activation whose batch dim is NOT reduced (a per-sample mask -- or a batch of one, whose dim 0 equals
        # the mask's): ATen reduces H of the NHWC tensor directly, into an NCHW-contiguous result.  Its order for one sample
        # is exactly its order for the batch reduction of the [H, C, 1, W] channels_last tensor that sample's memory also is
        # (rows H apart by W*C elements, positions W, channels innermost; tests/test_aten_contract.py pins the identity on the
        # CPU), which qs_mean_dim_cl reproduces: one launch per sample, no NCHW copy; the W stage is the usual NCHW one
        xm, _, like = _hip.mem_view(x, 1)
        if xm is not like:
            N, C, H, W = x.shape
            flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
            odt = torch.float32 if l0_flag is not None else x.dtype
            rows = [_hip.mean_dim_cl(xm[i].view(H, 1, W, C), odt, flags, False, l0_flag=l0_flag)[0] for i in range(N)]
            cur = (rows[0] if N == 1 else torch.cat(rows)).view(N, C, 1, W)
            dims, first = dims[1:], False
    if cur is None and x.dim() == 4 and dims and dims[0] == 1 and absmax_out is None:
        # channels_last activation, batch dim kept, CHANNEL dim reduced first (a (N, 1, H, W) or (1, 1, H, W) mask at batch one):
        # for ATen that is the inner reduction of the [N*H*W, C] matrix the memory is, into an NCHW-contiguous result -- the
        # plain last-dim stage on the memory view, no copy (identity pinned in tests/test_aten_contract.py)
        xm, _, like = _hip.mem_view(x, 1)
        if xm is not like:
            N, C, H, W = x.shape
            flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
            kw = {"l0_flag": l0_flag} if l0_flag is not None else {}
            cur = _hip.mean_dim(xm.reshape(N * H * W, C), N * H * W, C, 1, torch.float32 if l0_flag is not None else x.dtype,
                                flags, **kw).view(N, 1, H, W)
            dims, first = dims[1:], False
    if cur is None and x.dim() == 4 and dims and dims[0] == 3 and x.shape[2] > 1 and x.shape[3] > 1 and absmax_out is None:
        # channels_last activation whose first (and then only) reduced dim is W -- a mask that keeps N, C and H: ATen sums the
        # strided rows with its scalar inner loop (row-sum order), not in the vectorised order of the contiguous NCHW row:
        # qs_mean_cl_w on the memory view, no copy (identity pinned in tests/test_aten_contract.py)
        xm, _, like = _hip.mem_view(x, 1)
        if xm is not like:
            N, C, H, W = x.shape
            flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
            cur = _hip.mean_cl_w(xm, torch.float32 if l0_flag is not None else x.dtype, flags, l0_flag).view(N, C, H, 1)
            dims, first = dims[1:], False
    if cur is None and dims and absmax_out is None and not x.is_contiguous() and x.numel() > 0:
        # any other layout (a transposed weight, a permuted activation, NDHWC with the batch kept ...): the reference's importance
        # is `x.abs()`, a dense tensor in x's stride order, which ATen reduces where it lies -- in the order aten_reduce_plan
        # derives, executed by qs_mean_strided on x itself (a view that is not dense is first copied into that dense layout, as
        # `x.abs()` would lay it out); the result is contiguous like ATen's, the later stages are the usual ones
        xs = x if _dense_any_order(x) else torch.empty_like(x).copy_(x)
        d = dims[0]
        plan = aten_reduce_plan(list(xs.shape), list(xs.stride()), d) if x.shape[d] > 1 else None
        flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
        odt = torch.float32 if l0_flag is not None else x.dtype
        if plan is not None and plan[3] == 0:
            # the reduced dim is the unit-stride one and ATen takes its vectorised inner sum: the memory-order view of x is a
            # contiguous tensor whose LAST dim is reduced -- the coalesced last-dim stage on that view; only the (n times smaller)
            # result is put into the logical order
            order = sorted((i for i in range(x.dim()) if i != d), key=lambda i: (xs.shape[i] != 1, -xs.stride(i))) + [d]
            mem = xs.permute(order)
            if mem.is_contiguous():
                kw = {"l0_flag": l0_flag} if l0_flag is not None else {}
                stage = _hip.mean_dim(mem, mem.numel() // x.shape[d], x.shape[d], 1, odt, flags, **kw)
                stage = stage.view(tuple(mem.shape[:-1]) + (1,)).permute([order.index(i) for i in range(x.dim())]).contiguous()
                cur, dims, first, plan = stage, dims[1:], False, None
        if plan is not None and plan[3] == 2 and plan[2][0][1] == 1 and xs.element_size() == 2:
            # a cascade / row-sum split whose cascade part is a PREFIX of the memory-contiguous columns behind the reduced dim (the
            # split dim is the outermost of them): the memory-order view is a contiguous [pre, n, post] tensor that the tuned
            # outer-reduction kernels take with that prefix named (qs_mean_dim_split: 16-byte loads, 8 columns per lane) -- a
            # [B, T, C] activation seen as [B, C, T] and reduced over B, a transposed matrix reduced over its strided dim
            # (2-byte dtypes only, measured: [256, 768, 196] bf16 82 -> 43 us, a transposed 4096 x 4096 bf16 matrix 477 -> 187; in float32
            # one lane per output already reads 256 bytes per wave and row and is the faster of the two, 31 against 43 us)
            view = split_view_plan(list(xs.shape), list(xs.stride()), d, plan)
            if view is not None and view[3] >= 64 and view[3] % 8 == 0:
                order, pre, n0, post, mr_cols = view
                mem = xs.permute(order)
                kw = {"l0_flag": l0_flag} if l0_flag is not None else {}
                stage = _hip.mean_dim(mem, pre, n0, post, odt, flags, mr_cols=mr_cols, **kw)
                shape = list(mem.shape)
                shape[order.index(d)] = 1
                stage = stage.view(shape).permute([order.index(i) for i in range(x.dim())]).contiguous()
                cur, dims, first, plan = stage, dims[1:], False, None
        if plan is not None and len(plan[2]) <= _hip.STRIDED_MAX_KEPT:
            stage = _hip.mean_strided(xs, plan, odt, flags, l0_flag)
            cur, dims, first = stage.view([1 if i == d else s for i, s in enumerate(x.shape)]), dims[1:], False
    if cur is None:
        cur = _hip.dense(x)
    shape = list(cur.shape)
    out_dtype = torch.float32 if l0_flag is not None else cur.dtype
    for pos, d in enumerate(dims):
        nd = len(shape)
        if (not first and len(dims) - pos == 2 and d == nd - 2 and dims[pos + 1] == nd - 1
                and (shape[d] * shape[d + 1] + shape[d + 1]) * 4 <= _hip.LAST2_MAX_TILE_BYTES):
            # the two trailing dims are what is left: one fused launch (same order, same rounding points)
            pre = 1
            for s in shape[:d]:
                pre *= s
            rec = _record_for(record, pre)
            cur = _hip.mean_last2(cur, pre, shape[d], shape[d + 1], cur.dtype, None, record_absmax if rec is not None else None, rec)
            shape[d] = shape[d + 1] = 1
            return cur.view(shape)
        pre = 1
        for s in shape[:d]:
            pre *= s
        post = 1
        for s in shape[d + 1:]:
            post *= s
        flags = 0
        kw = {}
        if first:
            flags = _hip.mean_flags(take_abs, pre_relu, l0_flag is not None)
            if l0_flag is not None:
                kw["l0_flag"] = l0_flag
            if absmax_out is not None:
                cd = absmax_channel_dim
                assert cd is not None and cd > d, "fused abs-max needs the channel dim inside the kept columns"
                chan_div = 1
                for s in shape[cd + 1:]:
                    chan_div *= s
                kw.update(absmax_out=absmax_out, chan_div=chan_div, C=shape[cd])
        cur = _hip.mean_dim(cur, pre, shape[d], post, out_dtype if first else cur.dtype, flags, **kw)
        shape[d] = 1
        cur = cur.view(shape)
        first = False
    if first and take_abs:  # nothing to reduce: importance is |x| itself (|act(x)| under a folded activation)
        cur = _hip.act_torch(pre_relu, cur).abs() if pre_relu else cur.abs()
    return cur


def squeeze_tensor_to_shape(x: torch.Tensor, shape: List[int]) -> torch.Tensor:
    """average ``x`` down to ``shape`` one dim at a time (reference util.py:79-99): every stage
    accumulates in fp32 and rounds to ``x``'s dtype, exactly like ``Tensor.mean`` on CPU."""
    dims = _reduction_plan(x.shape, shape)
    if _hip.on_hip(x):
        return _staged_mean_hip(x, dims, take_abs=False)
    for i in dims:
        if x.is_cuda and x.is_floating_point():
            # a GPU tensor of a dtype the kernels are not written for (float64): ATen's CPU mean is sum(x) -> div_(n); its device
            # kernels multiply by 1/n instead, a last-bit difference that decides the rounding whenever the exact mean is a
            # midpoint of the float32 state it ends in (float32-valued data in a float64 tensor: 0.2 % of the entries)
            x = _hip.true_div(x.sum(i, keepdim=True), x.shape[i])
        else:
            x = x.mean(i, keepdim=True)
    return x


# ----------------------------------------------------------------------------------------------
# mask from importance
# ----------------------------------------------------------------------------------------------
def threshold_rank(sparsity: float, n: int) -> int:
    """position, in ascending order, of the threshold element (reference util.py:115-116)."""
    return max(int(sparsity * n - 1), 0) + 1


def calculate_mask_given_importance(importance: torch.Tensor, sparsity: float) -> torch.Tensor:
    """binary mask keeping everything ``>=`` the ``threshold_rank``-th smallest importance
    (reference util.py:103-117).  Ties at the threshold are all kept."""
    n = importance.numel()
    k = threshold_rank(sparsity, n)
    if k >= n:
        raise IndexError(f"index {k} is out of bounds for dimension 0 with size {n}")
    if _hip.on_hip(importance):
        imp = importance.detach().to(torch.float32).contiguous()
        thr = _hip.kth_value(imp, k)
        mask = torch.empty(importance.shape, dtype=torch.bool, device=importance.device)
        _hip.mask_ge(imp, thr, mask)
        return mask
    ordered = importance.flatten().sort()[0]
    return importance >= ordered[k]


# ----------------------------------------------------------------------------------------------
# checkpoint preload
# ----------------------------------------------------------------------------------------------
def preload_qsparse_state_dict(model: nn.Module, state_dict: Dict[str, torch.Tensor]) -> nn.Module:
    """install checkpoint tensors into the (shape-less until first forward) parameters of every
    Prune/Quantize layer so that a following ``load_state_dict`` succeeds (reference util.py:120-145)."""
    from qsparse_amd.common import STATE_KEYS, state_parameter
    from qsparse_amd.quantize import QuantizeLayer
    from qsparse_amd.sparse import PruneLayer

    device = next(iter(model.parameters())).device
    for path, layer in model.named_modules():
        if not isinstance(layer, (PruneLayer, QuantizeLayer)):
            continue
        for subpath, sub in layer.named_modules():
            prefix = "".join(p + "." for p in (path, subpath) if p)
            for key, value in state_dict.items():
                leaf = key[len(prefix):]
                if key.startswith(prefix) and "." not in leaf:
                    make = state_parameter if leaf in STATE_KEYS else (lambda t: nn.Parameter(t, requires_grad=False))
                    sub._parameters[leaf] = make(value.to(device))
    return model


def extra_state_dict(model: nn.Module) -> Dict[str, Dict[str, int]]:
    """the two pieces of operator state the reference's ``state_dict`` forgets (quirk B7): ``QuantizeLayer._quantized``
    (a plain attribute, reference quantize.py:466,505 -- without it a freshly loaded network in ``eval()`` passes its inputs
    through un-quantized until one training step has run) and the quantizer callback's running-mean count ``t`` (a Python
    int, quantize.py:307,348 -- without it the running scale restarts at the first resumed step).  Extension:
    a SEPARATE dict keyed by module path, so ``state_dict()`` keeps the reference's schema key for key; store it next to the
    checkpoint and hand it to ``load_extra_state_dict`` after ``load_state_dict`` to resume bit-identically."""
    from qsparse_amd.quantize import QuantizeLayer
    return {path: {"quantized": int(bool(m._quantized)), "t": int(m.callback.t)}       # (paths of the unwrapped network:
            for path, m in nn_module(model).named_modules() if isinstance(m, QuantizeLayer)}   # no DataParallel / DDP prefix)


def load_extra_state_dict(model: nn.Module, extra: Dict[str, Dict[str, int]], strict: bool = True) -> nn.Module:
    """restore what ``extra_state_dict`` saved (a weight and a bias quantizer that share one callback, reference
    quantize.py:548,559-571, carry the same ``t``)."""
    from qsparse_amd.quantize import QuantizeLayer
    layers = {path: m for path, m in nn_module(model).named_modules() if isinstance(m, QuantizeLayer)}
    if strict and set(layers) != set(extra):
        missing, unexpected = sorted(set(layers) - set(extra)), sorted(set(extra) - set(layers))
        raise KeyError(f"extra state does not match the network: missing {missing}, unexpected {unexpected}")
    for path, m in layers.items():
        rec = extra.get(path)
        if rec is not None:
            m._quantized = bool(rec["quantized"])
            m.callback.t = int(rec["t"])     # (a device-resident copy of t notices the change by itself: device_t)
    return model


# ----------------------------------------------------------------------------------------------
# print-based logger (reference util.py:150-182)
# ----------------------------------------------------------------------------------------------
class style:
    RED, GREEN, YELLOW, RESET = "\033[31m", "\033[32m", "\033[33m", "\033[0m"


def _printer(color: str = ""):
    def emit(msg: str):
        print(f"{color}{msg}{style.RESET}" if color else msg)

    return emit


def log_functor(name: str, color: str = ""):
    """the reference's factory for the methods below (util.py:163): `name` is unused there as well."""
    return _printer(color)


class logging:
    """console logger with the reference's method names."""

    info = _printer()
    debug = _printer(style.GREEN)
    warn = warning = _printer(style.YELLOW)
    error = danger = exception = _printer(style.RED)
