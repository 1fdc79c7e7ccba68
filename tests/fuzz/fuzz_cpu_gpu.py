#!/usr/bin/env python3
"""Randomised differential test, CPU path vs HIP path of the SAME modules: random operator configurations (quantizer
kinds, channel-wise modes, mask dimension sets, pruning policies, schedules, wrapped Conv/Linear layers with weight and
bias operators) run a few training + evaluation steps on CPU tensors (the reference's own op sequence, pinned by the
golden fixtures) and on GPU tensors (the kernels); outputs, input/parameter gradients and the whole state_dict must
agree bit for bit.  Development tool; usage:  python3 tests/fuzz/fuzz_cpu_gpu.py [cases=150] [seed=0]"""
import copy
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same

VERBOSE = bool(os.environ.get("QS_FUZZ_ONLY"))
FORCE_WHAT = os.environ.get("QS_FUZZ_WHAT")      # e.g. "net": every case is of that kind (campaigns aimed at one route)
_same_bits = same


def same(a, b):
    """bitwise equality, except that a NaN only has to be a NaN (payload and sign of NaNs depend on the instruction
    sequence, e.g. x86's negative "real indefinite")"""
    if _same_bits(a, b):
        return True
    if a.shape != b.shape or a.dtype != b.dtype or not a.is_floating_point():
        return False
    na, nb = torch.isnan(a), torch.isnan(b)
    if not na.any() or not torch.equal(na, nb):
        return False
    return _same_bits(torch.where(na, torch.zeros_like(a), a), torch.where(nb, torch.zeros_like(b), b))


class WeightReader(nn.Module):
    """a network whose forward only READS its layers' parameters through their operators (as Conv2d.forward / Linear.forward
    do: weight, then bias) -- the convolutions themselves round differently on the two devices and are not what is under test.
    `skip`: a layer this forward does not reach (a branch not taken: the multi-tensor path rolls that layer back)."""

    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def forward(self, skip=-1):
        read = []
        for i, layer in enumerate(self.layers):
            if i == skip:
                continue
            read.append(("w%d" % i, layer.weight))
            b = layer.bias
            if b is not None:
                read.append(("b%d" % i, b))
        return read


def make_quantizer(rng):
    kind = rng.choice(["scaler", "decimal", "adaptive"])
    kw = dict(flip_axis=rng.random() < 0.2, backward_passthrough=rng.random() < 0.15)
    cb = {"scaler": qs.ScalerQuantizer, "decimal": qs.DecimalQuantizer, "adaptive": qs.AdaptiveQuantizer}[kind](**kw)
    return kind, cb


def build_token_major(rng):
    """activation sites whose channel dim is not dim 1: token-major (B, T, C) and 5-d activations, masks over {last}, {1, 2}, {1},
    ... (the reference builds the mask for any dim set, sparse.py:231-239, and averages dim by dim, util.py:92-99), tensor-wise or
    last-dim channel-wise quantizers (quantize.py:100-107; batch of one for Scaler / Decimal, :341-343) -- as lone operators, as the
    pair, and behind an activation module as `convert` builds them.  (B, T, C) + dimensions={2} is the composite's layout 3."""
    dtype = rng.choice([torch.float32, torch.bfloat16, torch.bfloat16, torch.float16])
    rank = rng.choice([3, 3, 3, 5])
    if rank == 3:
        shape = (rng.choice([1, 2, 4, 8, 17, 64]), rng.choice([1, 2, 5, 16, 49, 197]), rng.choice([2, 6, 16, 33, 64, 96, 256]))
        dims = rng.choice([{2}, {2}, {2}, {1, 2}, {1}, {0, 2}, {0, 1, 2}])
    else:
        shape = (rng.choice([1, 2, 4]), rng.choice([2, 6, 16]), rng.choice([1, 3, 4]), rng.choice([2, 5]), rng.choice([3, 8]))
        dims = rng.choice([{1}, {4}, {1, 2}, {0, 1}])
    last = rank - 1
    bits = rng.choice([2, 4, 8])
    timeout = rng.choice([0, 1, 2])
    kind, qcb = make_quantizer(rng)
    start, interval, rep = rng.choice([0, 1, 2]), rng.choice([1, 2]), rng.choice([1, 2])
    sparsity = rng.choice([0.3, 0.5, 0.75])
    site = rng.choice(["pair", "pair", "act_p", "act_q"])
    act = rng.choice(["relu", "relu", "relu6", "leaky", "identity", "identity"])
    inplace = act != "identity" and rng.random() < 0.3
    if act == "leaky" and dtype == torch.float16:
        dtype = torch.bfloat16               # (ATen's own fp16 leaky_relu_backward differs between its CPU and GPU kernels)
    cw = rng.choice([-1, -1, -1, last])
    if site == "pair":
        cw = -1 if rng.random() < 0.8 else cw
    if cw >= 0 and (kind != "adaptive" or rng.random() < 0.7):
        # batched channel-wise Scaler / Decimal raises in the reference (quantize.py:341-343), and so does the Adaptive quantizer on
        # a channel dim other than 1 (:398-401 `.view` a transposed tensor): mostly a batch of one; the rest must raise alike
        shape = (1,) + shape[1:]
    policy = rng.choice(["default", "default", "no_avg", "refresh", "l0"])
    cbkw = {"default": {}, "no_avg": dict(running_average=False), "refresh": dict(mask_refresh_interval=2, stop_mask_refresh=4),
            "l0": dict(l0=True)}[policy]
    desc = dict(what="tok", dtype=str(dtype)[6:], shape=shape, dimensions=sorted(dims), bits=bits, timeout=timeout, quantizer=kind,
                start=start, interval=interval, rep=rep, sparsity=sparsity, site=site, act=act, inplace=inplace, channelwise=cw,
                policy=policy)
    make_act = {"relu": lambda: nn.ReLU(inplace=inplace), "relu6": lambda: nn.ReLU6(inplace=inplace),
                "leaky": lambda: nn.LeakyReLU(0.1, inplace=inplace), "identity": lambda: nn.Identity()}[act]

    def factory():
        net = nn.Sequential(make_act())
        types = [type(net[0])]
        if site in ("pair", "act_p"):
            net = qs.convert(net, qs.prune(sparsity=sparsity, dimensions=dims, start=start, interval=interval, repetition=rep,
                                           callback=qs.MagnitudePruningCallback(**cbkw)), activation_layers=types, log=False)
        if site in ("pair", "act_q"):
            net = qs.convert(net, qs.quantize(bits=bits, channelwise=cw, timeout=timeout, callback=copy.deepcopy(qcb)),
                             activation_layers=types, log=False)
        return net
    return desc, factory, shape, dtype


def build(rng):
    """returns (description, module factory, input shape, dtype)"""
    what = rng.choice(["act_q", "act_q", "act_p", "act_p", "act_pq", "conv", "linear", "site", "site", "site", "net", "net", "tok", "tok"])
    if FORCE_WHAT:
        what = FORCE_WHAT
    if what == "tok":
        return build_token_major(rng)
    dtype = rng.choice([torch.float32, torch.float32, torch.bfloat16, torch.bfloat16, torch.float16])
    if rng.random() < 0.08 and what in ("act_q", "act_p", "act_pq"):
        # a dtype the kernels are not written for: the GPU side evaluates the package's ATen expression on the device (_hip.on_hip)
        dtype = torch.float64
    if what in ("conv", "linear", "net"):
        dtype = torch.float32
    n = rng.choice([1, 2, 4, 8, 16, 17, 48, 64, 130, 256, 300])
    c = rng.choice([2, 4, 6, 16, 33, 64, 96, 256])
    hw = rng.choice([(1, 1), (3, 3), (7, 7), (8, 8), (5, 6), (14, 14), (16, 16), (28, 28)])
    if what in ("conv", "linear"):
        n, c, hw = min(n, 8), min(c, 33), hw if hw[0] <= 8 else (8, 8)
    while n * c * hw[0] * hw[1] > 1_500_000 and n > 1:
        n = max(1, n // 2)
    shape = (n, c) + hw
    bits = rng.choice([2, 4, 8, 4, 8, 1, 3, 6, 12, 16])
    timeout = rng.choice([0, 1, 2])
    kind, qcb = make_quantizer(rng)
    start, interval, rep = rng.choice([0, 1, 2]), rng.choice([1, 2]), rng.choice([1, 2])
    sparsity = rng.choice([0.3, 0.5, 0.75])
    desc = dict(what=what, dtype=str(dtype)[6:], shape=shape, bits=bits, timeout=timeout, quantizer=kind, start=start,
                interval=interval, rep=rep, sparsity=sparsity)

    def pcb():
        policy = rng.choice(["default", "default", "no_avg", "l0", "grad", "uniform", "refresh"])
        desc["policy"] = policy
        if policy == "no_avg":
            return qs.MagnitudePruningCallback(running_average=False)
        if policy == "l0":
            return qs.MagnitudePruningCallback(l0=True)
        if policy == "grad":
            return qs.MagnitudePruningCallback(use_gradient=True)
        if policy == "uniform":
            return qs.UniformPruningCallback()
        if policy == "refresh":
            return qs.MagnitudePruningCallback(mask_refresh_interval=2, stop_mask_refresh=4)
        return qs.MagnitudePruningCallback()

    if what == "net":
        # `convert(model, prune(...), weight_layers=[Conv2d, Linear])` -- then, mostly, `convert(model, quantize(...), ...)` -- over a
        # small network (reference convert.py:120-197 wraps every layer; imitation.py:61-68 reads through the operators): on the GPU
        # the DEFAULT route of such a network is the multi-tensor table (batch.py: qs_multi_stage_mean / magnitude / mask_refresh /
        # absmax / scale_update / quant_fwd / ste_bwd), installed by convert itself
        dims = rng.choice([{0, 1, 2, 3}, {0, 1, 2, 3}, {1}, {1}, {0}, {0, 1}])
        policy = rng.choice(["default", "default", "no_avg", "refresh", "no_avg_refresh", "freeze"])
        cbkw = {"default": {}, "no_avg": dict(running_average=False), "refresh": dict(mask_refresh_interval=2, stop_mask_refresh=4),
                "no_avg_refresh": dict(running_average=False, mask_refresh_interval=2),
                "freeze": dict(running_average=False, mask_refresh_interval=2, stop_mask_refresh=3)}[policy]
        with_quant = rng.random() < 0.75
        # (quantize-only networks reach the steady-state fast path of the weight path, batch._Steady: every tensor quantized on every
        #  read, the launch table re-issued after identity checks -- with the roll-backs of the skipped layers on top)
        with_prune = not with_quant or rng.random() < 0.8
        cw = rng.choice([-1, 0, 1, 1])
        bias_bits = rng.choice([-1, -1, 8])
        c1, c2, c3 = rng.choice([4, 8, 12, 33]), rng.choice([4, 8, 16]), rng.choice([3, 8, 10])
        c0 = min(c, 33)
        desc.update(dimensions=sorted(dims), policy=policy, with_quant=with_quant, with_prune=with_prune, channelwise=cw,
                    bias_bits=bias_bits, widths=(c0, c1, c2, c3))

        def net_factory():
            torch.manual_seed(1234)
            net = WeightReader([nn.Conv2d(c0, c1, 3, padding=1), nn.Conv2d(c1, c2, 3, bias=False), nn.Conv2d(c2, c2, 1),
                                nn.Conv2d(c2, c2, (1, 3)), nn.Linear(c2, c3)])
            if with_prune:
                net = qs.convert(net, qs.prune(sparsity=sparsity, dimensions=dims, start=start, interval=interval, repetition=rep,
                                               callback=qs.MagnitudePruningCallback(**cbkw)),
                                 weight_layers=[nn.Conv2d, nn.Linear], log=False)
            if with_quant:
                net = qs.convert(net, qs.quantize(bits=bits, channelwise=cw, timeout=timeout, callback=copy.deepcopy(qcb), bias_bits=bias_bits),
                                 weight_layers=[nn.Conv2d, nn.Linear], log=False)
            return net
        return desc, net_factory, (1,), dtype
    if what == "site":
        # a `convert`-built activation site (reference convert.py:199-229): activation module -> PruneLayer -> QuantizeLayer, the
        # activation folded into the kernels on the GPU (nn.ReLU / ReLU6 / Hardtanh / LeakyReLU, in place or not), composite or
        # fine-grained route, optional code saturation, masks that freeze (policy "refresh")
        act = rng.choice(["relu", "relu", "relu6", "hardtanh", "hardtanh_odd", "leaky", "identity"])
        inplace = act != "identity" and rng.random() < 0.35
        if act == "leaky" and dtype == torch.float16:
            # ATen's own fp16 leaky_relu_backward differs between its CPU and GPU kernels (a subnormal product on a rounding tie,
            # -0.0 * slope; tools/_leaky_check.py) -- and it is ATen's kernel wherever a site is not fused (Adaptive quantizer, ...)
            dtype = torch.bfloat16
            desc["dtype"] = "bfloat16"
        site = rng.choice(["pair", "pair", "act_q", "act_p"])
        saturate = kind != "adaptive" and rng.random() < 0.25
        if saturate:
            qcb.saturate = True
        flat = rng.random() < 0.2                # a 2-d activation (behind an nn.Linear)
        if flat:
            shape = shape[:2]
        cb = pcb()
        desc.update(act=act, inplace=inplace, site=site, saturate=saturate, shape=shape)
        make_act = {"relu": lambda: nn.ReLU(inplace=inplace), "relu6": lambda: nn.ReLU6(inplace=inplace),
                    "hardtanh": lambda: nn.Hardtanh(-0.75, 1.5, inplace=inplace),
                    "hardtanh_odd": lambda: nn.Hardtanh(0.1, 0.7, inplace=inplace),       # bounds bf16 / fp16 cannot represent
                    "leaky": lambda: nn.LeakyReLU(0.1, inplace=inplace),
                    "identity": lambda: nn.Identity()}[act]

        def site_factory():
            net = nn.Sequential(make_act())
            types = [type(net[0])]
            if site in ("pair", "act_p"):
                net = qs.convert(net, qs.prune(sparsity=sparsity, dimensions={1}, start=start, interval=interval, repetition=rep,
                                               callback=copy.deepcopy(cb)), activation_layers=types, log=False)
            if site in ("pair", "act_q"):
                net = qs.convert(net, qs.quantize(bits=bits, channelwise=-1, timeout=timeout, callback=copy.deepcopy(qcb)),
                                 activation_layers=types, log=False)
            return net
        return desc, site_factory, shape, dtype
    if what == "act_q":
        cw = rng.choice([-1, -1, 1])
        if cw == 1 and kind != "adaptive":
            shape = (1,) + shape[1:]          # batched channel-wise Scaler/Decimal raises in the reference
        desc.update(channelwise=cw, shape=shape)
        return desc, (lambda: qs.quantize(bits=bits, channelwise=cw, timeout=timeout, callback=copy.deepcopy(qcb))), shape, dtype
    if what == "act_p":
        dims = rng.choice([{1}, {1}, {0, 1}, {2, 3}, {1, 2, 3}, {0, 1, 2, 3}, {0}, {0, 1, 2}])
        desc.update(dimensions=sorted(dims))
        cb = pcb()
        return desc, (lambda: qs.prune(sparsity=sparsity, dimensions=dims, start=start, interval=interval, repetition=rep,
                                       callback=copy.deepcopy(cb))), shape, dtype
    if what == "act_pq":
        cb = pcb()
        return desc, (lambda: nn.Sequential(qs.prune(sparsity=sparsity, dimensions={1}, start=start, interval=interval, repetition=rep,
                                                     callback=copy.deepcopy(cb)),
                                            qs.quantize(bits=bits, channelwise=-1, timeout=timeout, callback=copy.deepcopy(qcb)))), shape, dtype
    cw = rng.choice([-1, 0, 0])
    bias_bits = rng.choice([-1, 8, 12])
    dims = rng.choice([{0, 1, 2, 3}, {1}, {0}, {0, 1}]) if what == "conv" else rng.choice([{0, 1}, {1}, {0}])
    with_prune = rng.random() < 0.65       # quantize-only layers are what the multi-tensor weight path takes
    if not with_prune:                     # (tensor-wise, no bias quantizer: make those common among them)
        cw, bias_bits = rng.choice([-1, -1, 0]), rng.choice([-1, -1, 8])
    desc.update(channelwise=cw, bias_bits=bias_bits, dimensions=sorted(dims), with_prune=with_prune)
    cb = pcb()
    cout = rng.choice([4, 8, 12])

    def factory():
        torch.manual_seed(1234)
        base = nn.Conv2d(c, cout, 3, padding=1) if what == "conv" else nn.Linear(c, cout)
        m = base
        if with_prune:
            m = qs.prune(base, sparsity=sparsity, dimensions=dims, start=start, interval=interval, repetition=rep, callback=copy.deepcopy(cb))
        return qs.quantize(m, bits=bits, channelwise=cw, timeout=timeout, callback=copy.deepcopy(qcb), bias_bits=bias_bits)

    if what == "linear":
        shape = (n, c)
        desc["shape"] = shape
    return desc, factory, shape, dtype


def functional_case(rng, idx):
    """one random call of the functional API on both devices: (description, callable(device) -> list of tensors)"""
    from qsparse_amd.quantize import quantize_with_decimal, quantize_with_line, quantize_with_scaler
    from qsparse_amd.sparse import apply_mask
    from qsparse_amd.util import calculate_mask_given_importance, squeeze_tensor_to_shape

    g = torch.Generator().manual_seed(9000 + idx)
    nd = rng.choice([1, 2, 3, 4, 4, 5])
    shape = tuple(rng.choice([1, 2, 3, 5, 8, 16, 31, 64]) for _ in range(nd))
    while int(np.prod(shape)) > 600_000:
        shape = tuple(max(1, d // 2) for d in shape)
    dtype = rng.choice([torch.float32, torch.bfloat16, torch.float16])
    x = (torch.randn(shape, generator=g) * rng.choice([0.01, 1.0, 30.0])).to(dtype)
    fn = rng.choice(["scaler", "decimal", "line", "squeeze", "mask", "apply_mask"])
    if rng.random() < 0.25 and x.numel() >= 4 and fn in ("scaler", "decimal", "line", "apply_mask"):
        specials = torch.tensor([float("nan"), float("inf"), float("-inf"), 3e38]).to(dtype)   # non-finite and huge inputs
        x.view(-1)[:4] = specials
        desc_special = True
    else:
        desc_special = False
    cl = nd == 4 and rng.random() < 0.5 and fn != "squeeze"
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    ci = rng.choice([-1] + list(range(nd)))
    bits = rng.choice([2, 4, 8])
    desc = dict(i=idx, fn=fn, shape=shape, dtype=str(dtype)[6:], ci=ci, bits=bits, channels_last=cl, specials=desc_special)
    C = shape[ci] if ci >= 0 else 1

    if fn in ("scaler", "decimal"):
        form = rng.choice(["tensor", "tensor", "float", "zerodim"]) if ci < 0 else "tensor"
        desc["param"] = form
        if fn == "scaler":
            p = torch.rand(C, 1, generator=g) * 0.2 + 0.01
        else:
            p = torch.randint(0, 9, (C, 1), generator=g).float()
        if form == "float":
            p = float(p.view(-1)[0]) if fn == "scaler" else int(p.view(-1)[0])
        elif form == "zerodim":
            p = p.view(-1)[0].clone()
        f = quantize_with_scaler if fn == "scaler" else quantize_with_decimal
        gout = torch.randn(shape, generator=g)
        flip, passthrough = rng.random() < 0.2, rng.random() < 0.2

        def call(dev):
            xd = x.detach().clone().to(dev).requires_grad_(True)
            pd = p.to(dev) if isinstance(p, torch.Tensor) else p
            y = f(xd, bits, pd, ci, False, passthrough, flip)
            y.backward(gout.to(dev).to(y.dtype).view(y.shape) if y.numel() == gout.numel() else torch.ones_like(y))
            return [y.detach().cpu(), xd.grad.cpu()]
        return desc, call
    if fn == "line":
        lo = -torch.rand(C, 1, generator=g) * 2
        hi = torch.rand(C, 1, generator=g) * 2 + rng.choice([0.0, 0.05])
        lines = torch.cat([lo, hi], 1)
        fzp = rng.random() < 0.5
        desc["fzp"] = fzp
        return desc, (lambda dev: [quantize_with_line(x.to(dev), bits, lines.to(dev), ci, False, fzp).cpu()])
    if fn == "squeeze":
        tgt = [d if rng.random() < 0.5 else 1 for d in shape]
        desc["target"] = tgt
        return desc, (lambda dev: [squeeze_tensor_to_shape(x.to(dev), tgt).cpu()])
    if fn == "mask":
        imp = x.float().flatten()
        if rng.random() < 0.5:
            imp = (imp * 4).round() / 4          # ties
        sp = rng.choice([0.0, 0.3, 0.5, 0.75, 0.95])
        desc["sparsity"] = sp
        return desc, (lambda dev: [calculate_mask_given_importance(imp.to(dev), sp).cpu()])
    mshape = [d if rng.random() < 0.5 else 1 for d in shape]
    mask = torch.rand(mshape, generator=g) > 0.4
    gout = torch.randn(shape, generator=g).to(dtype)
    desc["mask_shape"] = mshape

    def call(dev):
        xd = x.detach().clone().to(dev).requires_grad_(True)
        y = apply_mask(xd, mask.to(dev))
        y.backward(gout.to(dev))
        return [y.detach().cpu(), xd.grad.cpu()]
    return desc, call


def one_functional(rng, idx, dry=False):
    desc, call = functional_case(rng, idx)
    if dry:
        return None
    if VERBOSE:
        print(desc, flush=True)
    res = {}
    for dev in ("cpu", "cuda"):
        try:
            res[dev] = call(dev)
        except Exception as e:      # noqa: BLE001
            res[dev] = ("raised", type(e).__name__, str(e)[:120])
    a, b = res["cpu"], res["cuda"]
    if isinstance(a, tuple) or isinstance(b, tuple):
        if isinstance(a, tuple) and isinstance(b, tuple) and a[1] == b[1]:
            return "ok"
        return dict(desc, cpu=a if isinstance(a, tuple) else "ran", gpu=b if isinstance(b, tuple) else "ran")
    for k, (va, vb) in enumerate(zip(a, b)):
        if not same(va, vb):
            return dict(desc, mismatch=k, cpu=(tuple(va.shape), str(va.dtype)), gpu=(tuple(vb.shape), str(vb.dtype)),
                        max_abs=float((va.float() - vb.float()).abs().max()) if va.shape == vb.shape and va.numel() else None)
    return "ok"


EXCHANGE = bool(os.environ.get("QS_FUZZ_EXCHANGE"))      # run inside a one-rank RCCL process group
GRAPH = bool(os.environ.get("QS_FUZZ_GRAPH"))            # steady-state steps of the site cases replayed from a hipGraph
GRAPHED = [0]
COLLECTIVES = [0]
ENGAGED = [0]      # cases in which the multi-tensor weight path actually took the layer
STEADY = [0]       # ... and in which its steady-state fast path (batch._Steady) was armed at some step


def run(factory, shape, dtype, device, seed, steps, eval_from, weight_mode=False, channels_last=False, batcher=False, twin=False,
        site=False, nonfinite=None, graph=False, permute=None):
    np.random.seed(seed)
    torch.manual_seed(seed)
    m = factory().to(device)
    # the multi-tensor weight path on the GPU side (a no-op for layers it does not take): same results as the inline one
    wb = qs.WeightBatcher(m) if (batcher and weight_mode and device == "cuda") else None
    if wb is not None and wb.layers:
        ENGAGED[0] += 1
    g = torch.Generator().manual_seed(seed)
    outs = []
    if isinstance(m, WeightReader):
        if channels_last:        # (before the first forward: full-shape masks and magnitudes are created in their weight's layout)
            m = m.to(memory_format=torch.channels_last)
        if device == "cuda" and len(getattr(m.__dict__.get("_qs_weight_batcher"), "layers", ())) > 0:
            ENGAGED[0] += 1
        steady_seen = False
        for s in range(steps):
            m.train(s < eval_from)
            for p in m.parameters():
                p.grad = None
            if nonfinite is not None and s == nonfinite[1]:      # a diverged weight: a NaN / Inf in a raw parameter from here on
                raw = m.layers[seed % len(m.layers)]._parameters["weight"]
                with torch.no_grad():
                    raw.view(-1)[(seed * 7919) % raw.numel()] = nonfinite[0]
            # (under an initialised process group a rolled-back layer's parameter receives ZEROS, what DistributedDataParallel itself
            #  uses for an unused parameter -- batch.py `_GroupSte` -- instead of no gradient: no skipped layers in that mode)
            skip = (seed + s) % len(m.layers) if (s % 3 == 1 and not EXCHANGE) else -1
            read = m(skip)
            grads = [torch.randn(t.shape, generator=g) * 3 for _, t in read]
            for k, t in read:
                outs.append((k, t.detach().cpu().clone()))
            live = [(t, gr.to(device)) for (_, t), gr in zip(read, grads) if t.requires_grad]
            if live and m.training:
                torch.autograd.backward([t for t, _ in live], [gr for _, gr in live])
            for name, p in m.named_parameters():
                if p.grad is not None:
                    outs.append(("grad:" + name, p.grad.cpu().contiguous()))
            with torch.no_grad():                # seeded pseudo-update of the raw parameters (identical on both devices)
                for name, p in m.named_parameters():
                    if p.requires_grad:
                        p.add_((torch.randn(p.shape, generator=g) * 0.05).to(device))
            wb_ = m.__dict__.get("_qs_weight_batcher")
            if device == "cuda" and wb_ is not None and getattr(wb_, "_steady", None) is not None and not steady_seen:
                steady_seen = True
                STEADY[0] += 1
        for k, v in m.state_dict().items():
            outs.append(("state:" + k, v.detach().cpu().contiguous()))
        return outs
    captured = None         # (graph, static x, static gradient, static y): whole steps replayed from a hipGraph (graphs.py)
    y_dtype = None
    for s in range(steps):
        m.train(s < eval_from)
        if captured is not None and not m.training:
            qs.graphs.resync_host_state(m)      # back to eager: the host mirrors re-read the device state
            captured = None
        if weight_mode:
            # read the layer's weight (and bias) through its operators, as its forward does, without the convolution
            # itself (whose CPU and GPU algorithms round differently)
            for p in m.parameters():
                p.grad = None
            if nonfinite is not None and s == nonfinite[1]:      # a diverged weight: a NaN / Inf in the raw parameter from here on
                raw = m._parameters["weight"]
                with torch.no_grad():
                    raw.view(-1)[(seed * 7919) % raw.numel()] = nonfinite[0]
            if wb is not None:
                wb._precompute(m, ())       # what the forward pre-hook does
            w = m.weight
            gw = torch.randn(w.shape, generator=g)
            outs.append(("w", w.detach().cpu().clone()))
            if twin and m.training:     # a second read before the first one's backward (siamese recipes): the statistics move on,
                w2 = m.weight           # each read's backward clamps with what the reference's Function saved for it
                outs.append(("w2", w2.detach().cpu().clone()))
                (w2 * 1.0).backward(torch.randn(w2.shape, generator=g).to(device) * 3, retain_graph=True)
            b = m.bias
            if b is not None and b.requires_grad:
                (w * 1.0).backward(gw.to(device), retain_graph=True)
                gb = torch.randn(b.shape, generator=g)
                outs.append(("b", b.detach().cpu().clone()))
                b.backward(gb.to(device))
            else:
                w.backward(gw.to(device))
            for name, p in m.named_parameters():
                if p.grad is not None:
                    outs.append(("grad:" + name, p.grad.cpu()))
            continue
        x = (torch.randn(shape, generator=g) * (3.0 if site else 1.5)).to(dtype)
        if dtype == torch.float64:
            # genuinely double-precision data.  (float32-VALUED doubles put the exact mean of a few elements on a rounding
            # midpoint of the float32 state it ends in for ~0.1 % of the entries, where the last float64 bit -- the device's
            # summation order, which the ATen-on-device route does not control -- decides: a 1-ulp float32 difference, inside
            # north_star's 1e-6 and documented in DESIGN section 5, but not what a bit-for-bit harness can hold.)
            x = x * 1.0000000001234567
        x.view(-1)[:2] = torch.tensor([0.0, -0.5]).to(dtype)
        if site and x.numel() >= 8:             # the boundary values of the activations' gates
            x.view(-1)[2:8] = torch.tensor([6.0, -0.75, 1.5, 7.5, -3.0, 1e-30]).to(dtype)
            if x.numel() >= 12:
                x.view(-1)[8:12] = torch.tensor([0.1, 0.7, 0.69921875, 0.10009765625]).to(dtype)
        x[x == 0] = 0.0             # no -0.0 (see tests/fuzz/fuzz_parity.py)
        if nonfinite is not None and s >= nonfinite[1]:     # a NaN / Inf somewhere -- on a kept or a pruned channel, whichever
            x.view(-1)[(seed * 7919 + s * 31) % x.numel()] = nonfinite[0]
        if channels_last and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
        if permute is not None:      # the same values, dense in memory in another dim order (a transposed / permuted activation)
            x = x.permute(permute).contiguous().permute([permute.index(i) for i in range(x.dim())])
        if (graph and device == "cuda" and captured is None and m.training and not twin and s >= 2 and y_dtype is not None
                and qs.graphs.steady_state(m)):
            sx = torch.empty_like(x.to(device)).requires_grad_(True)
            sg = torch.empty_like(sx, dtype=y_dtype).detach()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                sy = m(sx.clone() if site else sx)
                sy.backward(sg)
            captured = (gr, sx, sg, sy)
            GRAPHED[0] += 1
        if captured is not None:        # this step is a replay of the captured one on this step's data
            gr, sx, sg, sy = captured
            gout = torch.randn(sy.shape, generator=g).to(sy.dtype)
            with torch.no_grad():
                sx.copy_(x.to(device))
                sg.copy_(gout.to(device))
            gr.replay()
            outs.append(("layout", torch.tensor([st for st, n in zip(sy.stride(), sy.shape) if n > 1])))
            outs.append(("y", sy.detach().cpu().clone()))
            outs.append(("gx", sx.grad.cpu().clone()))
            continue
        xd = x.to(device).requires_grad_(True)
        y = m(xd.clone() if site else xd)       # (an in-place activation needs a non-leaf input, as behind a convolution)
        y_dtype = y.dtype
        if twin and m.training:         # a second forward before the first one's backward
            # (float64: stay genuinely double-precision, see above -- a float32 round trip would put the means on midpoints again)
            x2 = ((x.detach() * 4.000000000345) if dtype == torch.float64 else (x.detach().float() * 4).to(dtype)).to(device).requires_grad_(True)
            y2 = m(x2.clone() if site else x2)
            y2.backward((torch.randn(y2.shape, generator=g) * 3).to(y2.dtype).to(device))
            outs.append(("y2", y2.detach().cpu()))
            outs.append(("gx2", x2.grad.cpu()))
        gout = torch.randn(y.shape, generator=g).to(y.dtype)
        if channels_last and gout.dim() == 4:
            gout = gout.contiguous(memory_format=torch.channels_last)
        outs.append(("layout", torch.tensor([st for st, n in zip(y.stride(), y.shape) if n > 1])))     # (the stride of an extent-1 dim means nothing)
        for p in m.parameters():
            p.grad = None
        y.backward(gout.to(device))
        outs.append(("y", y.detach().cpu()))
        outs.append(("gx", xd.grad.cpu()))
        for name, p in m.named_parameters():
            if p.grad is not None:
                outs.append(("grad:" + name, p.grad.cpu()))
    if captured is not None:
        qs.graphs.resync_host_state(m)
    for k, v in m.state_dict().items():
        outs.append(("state:" + k, v.detach().cpu()))
    return outs


def one_case(rng, idx, dry=False):
    desc, factory, shape, dtype = build(rng)
    steps = rng.choice([3, 5, 6]) if desc["what"] not in ("site", "tok") else rng.choice([6, 8, 10])
    eval_from = rng.choice([steps, steps - 1])
    channels_last = rng.random() < 0.4
    # (statistics of channels_last inputs follow ATen's own order for that layout whichever dim leads the reduced ones -- N: the
    # multi-row / row-sum split; C or H with the batch dim kept; W: the scalar inner sum, qs_mean_cl_w since ABI v20; W with H == 1,
    # where "channels_last" strides say nothing about the order: the general route, qs_mean_strided, since ABI v22)
    if desc["what"] in ("act_p", "act_pq"):
        channels_last = channels_last and len(shape) == 4
    permute = None
    if desc["what"] in ("site", "tok", "act_q", "act_p", "act_pq") and not channels_last and len(shape) >= 2 and rng.random() < 0.25:
        # any other dense layout: statistics in ATen's order for it (qs_mean_strided, ABI v22), results back in the input's layout
        permute = list(range(len(shape)))
        while permute == sorted(permute):
            rng.shuffle(permute)
    batcher = rng.random() < 0.6
    twin = rng.random() < 0.25
    nonfinite = None
    if desc["what"] in ("conv", "linear", "net") and rng.random() < 0.25:
        nonfinite = (rng.choice([float("nan"), float("inf"), float("-inf")]), rng.choice([steps - 2, steps - 1]))
    if desc["what"] in ("site", "tok", "act_q", "act_p", "act_pq") and rng.random() < 0.3:
        # (ATen's CPU hardtanh_backward gates a NaN input differently in its vector body and its scalar tail -- qs_common.h,
        # act_open -- so the clamping activations get infinities only)
        values = [float("inf"), float("-inf")] + ([] if desc.get("act") in ("relu6", "hardtanh", "hardtanh_odd") else [float("nan")] * 2)
        nonfinite = (rng.choice(values), rng.choice([steps - 3, steps - 2, steps - 1]))
    # route switches of the HIP path (the CPU path has none of these routes: the results must not depend on them)
    routes = dict(fold_relu=rng.random() < 0.8, relu_gate=rng.random() < 0.8, elide_pruned=rng.choice(["forward", "forward", "off"]),
                  graph_safe=rng.random() < 0.2)
    if desc["what"] == "net":
        steps = rng.choice([6, 8, 10, 12])
        eval_from = rng.choice([steps, steps - 1])
        routes["batch_weights"] = batcher = rng.random() < 0.85      # (off: the same network layer by layer)
        if nonfinite is not None:
            nonfinite = (nonfinite[0], rng.choice([steps - 3, steps - 2]))
    # (float64: the ATen-on-device route keeps its running means on the host, as the CPU path does -- not capturable, refused)
    graph = GRAPH and desc["what"] in ("site", "tok", "act_q", "act_p", "act_pq") and rng.random() < 0.7 and desc["dtype"] != "float64"
    if graph:
        routes["graph_safe"] = True
    if EXCHANGE and rng.random() < 0.6:     # the statistics exchange of a data-parallel run, live on a one-rank group (same values)
        routes["sync_statistics"] = "always"
    desc.update(i=idx, steps=steps, eval_from=eval_from, channels_last=channels_last, permute=permute, batcher=batcher, twin=twin,
                nonfinite=nonfinite, routes=routes)
    if dry:
        return None
    if VERBOSE:
        print(desc, flush=True)
    results = {}
    for device in ("cpu", "cuda"):
        try:
            if device == "cuda":
                qs.set_qsparse_options(**routes)
            results[device] = run(factory, shape, dtype, device, 4000 + idx, steps, eval_from, desc['what'] in ('conv', 'linear'),
                                  channels_last, batcher, twin, desc['what'] in ('site', 'tok'), nonfinite, graph, permute)
        except Exception as e:      # noqa: BLE001 -- both paths must fail alike
            results[device] = ("raised", type(e).__name__)
        finally:
            qs.set_qsparse_options(fold_relu=True, relu_gate=True, elide_pruned="forward", graph_safe=False, batch_weights=True)
            qs.set_qsparse_options(sync_statistics=False) if EXCHANGE else None
    a, b = results["cpu"], results["cuda"]
    if isinstance(a, tuple) or isinstance(b, tuple):
        if isinstance(a, tuple) and isinstance(b, tuple) and a[1] == b[1]:
            return "ok"
        return dict(desc, cpu=a if isinstance(a, tuple) else "ran", gpu=b if isinstance(b, tuple) else "ran")
    if len(a) != len(b):
        return dict(desc, mismatch="number of outputs")
    for (ka, va), (kb, vb) in zip(a, b):
        if (ka == kb and ka.startswith(("gx", "grad:")) and va.shape == vb.shape and va.dtype == vb.dtype
                and bool(((va == vb) | (va.isnan() & vb.isnan())).all())):
            continue   # gradients clamped to [-0, +0] by a zero scale: ATen's own vector body and scalar tail disagree on the sign
            #            (equal as numbers, NaNs -- a non-finite case -- in the same places)
        if ka != kb or not same(va, vb):
            if VERBOSE and va.shape == vb.shape and va.is_floating_point():
                bad = ((va.float() != vb.float()) | (torch.signbit(va) != torch.signbit(vb))).view(-1).nonzero().view(-1)
                print(f"   {ka}: {bad.numel()} mismatching elements; first (index, cpu, gpu):",
                      [(i, float(va.reshape(-1)[i]), float(vb.reshape(-1)[i])) for i in bad[:6].tolist()], flush=True)
            return dict(desc, mismatch=(ka, kb), cpu=(tuple(va.shape), str(va.dtype)), gpu=(tuple(vb.shape), str(vb.dtype)),
                        max_abs=float((va.float() - vb.float()).abs().max()) if va.shape == vb.shape and va.numel() else None)
    return "ok"


def main():
    torch.set_num_threads(1)     # see tests/fuzz/fuzz_parity.py
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    if EXCHANGE:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29711", rank=0, world_size=1)
        for name in ("all_gather_into_tensor", "all_reduce"):       # how many collectives the cases really issued
            def counted(*a, _f=getattr(dist, name), **k):
                COLLECTIVES[0] += 1
                return _f(*a, **k)
            setattr(dist, name, counted)
        qs.set_qsparse_options(sync_statistics=False)     # (the CPU runs: no collective on CPU tensors through RCCL)
    only = os.environ.get("QS_FUZZ_ONLY")
    ran = fails = 0
    mode = os.environ.get("QS_FUZZ_MODE", "modules")
    for i in range(cases):
        r = (one_functional if mode == "functional" else one_case)(rng, i, dry=only is not None and i != int(only))
        if r is None:
            continue
        ran += 1
        if r != "ok":
            fails += 1
            print("FAIL", r, flush=True)
    print(f"fuzz cpu-vs-gpu: {ran} cases, {fails} failures (seed {seed}); weight batcher engaged in {ENGAGED[0]}"
          + (f" (steady-state fast path armed in {STEADY[0]})" if STEADY[0] else "")
          + (f"; {COLLECTIVES[0]} collectives on the one-rank group" if EXCHANGE else "")
          + (f"; {GRAPHED[0]} cases captured into a hipGraph" if GRAPH else ""))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
