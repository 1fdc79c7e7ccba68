#!/usr/bin/env python3
"""Golden-vector generator for the quantize/prune hot path.

Runs ONLY in the build container: it imports the real mlzxy/qsparse (v2.0.1) from the read-only
checkout at /root/reference and records inputs -> outputs of the reference's own functions and
layers as small .npz fixtures next to this file.  Nothing from the reference travels: the fixtures
hold data only (seeded inputs, the reference's outputs, and per-step state trajectories).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate.py

bf16 tensors are stored as their uint16 bit patterns under the key suffix ``__bf16``; every other
array is stored in its own dtype.  ``meta`` (a JSON string) lists the cases of each file.

Reference entry points exercised (file:line in /root/reference):
  F1 ScalerQuantization.forward/backward       qsparse/quantize.py:87-131
  F2 DecimalQuantization.forward/backward      qsparse/quantize.py:31-77
  F3 LineQuantization.forward                  qsparse/quantize.py:141-185
  F4 QuantizeLayer trajectories                qsparse/quantize.py:434-518 (+ quantizers :275-430)
  F5 squeeze_tensor_to_shape                   qsparse/util.py:79-99
  F6 calculate_mask_given_importance           qsparse/util.py:103-117
  F7 PruneLayer trajectories                   qsparse/sparse.py:157-273 (+ callbacks :18-152)
  F8 convert() module trees                    qsparse/convert.py:21-245
  F9 state_dict schema + preload               qsparse/util.py:120-145
  F10 prune->quantize pair trajectory (the headline pair, small shape)
  F13 UniformPruningCallback trajectories      qsparse/sparse.py:125-152 (numpy global RNG, seeded per case)
  F14 counters written through ``.data``       qsparse/quantize.py:495, qsparse/sparse.py:251-269,104-118
  F15 MagnitudePruningCallback(use_gradient=True)  qsparse/sparse.py:69-80 (tensor hook -> update_magnitude(grad), :82-89)
  F16 the MNIST --pq recipe with devise_layerwise_pruning_schedule  examples/mnist.py:17-44,193-199; qsparse/sparse.py:343-359
  F18 error behaviour: misuse scenarios (tests/golden/error_scenarios.py) -> exception type / message or "ok"
  F17 the prune->quantize pair with NaN / Inf / -Inf on PRUNED channels  qsparse/sparse.py:263 (x * mask: NaN there),
      qsparse/quantize.py:109 (.int() of NaN: INT_MIN), :329-347 (x.abs().max() carries the NaN into a live scale),
      :126-130 (clamp with NaN bounds), :316 (nan_to_num in the decimal)
"""
import io
import json
import os
import sys
import contextlib

sys.dont_write_bytecode = True
REF = os.environ.get("QSPARSE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse  # the real reference
from qsparse import convert, prune, quantize
from qsparse.quantize import (AdaptiveQuantizer, DecimalQuantizer, ScalerQuantizer,
                              quantize_with_decimal, quantize_with_line, quantize_with_scaler)
from qsparse.sparse import MagnitudePruningCallback, UniformPruningCallback, devise_layerwise_pruning_schedule
from qsparse.util import (calculate_mask_given_importance, preload_qsparse_state_dict,
                          squeeze_tensor_to_shape)

assert qsparse.__version__ == "2.0.1"
HERE = os.path.dirname(os.path.abspath(__file__))
qsparse.set_qsparse_options(log_on_created=False, log_during_train=False)


def put(store, key, t):
    """store a tensor / array / scalar under `key` (bf16 as uint16 bits)."""
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().clone()  # clone: layer state is updated in place later
        if t.dtype == torch.bfloat16:
            store[key + "__bf16"] = t.contiguous().view(torch.int16).numpy().view(np.uint16)
            return
        if t.dtype == torch.float16:
            store[key + "__f16"] = t.contiguous().numpy()
            return
        store[key] = t.contiguous().numpy()
    else:
        store[key] = np.asarray(t)


def save(name, store, meta):
    store = dict(store)
    # the torch whose ATen produced these bits: the staged means follow THAT version's summation order (SumKernel.cpp), which the
    # HIP kernels reproduce -- qsparse_amd warns at import when it runs under another one (`PINNED_TORCH`)
    meta = dict(meta, torch=torch.__version__, intra_op_threads=torch.get_num_threads())
    store["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name}.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(meta['cases'])} cases")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


# --------------------------------------------------------------------------------------------
# F1 / F2: scaler + decimal STE quantizers, forward and backward
# --------------------------------------------------------------------------------------------
def f1_f2():
    for fname, fn in (("f1_scaler", quantize_with_scaler), ("f2_decimal", quantize_with_decimal)):
        store, cases = {}, []
        shape = (3, 6, 4, 4)
        C = shape[1]
        idx = 0
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            for bits in (4, 8):
                for pkind in ("pyfloat", "zerodim", "t11", "chan1", "chan0"):
                    for flip, passthrough in ((False, False), (True, False), (False, True)):
                        if (flip or passthrough) and pkind not in ("t11", "chan1"):
                            continue
                        if dt == torch.float16 and (bits == 4 or pkind in ("zerodim", "chan0")):
                            continue
                        x = (torch.randn(shape, generator=gen(100 + idx)) * 2).to(dt)
                        flat = x.view(-1)
                        if fname == "f1_scaler":
                            s0 = 0.1 if bits == 8 else 0.37
                            # exact half-way quotients, and values far outside the code range
                            hw = torch.tensor([0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 3.5, 1e3, -1e3, 300.25]) * s0
                            flat[: len(hw)] = hw.to(dt)
                            if pkind == "pyfloat":
                                param = s0
                            elif pkind == "zerodim":
                                param = torch.tensor(s0)
                            elif pkind == "t11":
                                param = torch.full((1, 1), s0)
                            else:
                                n = C if pkind == "chan1" else shape[0]
                                param = (torch.rand(n, 1, generator=gen(7 + idx)) * 0.3 + 0.01)
                        else:
                            d0 = 5 if bits == 8 else 2
                            hw = torch.tensor([0.74, -0.74, 0.76, -0.76, 1.0, -1.0, 1.99, 700.3, -700.3]) / 2.0 ** d0
                            flat[: len(hw)] = hw.to(dt)
                            if pkind == "pyfloat":
                                param = d0
                            elif pkind == "zerodim":
                                param = torch.tensor(float(d0))
                            elif pkind == "t11":
                                param = torch.full((1, 1), float(d0))
                            else:
                                n = C if pkind == "chan1" else shape[0]
                                param = torch.randint(0, 8, (n, 1), generator=gen(7 + idx)).float()
                        ci = {"chan1": 1, "chan0": 0}.get(pkind, -1)
                        xin = x.clone().requires_grad_(True)
                        y = fn(xin, bits, param, ci, False, passthrough, flip)
                        gout = (torch.randn(shape, generator=gen(900 + idx)) * (3.0 if bits == 8 else 1.5))
                        gout = gout.to(y.dtype)
                        y.backward(gout.clone())
                        # integer codes, computed with the reference's own expression
                        with torch.no_grad():
                            if fname == "f1_scaler":
                                pv = param.view([-1 if i == ci else 1 for i in range(4)]) if (
                                    isinstance(param, torch.Tensor) and param.numel() > 1) else param
                                codes = (x / pv).round().int()
                            else:
                                toi = 2.0 ** param
                                if isinstance(param, torch.Tensor) and param.numel() > 1:
                                    toi = toi.view([-1 if i == ci else 1 for i in range(4)])
                                codes = (x * toi).int()
                        k = f"c{idx}_"
                        put(store, k + "x", x)
                        put(store, k + "param", param if isinstance(param, torch.Tensor) else np.float64(param))
                        put(store, k + "y", y)
                        put(store, k + "codes", codes)
                        put(store, k + "gout", gout)
                        put(store, k + "gx", xin.grad)
                        cases.append(dict(id=idx, dtype=str(dt).replace("torch.", ""), bits=bits, pkind=pkind,
                                          channel_index=ci, flip_axis=flip, backward_passthrough=passthrough,
                                          y_dtype=str(y.dtype).replace("torch.", ""),
                                          gx_dtype=str(xin.grad.dtype).replace("torch.", "")))
                        idx += 1
        save(fname, store, dict(cases=cases, shape=list(shape)))


# --------------------------------------------------------------------------------------------
# F3: line (asymmetric) quantizer
# --------------------------------------------------------------------------------------------
def f3():
    store, cases = {}, []
    shape = (3, 6, 5, 5)
    idx = 0
    for dt in (torch.float32, torch.bfloat16):
        for bits in (4, 8):
            for lkind in ("tuple", "t12", "chan1", "chan0"):
                for fzp in (True, False):
                    x = (torch.randn(shape, generator=gen(300 + idx)) * 1.5).to(dt)
                    if lkind == "tuple":
                        lines, ci = (-0.1, 0.9), -1
                    elif lkind == "t12":
                        lines, ci = torch.tensor([[-1.3, 2.1]]), -1
                    else:
                        ci = 1 if lkind == "chan1" else 0
                        n = shape[ci]
                        lo = -torch.rand(n, 1, generator=gen(40 + idx)) * 2
                        hi = torch.rand(n, 1, generator=gen(41 + idx)) * 2 + 0.05
                        lines = torch.cat([lo, hi], 1)
                        lines[0, 1] = lines[0, 0]  # a degenerate row: step == 0 -> 1e-4
                    y = quantize_with_line(x.clone(), bits, lines if not isinstance(lines, torch.Tensor) else lines.clone(),
                                           ci, False, fzp)
                    k = f"c{idx}_"
                    put(store, k + "x", x)
                    put(store, k + "lines", torch.tensor(lines).view(-1, 2) if not isinstance(lines, torch.Tensor) else lines)
                    put(store, k + "y", y)
                    cases.append(dict(id=idx, dtype=str(dt).replace("torch.", ""), bits=bits, lkind=lkind,
                                      channel_index=ci, float_zero_point=fzp,
                                      y_dtype=str(y.dtype).replace("torch.", "")))
                    idx += 1
    save("f3_line", store, dict(cases=cases, shape=list(shape)))


# --------------------------------------------------------------------------------------------
# F4: QuantizeLayer trajectories
# --------------------------------------------------------------------------------------------
def f4():
    store, cases = {}, []
    idx = 0
    specs = []
    for cbname in ("scaler", "decimal", "adaptive"):
        specs.append(dict(cb=cbname, channelwise=-1, shape=(3, 5, 6, 6), timeout=2, steps=6, bits=8, dtype="float32"))
        specs.append(dict(cb=cbname, channelwise=1, shape=(1, 5, 6, 6), timeout=1, steps=5, bits=4, dtype="float32"))
        specs.append(dict(cb=cbname, channelwise=-1, shape=(3, 5, 6, 6), timeout=1, steps=5, bits=4, dtype="bfloat16"))
    specs.append(dict(cb="adaptive", channelwise=1, shape=(4, 5, 6, 6), timeout=1, steps=5, bits=8, dtype="float32"))
    specs.append(dict(cb="adaptive", channelwise=1, shape=(4, 5, 6, 6), timeout=1, steps=5, bits=8, dtype="bfloat16"))
    specs.append(dict(cb="scaler", channelwise=-1, shape=(3, 5, 6, 6), timeout=0, steps=3, bits=8, dtype="float32"))
    specs.append(dict(cb="scaler", channelwise=-1, shape=(2, 12), timeout=1, steps=4, bits=8, dtype="float32"))
    mk = dict(scaler=ScalerQuantizer, decimal=DecimalQuantizer, adaptive=AdaptiveQuantizer)
    for sp in specs:
        dt = getattr(torch, sp["dtype"])
        with quiet():
            layer = quantize(bits=sp["bits"], channelwise=sp["channelwise"], timeout=sp["timeout"], callback=mk[sp["cb"]]())
        layer.train()
        k = f"c{idx}_"
        for s in range(sp["steps"] + 2):
            if s == sp["steps"]:
                layer.eval()  # last two steps in eval mode
            x = ((torch.rand(sp["shape"], generator=gen(1000 + 17 * idx + s)) - 0.5) * (4 + s)).to(dt)
            with quiet():
                y = layer(x)
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_y", y)
            put(store, k + f"s{s}_weight", layer.weight)
            put(store, k + f"s{s}_n_updates", layer._n_updates)
        cases.append(dict(id=idx, **{**sp, "shape": list(sp["shape"])}, total_steps=sp["steps"] + 2))
        idx += 1

    # weight-side: quantize(Conv2d) with bias, channelwise=0; shared callback between weight and bias
    for cbname in ("scaler", "decimal"):
        torch.manual_seed(5)
        conv = nn.Conv2d(4, 6, 3)
        w0, b0 = conv.weight.detach().clone(), conv.bias.detach().clone()
        with quiet():
            qconv = quantize(conv, bits=8, bias_bits=8, timeout=2, channelwise=0, callback=mk[cbname]())
        qconv.train()
        k = f"c{idx}_"
        put(store, k + "w0", w0)
        put(store, k + "b0", b0)
        steps = 5
        for s in range(steps):
            x = torch.rand((2, 4, 8, 8), generator=gen(2000 + s))
            with quiet():
                y = qconv(x)
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_y", y)
            with quiet():
                put(store, k + f"s{s}_qweight", qconv.weight)  # property: advances the schedule once more
                put(store, k + f"s{s}_qbias", qconv.bias)
            put(store, k + f"s{s}_wscale", qconv.quantize.weight)
            put(store, k + f"s{s}_bscale", qconv.quantize_bias.weight)
        cases.append(dict(id=idx, cb=cbname, kind="conv_weight", steps=steps, bits=8, timeout=2, channelwise=0))
        idx += 1
    save("f4_quantize_layer", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F5: squeeze_tensor_to_shape (staged means)
# --------------------------------------------------------------------------------------------
def f5():
    store, cases = {}, []
    idx = 0
    for dt in (torch.float32, torch.bfloat16):
        for shape, mshape in (((40, 6, 9, 8), (1, 6, 1, 1)),
                              ((6, 6, 20, 8), (6, 6, 1, 1)),
                              ((6, 6, 8, 8), (1, 1, 1, 1)),
                              ((6, 6, 8, 40), (1, 6, 8, 1)),
                              ((7, 5, 3, 9), (1, 5, 1, 1)),
                              ((20, 33), (1, 33)),
                              ((6, 10, 3, 3), (1, 10, 3, 3))):
            x = (torch.randn(shape, generator=gen(500 + idx)) * torch.linspace(0.25, 4, shape[1]).view(
                [1, -1] + [1] * (len(shape) - 2))).to(dt)
            out = squeeze_tensor_to_shape(x.abs(), mshape)
            k = f"c{idx}_"
            put(store, k + "x", x)
            put(store, k + "out", out)
            cases.append(dict(id=idx, dtype=str(dt).replace("torch.", ""), shape=list(shape), mask_shape=list(mshape)))
            idx += 1
    save("f5_squeeze", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F6: calculate_mask_given_importance
# --------------------------------------------------------------------------------------------
def f6():
    store, cases = {}, []
    idx = 0
    sched = [float(torch.tensor(0.75 * (1 - (1 - (i + 1) / 4) ** 3), dtype=torch.float32).item()) for i in range(4)]
    for n_shape in ((1, 256, 1, 1), (5, 30, 7, 8), (32, 16, 3, 3), (1, 7, 1, 1)):
        for kind in ("random", "ties", "negzero"):
            imp = torch.rand(n_shape, generator=gen(600 + idx))
            if kind == "ties":
                imp = (imp * 6).floor() / 6
            if kind == "negzero":
                imp = imp - 0.5
                imp.view(-1)[::5] = 0.0
            for s in [0.0, 0.47, 0.5, 0.75, 0.999] + sched:
                n = imp.numel()
                if max(int(s * n - 1), 0) + 1 >= n:
                    continue
                mask = calculate_mask_given_importance(imp, s)
                k = f"c{idx}_"
                ikey = f"imp{len(n_shape)}_{n_shape[0]}_{n_shape[1]}_{kind}"
                put(store, ikey, imp)
                store[k + "mask"] = np.packbits(mask.numpy().reshape(-1))
                cases.append(dict(id=idx, kind=kind, shape=list(n_shape), sparsity=s, imp_key=ikey))
                idx += 1
    save("f6_mask", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F7: PruneLayer trajectories
# --------------------------------------------------------------------------------------------
def f7():
    store, cases = {}, []
    idx = 0
    specs = [
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=5, interval=2, repetition=3, rampup=False,
             cb=dict(), dtype="float32", steps=16),
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.75, start=3, interval=2, repetition=3, rampup=True,
             cb=dict(), dtype="bfloat16", steps=14),
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=2, interval=3, repetition=2, rampup=False,
             cb=dict(running_average=False), dtype="float32", steps=12),
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.6, start=2, interval=2, repetition=4, rampup=False,
             cb=dict(mask_refresh_interval=3, stop_mask_refresh=9), dtype="bfloat16", steps=16),
        dict(shape=(2, 6, 5, 5), dims=[0, 1, 2, 3], sparsity=0.5, start=2, interval=2, repetition=2, rampup=False,
             cb=dict(), dtype="float32", steps=10),
        dict(shape=(2, 6, 5, 5), dims=[1, 2, 3], sparsity=0.7, start=1, interval=2, repetition=2, rampup=False,
             cb=dict(), dtype="bfloat16", steps=9),
        dict(shape=(8, 24), dims=[1], sparsity=0.5, start=1, interval=1, repetition=3, rampup=False,
             cb=dict(), dtype="float32", steps=8),
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=2, interval=2, repetition=2, rampup=False,
             cb=dict(l0=True), dtype="float32", steps=9, relu=True),
    ]
    for sp in specs:
        dt = getattr(torch, sp["dtype"])
        with quiet():
            layer = prune(sparsity=sp["sparsity"], dimensions=set(sp["dims"]), start=sp["start"], interval=sp["interval"],
                          repetition=sp["repetition"], rampup=sp["rampup"], callback=MagnitudePruningCallback(**sp["cb"]))
        layer.train()
        k = f"c{idx}_"
        scale = torch.linspace(0.25, 4, sp["shape"][1]).view([1, -1] + [1] * (len(sp["shape"]) - 2))
        for s in range(sp["steps"] + 2):
            if s == sp["steps"]:
                layer.eval()
            x = torch.randn(sp["shape"], generator=gen(3000 + 31 * idx + s)) * scale
            if sp.get("relu"):
                x = x.relu()
            x = x.to(dt).requires_grad_(True)
            gout = torch.randn(sp["shape"], generator=gen(4000 + 31 * idx + s)).to(dt)
            with quiet():
                y = layer(x)
            y.backward(gout)
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_gout", gout)
            put(store, k + f"s{s}_y", y)
            put(store, k + f"s{s}_gx", x.grad)
            put(store, k + f"s{s}_mask", layer.mask)
            put(store, k + f"s{s}_n_updates", layer._n_updates)
            put(store, k + f"s{s}_cur_sparsity", layer._cur_sparsity)
            put(store, k + f"s{s}_t", layer.callback.t)
            if hasattr(layer.callback, "magnitude"):
                put(store, k + f"s{s}_magnitude", layer.callback.magnitude)
        cases.append(dict(id=idx, **{**sp, "shape": list(sp["shape"])}, total_steps=sp["steps"] + 2))
        idx += 1

    # weight-side: prune(Conv2d), unstructured over dims {0,1,2,3}? reference default dims={1}: per-input-channel
    for cbkw, dims in ((dict(running_average=False), [1]), (dict(), [0, 1, 2, 3])):
        torch.manual_seed(11)
        conv = nn.Conv2d(10, 12, 3)
        w0 = conv.weight.detach().clone()
        with quiet():
            pconv = prune(conv, sparsity=0.5, dimensions=set(dims), start=2, interval=2, repetition=2,
                          callback=MagnitudePruningCallback(**cbkw))
        pconv.train()
        k = f"c{idx}_"
        put(store, k + "w0", w0)
        steps = 9
        x = torch.rand((1, 10, 8, 8), generator=gen(77))
        put(store, k + "x", x)
        for s in range(steps):
            with quiet():
                y = pconv(x)
            put(store, k + f"s{s}_y", y)
            put(store, k + f"s{s}_mask", pconv.prune.mask)
        cases.append(dict(id=idx, kind="conv_weight", dims=dims, cb=cbkw, steps=steps, sparsity=0.5, start=2, interval=2,
                          repetition=2))
        idx += 1
    save("f7_prune_layer", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F8/F9: convert() trees, state_dict schema, preload round trip
# --------------------------------------------------------------------------------------------
class LeNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 6, kernel_size=5)
        self.conv2 = nn.Conv2d(6, 16, kernel_size=5)
        self.fc1 = nn.Linear(16 * 5 * 5, 120)
        self.fc2 = nn.Linear(120, 84)
        self.fc3 = nn.Linear(84, 10)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.conv1(x)), 2)
        x = F.max_pool2d(F.relu(self.conv2(x)), 2)
        x = x.view(x.size(0), -1)
        return self.fc3(F.relu(self.fc2(F.relu(self.fc1(x)))))


class MnistNet(nn.Module):
    """the CNN of the reference's MNIST example (architecture only), examples/mnist.py:17-44"""

    def __init__(self):
        super().__init__()
        self.conv_part = nn.Sequential(nn.Conv2d(1, 32, 3, 1), nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 64, 3, 1),
                                       nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2), nn.Dropout(0.25))
        self.linear_part = nn.Sequential(nn.Flatten(), nn.Linear(9216, 128), nn.BatchNorm1d(128), nn.ReLU(),
                                         nn.Dropout(0.5), nn.Linear(128, 10))

    def forward(self, x):
        return F.log_softmax(self.linear_part(self.conv_part(x)), dim=1)


def f8_f9():
    trees = {}
    with quiet():
        excl = [(nn.Conv2d, [0]), (nn.Linear, [-1])]
        m = convert(LeNet(), prune(sparsity=0.5, callback=MagnitudePruningCallback()),
                    weight_layers=[nn.Conv2d, nn.Linear], activation_layers=[nn.Conv2d, nn.Linear],
                    excluded_weight_layer_indexes=excl, excluded_activation_layer_indexes=excl)
        m = convert(m, quantize(bits=8), weight_layers=[nn.Conv2d, nn.Linear],
                    activation_layers=[nn.Conv2d, nn.Linear], input=True)
    trees["lenet_pq"] = {k: v.__class__.__name__ for k, v in m.named_modules()}
    trees["lenet_pq_str"] = str(m)
    with quiet():
        m = convert(MnistNet(), prune(sparsity=0.75, dimensions={1}), activation_layers=[nn.ReLU],
                    excluded_activation_layer_indexes=[(nn.ReLU, [-1])])
        m = convert(m, quantize(bits=4, channelwise=-1, timeout=50), activation_layers=[nn.ReLU],
                    weight_layers=[nn.Conv2d, nn.Linear], input=True)
    trees["mnist_pq"] = {k: v.__class__.__name__ for k, v in m.named_modules()}
    trees["mnist_pq_str"] = str(m)
    with quiet():
        from collections import OrderedDict
        net = nn.Sequential(OrderedDict([("conv1", nn.Conv2d(3, 6, kernel_size=5)), ("fc1", nn.Linear(84, 10))]))
        c = convert(net, quantize(bits=8), activation_layers=[nn.Conv2d, nn.Linear], order="pre")
        trees["order_pre_str"] = str(c)
        c = convert(c, prune(sparsity=0.5), activation_layers=[nn.Conv2d, nn.Linear])
        trees["order_pre_nested_str"] = str(c)

    # F9: state dict schema + values after a short run of quantize(prune(conv))
    def make_conv():
        torch.manual_seed(3)
        with quiet():
            return quantize(prune(nn.Conv2d(16, 32, 3), sparsity=0.5, start=20, interval=5, repetition=4), bits=8, timeout=10)

    conv = make_conv()
    conv.train()
    for s in range(45):
        with quiet():
            conv(torch.rand(4, 16, 7, 7, generator=gen(5000 + s)))
    sd = conv.state_dict()
    store = {}
    schema = {}
    for kname, v in sd.items():
        put(store, "sd_" + kname, v)
        schema[kname] = dict(dtype=str(v.dtype).replace("torch.", ""), shape=list(v.shape))
    xt = torch.rand(4, 16, 7, 7, generator=gen(9999))
    conv.eval()
    conv3 = make_conv()
    preload_qsparse_state_dict(conv3, sd)
    conv3.load_state_dict(sd)
    conv3.eval()
    with quiet():
        put(store, "eval_x", xt)
        put(store, "eval_y_trained", conv(xt))
        put(store, "eval_y_reloaded", conv3(xt))  # NOTE: _quantized is not restored by the reference (quirk B7)
    save("f8_f9_convert_state", store, dict(cases=[dict(trees=trees, state_schema=schema)]))


# --------------------------------------------------------------------------------------------
# F10: the headline pair at a small shape: ReLU -> prune(0.75,{1}) -> quantize(4-bit, tensor-wise)
# --------------------------------------------------------------------------------------------
def f10():
    store, cases = {}, []
    idx = 0
    for dt, shape, bits, sp in ((torch.bfloat16, (4, 16, 4, 4), 4, 0.75), (torch.float32, (4, 16, 4, 4), 8, 0.5),
                                (torch.bfloat16, (8, 32, 6, 6), 4, 0.75)):
        with quiet():
            pl = prune(sparsity=sp, dimensions={1}, start=1, interval=2, repetition=3)
            ql = quantize(bits=bits, channelwise=-1, timeout=2)
        pl.train(), ql.train()
        steps = 10
        k = f"c{idx}_"
        C = shape[1]
        for s in range(steps + 1):
            if s == steps:
                pl.eval(), ql.eval()
            x = (torch.randn(shape, generator=gen(7000 + 13 * idx + s)).relu() * torch.linspace(0.25, 4, C).view(1, -1, 1, 1)).to(dt)
            x.requires_grad_(True)
            with quiet():
                h = pl(x)
                y = ql(h)
            gout = torch.randn(shape, generator=gen(8000 + 13 * idx + s)).to(y.dtype)
            y.backward(gout.clone())
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_gout", gout)
            put(store, k + f"s{s}_y", y)
            put(store, k + f"s{s}_gx", x.grad)
            put(store, k + f"s{s}_mask", pl.mask)
            if hasattr(pl.callback, "magnitude"):
                put(store, k + f"s{s}_magnitude", pl.callback.magnitude)
            put(store, k + f"s{s}_cur_sparsity", pl._cur_sparsity)
            put(store, k + f"s{s}_scale", ql.weight)
        cases.append(dict(id=idx, dtype=str(dt).replace("torch.", ""), shape=list(shape), bits=bits, sparsity=sp,
                          start=1, interval=2, repetition=3, timeout=2, total_steps=steps + 1))
        idx += 1
    save("f10_prune_quant_pair", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F11: fuse_bn (qsparse/fuse.py:76-163)
# --------------------------------------------------------------------------------------------
def make_bn_nets():
    """deterministic conv/linear/deconv + BatchNorm stacks with non-trivial running statistics"""
    from collections import OrderedDict
    torch.manual_seed(21)
    nets = OrderedDict()
    nets["conv"] = (nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5)), (4, 3, 8, 8))
    nets["linear"] = (nn.Sequential(nn.Linear(12, 7, bias=False), nn.BatchNorm1d(7)), (6, 12))
    nets["deconv"] = (nn.Sequential(nn.ConvTranspose2d(3, 5, 3), nn.BatchNorm2d(5)), (4, 3, 6, 6))
    nets["nested"] = (nn.Sequential(nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)), nn.ReLU(),
                                    nn.Sequential(nn.Conv2d(4, 4, 3)), nn.BatchNorm2d(4), nn.ReLU(),
                                    nn.Sequential(nn.BatchNorm2d(4), nn.ConvTranspose2d(4, 2, 3), nn.BatchNorm2d(2))),
                       (3, 3, 12, 12))
    for net, shape in nets.values():
        for m in net.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.uniform_(m.weight, 0.5, 1.5)
                nn.init.uniform_(m.bias, -0.5, 0.5)
        net.train()
        for i in range(3):
            net(torch.randn(shape, generator=gen(600 + i)))
        net.eval()
    return nets


def f11():
    from qsparse import fuse_bn
    store, cases = {}, []
    for name, (net, shape) in make_bn_nets().items():
        x = torch.randn(shape, generator=gen(700))
        for kname, v in net.state_dict().items():
            put(store, f"{name}_in_{kname}", v)
        with quiet():
            fused = fuse_bn(net, log=False)
        for kname, v in fused.state_dict().items():
            put(store, f"{name}_out_{kname}", v)
        put(store, name + "_x", x)
        put(store, name + "_y", fused(x))
        cases.append(dict(name=name, shape=list(shape), tree=str(fused), in_keys=None, out_keys=list(fused.state_dict().keys())))
    save("f11_fuse_bn", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F12: public API surface (names, parameter names/defaults/order) -- data about signatures only
# --------------------------------------------------------------------------------------------
def f12():
    import inspect
    # `qsparse.quantize` / `qsparse.convert` the ATTRIBUTES are functions (the package re-exports them), so the
    # modules have to be fetched from sys.modules
    Q, S, U, FU, IM = (sys.modules["qsparse." + m] for m in ("quantize", "sparse", "util", "fuse", "imitation"))

    def sig(obj):
        target = obj.__init__ if inspect.isclass(obj) else obj
        out = []
        for name, prm in inspect.signature(target).parameters.items():
            if name == "self":
                continue
            d = prm.default
            if d is inspect.Parameter.empty:
                dflt = "<required>"
            elif isinstance(d, (int, float, str, bool, type(None))):
                dflt = repr(d)
            elif isinstance(d, (list, tuple, set)):
                dflt = repr(sorted(d) if isinstance(d, set) else list(d))
            else:
                dflt = "<" + type(d).__name__ + ">"
            out.append([name, str(prm.kind), dflt])
        return out

    api = {}
    for modname, mod, names in (
            ("", qsparse, ["convert", "fuse_bn", "quantize", "DecimalQuantizer", "ScalerQuantizer", "AdaptiveQuantizer",
                           "MagnitudePruningCallback", "UniformPruningCallback", "prune", "devise_layerwise_pruning_schedule",
                           "auto_name_prune_quantize_layers", "calculate_mask_given_importance", "get_qsparse_option",
                           "set_qsparse_options"]),
            ("quantize.", Q, ["quantize_with_decimal", "quantize_with_scaler", "quantize_with_line", "QuantizeLayer",
                              "BaseQuantizer"]),
            ("sparse.", S, ["PruneLayer"]),
            ("util.", U, ["squeeze_tensor_to_shape", "preload_qsparse_state_dict", "nn_module"]),
            ("imitation.", IM, ["imitate"]),
            ("fuse.", FU, ["conv2d_bn_fuser", "linear_bn_fuser", "deconv2d_bn_fuser"])):
        for n in names:
            api[modname + n] = sig(getattr(mod, n))
    methods = {}
    for cls in (Q.DecimalQuantizer, Q.AdaptiveQuantizer, Q.BaseQuantizer, S.MagnitudePruningCallback):
        for m in ("optimize", "forward", "quantize", "get_weight_shape", "prune_and_update_mask", "receive_input",
                  "update_magnitude", "initialize"):
            if hasattr(cls, m):
                methods[cls.__name__ + "." + m] = sig(getattr(cls, m))
    save("f12_api_surface", {}, dict(cases=[dict(api=api, methods=methods, version=qsparse.__version__)]))


# --------------------------------------------------------------------------------------------
# F13: UniformPruningCallback.  The reference draws from numpy's GLOBAL RNG (sparse.py:148-150), so a case is pinned by
# seeding it right before the trajectory; a replay seeds it with the same value and must reproduce every mask.
# --------------------------------------------------------------------------------------------
def f13():
    store, cases = {}, []
    idx = 0
    specs = [
        dict(shape=(3, 10, 4, 4), dims=[1, 2, 3], sparsity=0.5, start=2, interval=2, repetition=3, rampup=False,
             cb=dict(), dtype="float32", steps=11, seed=1301),
        dict(shape=(3, 10, 4, 4), dims=[0, 1, 2, 3], sparsity=0.7, start=1, interval=3, repetition=2, rampup=True,
             cb=dict(mask_refresh_interval=2), dtype="bfloat16", steps=12, seed=1302),
        dict(shape=(4, 24), dims=[1], sparsity=0.75, start=1, interval=1, repetition=4, rampup=False,
             cb=dict(), dtype="float32", steps=8, seed=1303),
        dict(shape=(2, 6, 5, 5), dims=[1, 2, 3], sparsity=0.6, start=2, interval=2, repetition=2, rampup=False,
             cb=dict(mask_refresh_interval=3, stop_mask_refresh=6), dtype="float32", steps=10, seed=1304),
    ]
    for sp in specs:
        dt = getattr(torch, sp["dtype"])
        with quiet():
            layer = prune(sparsity=sp["sparsity"], dimensions=set(sp["dims"]), start=sp["start"], interval=sp["interval"],
                          repetition=sp["repetition"], rampup=sp["rampup"], callback=UniformPruningCallback(**sp["cb"]))
        layer.train()
        k = f"c{idx}_"
        np.random.seed(sp["seed"])
        for s in range(sp["steps"] + 2):
            if s == sp["steps"]:
                layer.eval()
            x = torch.randn(sp["shape"], generator=gen(13000 + 31 * idx + s)).to(dt).requires_grad_(True)
            gout = torch.randn(sp["shape"], generator=gen(14000 + 31 * idx + s)).to(dt)
            with quiet():
                y = layer(x)
            y.backward(gout)
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_gout", gout)
            put(store, k + f"s{s}_y", y)
            put(store, k + f"s{s}_gx", x.grad)
            put(store, k + f"s{s}_mask", layer.mask)
            put(store, k + f"s{s}_n_updates", layer._n_updates)
            put(store, k + f"s{s}_cur_sparsity", layer._cur_sparsity)
            put(store, k + f"s{s}_t", layer.callback.t)
        cases.append(dict(id=idx, **{**sp, "shape": list(sp["shape"])}, total_steps=sp["steps"] + 2))
        idx += 1

    # weight side: prune(Conv2d) with unstructured uniform masks
    torch.manual_seed(13)
    conv = nn.Conv2d(6, 8, 3)
    w0 = conv.weight.detach().clone()
    with quiet():
        pconv = prune(conv, sparsity=0.6, dimensions={0, 1, 2, 3}, start=1, interval=2, repetition=3,
                      callback=UniformPruningCallback())
    pconv.train()
    k = f"c{idx}_"
    put(store, k + "w0", w0)
    put(store, k + "b0", conv.bias)
    x = torch.rand((2, 6, 8, 8), generator=gen(1377))
    put(store, k + "x", x)
    np.random.seed(1305)
    steps = 9
    for s in range(steps):
        with quiet():
            y = pconv(x)
        put(store, k + f"s{s}_y", y)
        put(store, k + f"s{s}_mask", pconv.prune.mask)
    cases.append(dict(id=idx, kind="conv_weight", dims=[0, 1, 2, 3], steps=steps, sparsity=0.6, start=1, interval=2,
                      repetition=3, seed=1305))
    save("f13_uniform_prune", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F14: step counters written from outside through ``.data`` (the reference's own idiom, sparse.py:106) between
# forwards.  The reference re-reads every counter with .item() on each forward, so the write decides the very next
# step: fast-forwarding a quantizer past its timeout, rewinding it, moving a pruning schedule, restarting callback.t.
# --------------------------------------------------------------------------------------------
def f14():
    store, cases = {}, []
    idx = 0

    def record(k, s, layer, x, gout, y, extra=()):
        put(store, k + f"s{s}_x", x)
        put(store, k + f"s{s}_gout", gout)
        put(store, k + f"s{s}_y", y)
        put(store, k + f"s{s}_gx", x.grad)
        for name, t in extra:
            put(store, k + f"s{s}_{name}", t)

    # quantizers: (step, value) writes to _n_updates.data
    qspecs = [
        dict(kind="scaler", bits=8, channelwise=-1, timeout=3, shape=(4, 6, 5, 5), dtype="float32", steps=9,
             writes={2: 10, 5: 0, 7: 3}),
        dict(kind="scaler", bits=4, channelwise=-1, timeout=4, shape=(4, 6, 5, 5), dtype="bfloat16", steps=8,
             writes={1: 100, 4: 2}),
        dict(kind="decimal", bits=6, channelwise=-1, timeout=2, shape=(3, 8), dtype="float32", steps=7,
             writes={1: 2, 3: 1}),
    ]
    for sp in qspecs:
        dt = getattr(torch, sp["dtype"])
        cb = ScalerQuantizer() if sp["kind"] == "scaler" else DecimalQuantizer()
        with quiet():
            layer = quantize(bits=sp["bits"], channelwise=sp["channelwise"], timeout=sp["timeout"], callback=cb)
        layer.train()
        k = f"c{idx}_"
        for s in range(sp["steps"] + 1):
            if s == sp["steps"]:
                layer.eval()
            if s in sp["writes"]:
                if not layer.initted:     # the parameter exists only after the first forward
                    raise RuntimeError("write before first forward")
                layer._n_updates.data[:] = sp["writes"][s]
            x = (torch.randn(sp["shape"], generator=gen(15000 + 31 * idx + s)) * 3).to(dt).requires_grad_(True)
            gout = torch.randn(sp["shape"], generator=gen(16000 + 31 * idx + s))
            with quiet():
                y = layer(x)
            gout = gout.to(y.dtype)
            y.backward(gout)
            record(k, s, layer, x, gout, y, (("n_updates", layer._n_updates), ("weight", layer.weight)))
        cases.append(dict(id=idx, op="quantize", **{**sp, "shape": list(sp["shape"]),
                                                     "writes": {str(a): b for a, b in sp["writes"].items()}},
                          total_steps=sp["steps"] + 1))
        idx += 1

    # prune layers: writes to _n_updates.data and callback.t.data
    pspecs = [
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=4, interval=2, repetition=2, dtype="float32", steps=12,
             writes={2: ("n", 4), 6: ("t", 0), 8: ("n", 6), 10: ("n", 0)}),
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.75, start=2, interval=3, repetition=2, dtype="bfloat16", steps=11,
             writes={1: ("n", 5), 5: ("t", 7), 7: ("n", 2)}),
    ]
    for sp in pspecs:
        dt = getattr(torch, sp["dtype"])
        with quiet():
            layer = prune(sparsity=sp["sparsity"], dimensions=set(sp["dims"]), start=sp["start"], interval=sp["interval"],
                          repetition=sp["repetition"], callback=MagnitudePruningCallback())
        layer.train()
        k = f"c{idx}_"
        scale = torch.linspace(0.25, 4, sp["shape"][1]).view(1, -1, 1, 1)
        for s in range(sp["steps"] + 1):
            if s == sp["steps"]:
                layer.eval()
            if s in sp["writes"]:
                which, v = sp["writes"][s]
                if which == "n":
                    layer._n_updates.data[:] = v
                else:
                    layer.callback.t.data[:] = v
            x = (torch.randn(sp["shape"], generator=gen(17000 + 31 * idx + s)) * scale).to(dt).requires_grad_(True)
            gout = torch.randn(sp["shape"], generator=gen(18000 + 31 * idx + s)).to(dt)
            with quiet():
                y = layer(x)
            y.backward(gout)
            extra = [("mask", layer.mask), ("n_updates", layer._n_updates), ("cur_sparsity", layer._cur_sparsity),
                     ("t", layer.callback.t)]
            if hasattr(layer.callback, "magnitude"):
                extra.append(("magnitude", layer.callback.magnitude))
            record(k, s, layer, x, gout, y, extra)
        cases.append(dict(id=idx, op="prune", **{**sp, "shape": list(sp["shape"]),
                                                  "writes": {str(a): list(b) for a, b in sp["writes"].items()}},
                          total_steps=sp["steps"] + 1))
        idx += 1
    save("f14_state_writes", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F15: gradient-magnitude pruning -- MagnitudePruningCallback(use_gradient=True), qsparse/sparse.py:69-80.  The callback
# registers a tensor hook on the layer INPUT; the hook runs update_magnitude(grad) during backward, i.e. after the
# forward has already advanced `t` (:117), with the TOTAL gradient of the input (every consumer of x, not only x * mask).
# --------------------------------------------------------------------------------------------
def f15():
    store, cases = {}, []
    specs = [
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=2, interval=2, repetition=3, rampup=False,
             cb=dict(), dtype="float32", steps=12),
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.75, start=1, interval=2, repetition=2, rampup=True,
             cb=dict(), dtype="bfloat16", steps=10),
        dict(shape=(4, 12, 3, 5), dims=[1], sparsity=0.6, start=1, interval=2, repetition=3, rampup=False,
             cb=dict(mask_refresh_interval=2, stop_mask_refresh=6), dtype="bfloat16", steps=11),
        # non-negative upstream gradients with exact zeros: grad.min() == 0 switches the L0 variant on (:85-86)
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=1, interval=1, repetition=2, rampup=False,
             cb=dict(l0=True), dtype="float32", steps=8, relu_gout=True),
        # the input has a second consumer: the hook sees gout * mask + 0.5 * gout
        dict(shape=(2, 8, 6, 6), dims=[1], sparsity=0.5, start=1, interval=1, repetition=2, rampup=False,
             cb=dict(), dtype="float32", steps=8, residual=True),
        # steps 3 and 6 feed a tensor that does not require grad ("meeting no-grad tensor"), step 4 runs no backward
        dict(shape=(3, 10, 4, 4), dims=[1], sparsity=0.5, start=1, interval=1, repetition=2, rampup=False,
             cb=dict(), dtype="float32", steps=9, no_grad_steps=[3, 6], no_backward_steps=[4]),
        dict(shape=(6, 16), dims=[1], sparsity=0.5, start=1, interval=1, repetition=2, rampup=False,
             cb=dict(), dtype="bfloat16", steps=7),
        dict(shape=(2, 6, 5, 5), dims=[1, 2, 3], sparsity=0.7, start=1, interval=2, repetition=2, rampup=False,
             cb=dict(), dtype="float32", steps=8),
    ]
    for idx, sp in enumerate(specs):
        dt = getattr(torch, sp["dtype"])
        with quiet():
            layer = prune(sparsity=sp["sparsity"], dimensions=set(sp["dims"]), start=sp["start"], interval=sp["interval"],
                          repetition=sp["repetition"], rampup=sp["rampup"],
                          callback=MagnitudePruningCallback(use_gradient=True, **sp["cb"]))
        layer.train()
        k = f"c{idx}_"
        scale = torch.linspace(0.25, 4, sp["shape"][1]).view([1, -1] + [1] * (len(sp["shape"]) - 2))
        for s in range(sp["steps"] + 2):
            if s == sp["steps"]:
                layer.eval()
            needs_grad = s not in sp.get("no_grad_steps", [])
            runs_backward = needs_grad and s not in sp.get("no_backward_steps", [])
            x = (torch.randn(sp["shape"], generator=gen(9000 + 37 * idx + s)) * scale).to(dt).requires_grad_(needs_grad)
            gout = torch.randn(sp["shape"], generator=gen(9500 + 37 * idx + s)) * scale.flip(1)
            if sp.get("relu_gout"):
                gout = gout.relu()
            gout = gout.to(dt)
            with quiet():
                y = layer(x)
                out = y + x * 0.5 if sp.get("residual") else y
                if runs_backward:
                    out.backward(gout)
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_gout", gout)
            put(store, k + f"s{s}_y", y)
            if runs_backward:
                put(store, k + f"s{s}_gx", x.grad)
            put(store, k + f"s{s}_mask", layer.mask)
            put(store, k + f"s{s}_n_updates", layer._n_updates)
            put(store, k + f"s{s}_cur_sparsity", layer._cur_sparsity)
            put(store, k + f"s{s}_t", layer.callback.t)
            if hasattr(layer.callback, "magnitude"):
                put(store, k + f"s{s}_magnitude", layer.callback.magnitude)     # AFTER the backward of this step
        cases.append(dict(id=idx, **{**sp, "shape": list(sp["shape"])}, total_steps=sp["steps"] + 2))
    save("f15_prune_use_gradient", store, dict(cases=cases))


# --------------------------------------------------------------------------------------------
# F16: the reference's own config-1 recipe -- examples/mnist.py:193-199: convert(prune) + convert(quantize) and THEN
# devise_layerwise_pruning_schedule(start=2E, interval=0.4E, mask_refresh_interval=0.1E) -- on its MNIST Net (:17-44) with
# synthetic digits, Adadelta, dropout and batch norm live.  E (the epoch size) is scaled down so that the whole schedule
# runs in a few dozen steps; the four variants keep what matters about the real E = 938:
#   fractional        E = 7: interval 2.8, refresh interval 0.7 -- as with 938 (375.2 / 93.8) `t % interval == 0` never holds
#                     for t > 0, the second layer's start (17.8) is never a step index: masks stay all-ones
#   integer_default   E = 10, the recipe as written: interval 4.0, refresh 1.0 (floats, integer-valued).  The layers keep the
#                     `rampup_interval` of prune()'s DEFAULT interval (1000, quirk B10), so the first scheduled sparsity is
#                     0.75 * (1 - (1 - 1000/4)^3) and the first mask refresh raises IndexError -- recorded: step and message
#   integer_prunes    E = 10 with prune(..., interval=4): rampup_interval == interval, the schedule lands on 0.75 and prunes
#   weights           E = 10, convert(prune(0.5, interval=4), weight_layers=[Conv2d]): every PruneLayer is a `.prune`, so the
#                     schedule switches the callbacks to running_average=False (sparse.py:355-356)
# Recorded per step: the loss and every state tensor of every PruneLayer / QuantizeLayer (masks, magnitudes, scales, counters,
# `_cur_sparsity`, callback.t); per case: the schedule attributes the function wrote, str(model), the exception if any.
# --------------------------------------------------------------------------------------------
F16_CASES = (dict(name="fractional", E=7, steps=50, prune=dict(sparsity=0.75, dimensions=[1]), where="activations", seed=11),
             dict(name="integer_default", E=10, steps=40, prune=dict(sparsity=0.75, dimensions=[1]), where="activations", seed=12),
             dict(name="integer_prunes", E=10, steps=66, prune=dict(sparsity=0.75, dimensions=[1], interval=4), where="activations",
                  seed=13),
             dict(name="weights", E=10, steps=60, prune=dict(sparsity=0.5, interval=4), where="weights", seed=14))


# --------------------------------------------------------------------------------------------
# F17: the pair of F10 meeting non-finite values on pruned channels -- in evaluation with a finite scale (quirk B15:
# f32(INT_MIN) * s) and in training with a live scale (the scale turns NaN; the backward's clamp bounds with it)
# --------------------------------------------------------------------------------------------
def f17():
    store, cases = {}, []
    shape, C = (4, 16, 4, 4), 16
    specs = ((torch.bfloat16, "scaler", None, 8), (torch.bfloat16, "scaler", 3, 5), (torch.float32, "scaler", None, 6),
             (torch.bfloat16, "decimal", 3, 5), (torch.float32, "decimal", None, 8), (torch.float32, "scaler", 3, 8))
    for idx, (dt, kind, stop, inject_from) in enumerate(specs):
        with quiet():
            cb = MagnitudePruningCallback() if stop is None else MagnitudePruningCallback(stop_mask_refresh=stop)
            pl = prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2, callback=cb)
            ql = quantize(bits=4, channelwise=-1, timeout=1, callback=ScalerQuantizer() if kind == "scaler" else DecimalQuantizer())
        pl.train(), ql.train()
        steps = 8                    # the last one (index 8) in evaluation
        k = f"c{idx}_"
        for s in range(steps + 1):
            if s == steps:
                pl.eval(), ql.eval()
            x = (torch.randn(shape, generator=gen(9100 + 17 * idx + s)) * torch.linspace(0.25, 4, C).view(1, -1, 1, 1)).to(dt)
            if s >= inject_from:
                pruned = (~pl.mask.view(-1)).nonzero().view(-1).tolist()
                assert len(pruned) >= 3
                for j, c in enumerate(pruned[:3]):
                    x[j, c, j, j] = (float("nan"), float("inf"), float("-inf"))[j]
            x.requires_grad_(True)
            with quiet():
                y = ql(pl(x))
            gout = torch.randn(shape, generator=gen(9500 + 17 * idx + s)).to(y.dtype)
            y.backward(gout.clone())
            put(store, k + f"s{s}_x", x)
            put(store, k + f"s{s}_gout", gout)
            put(store, k + f"s{s}_y", y)
            put(store, k + f"s{s}_gx", x.grad)
            put(store, k + f"s{s}_mask", pl.mask)
            if hasattr(pl.callback, "magnitude"):
                put(store, k + f"s{s}_magnitude", pl.callback.magnitude)
            put(store, k + f"s{s}_scale", ql.weight)
        cases.append(dict(id=idx, dtype=str(dt).replace("torch.", ""), shape=list(shape), kind=kind, bits=4, sparsity=0.5,
                          start=1, interval=1, repetition=2, timeout=1, stop_mask_refresh=stop, inject_from=inject_from,
                          total_steps=steps + 1))
    save("f17_pair_non_finite", store, dict(cases=cases))


def f16_recipe(ns, case):
    """the recipe of examples/mnist.py:193-199 against the namespace `ns` (the reference here; tests/test_host_golden.py
    carries the same lines against the package)"""
    E = case["E"]
    kw = dict(case["prune"])
    if "dimensions" in kw:
        kw["dimensions"] = set(kw["dimensions"])
    torch.manual_seed(case["seed"])
    model = MnistNet()
    if case["where"] == "activations":
        model = ns.convert(model, ns.prune(**kw), activation_layers=[nn.ReLU], excluded_activation_layer_indexes=[(nn.ReLU, [-1])])
    else:
        model = ns.convert(model, ns.prune(**kw), weight_layers=[nn.Conv2d])
    model = ns.convert(model, ns.quantize(bits=4, channelwise=-1, timeout=5 * E), activation_layers=[nn.ReLU],
                       weight_layers=[nn.Conv2d, nn.Linear], input=True)
    return ns.devise_layerwise_pruning_schedule(model, start=2 * E, interval=0.4 * E, mask_refresh_interval=0.1 * E)


def f16_batches(case, batch=8):
    g = gen(1600 + case["seed"])
    protos = torch.randn(10, 1, 28, 28, generator=g)
    for _ in range(case["steps"]):
        y = torch.randint(0, 10, (batch,), generator=g)
        yield protos[y] + 0.5 * torch.randn(batch, 1, 28, 28, generator=g), y


def f16_operator_state(model, prune_cls, quant_cls):
    out = {}
    for path, m in model.named_modules():
        if isinstance(m, (prune_cls, quant_cls)):
            for k, v in m.state_dict().items():
                out[f"{path}.{k}"] = v
    return out


def f16():
    rs, rq = sys.modules["qsparse.sparse"], sys.modules["qsparse.quantize"]     # (`qsparse.quantize` the attribute is the function)

    class NS:
        convert, prune, quantize = staticmethod(convert), staticmethod(prune), staticmethod(quantize)
        devise_layerwise_pruning_schedule = staticmethod(devise_layerwise_pruning_schedule)

    store, cases = {}, []
    for idx, case in enumerate(F16_CASES):
        with quiet():
            model = f16_recipe(NS, case)
        players = [(p, m) for p, m in model.named_modules() if isinstance(m, rs.PruneLayer)]
        sched = [dict(path=p, start=m.start, interval=m.interval, repetition=m.repetition, schedules=list(m.schedules),
                      rampup_interval=m.rampup_interval, mask_refresh_interval=m.callback.mask_refresh_interval,
                      stop_mask_refresh=m.callback.stop_mask_refresh, running_average=bool(m.callback.running_average))
                 for p, m in players]
        opt = torch.optim.Adadelta(model.parameters(), lr=1.0)
        model.train()
        k = f"c{idx}_"
        error = None
        done = 0
        series = {}
        torch.manual_seed(100 + case["seed"])               # dropout
        for s, (x, y) in enumerate(f16_batches(case)):
            opt.zero_grad()
            try:
                with quiet():
                    loss = F.nll_loss(model(x), y)
            except Exception as e:                          # noqa: BLE001  (the reference's own failure is the datum)
                error = dict(step=s, type=type(e).__name__, message=str(e))
                break
            loss.backward()
            opt.step()
            series.setdefault("loss", []).append(np.float64(loss.item()))
            for name, v in f16_operator_state(model, rs.PruneLayer, rq.QuantizeLayer).items():
                series.setdefault(name, []).append(v.detach().cpu().numpy().copy())
            done = s + 1
        for name, vals in series.items():                   # one array per state tensor: [steps, *shape] (a tensor whose shape
            try:                                            # changes on first use -- the placeholders -- is stored per step)
                store[k + name] = np.stack(vals)
            except ValueError:
                for s, v in enumerate(vals):
                    store[k + f"{name}@{s}"] = v
        cases.append(dict(id=idx, **case, steps_done=done, error=error, schedule=sched, tree=str(model),
                          state_keys=sorted(f16_operator_state(model, rs.PruneLayer, rq.QuantizeLayer))))
        print(f"  F16 {case['name']}: {done} steps, error={error}, kept="
              f"{[int(m.mask.sum()) if m.mask.dim() else None for _, m in players]}")
    save("f16_mnist_layerwise_recipe", store, dict(cases=cases))


def f18():
    """F18 error behaviour: what the reference does for the misuse scenarios of tests/golden/error_scenarios.py (argument errors:
    exception type and first line of the message, or "ok")"""
    sys.path.insert(0, HERE)
    import error_scenarios as E
    cases = []
    for fn in E.SCENARIOS:
        torch.manual_seed(0)
        try:
            with quiet(), contextlib.redirect_stderr(io.StringIO()):
                fn(qsparse, "cpu")
            outcome = ["ok"]
        except BaseException as e:      # noqa: BLE001  (AssertionError, KeyError, ... are the point)
            outcome = ["raised", type(e).__name__, (str(e).splitlines() or [""])[0][:120]]
        cases.append({"name": fn.__name__, "outcome": outcome})
    save("f18_error_behaviour", {"count": np.array(len(cases))}, {"cases": cases})


if __name__ == "__main__":
    torch.set_num_threads(1)
    if len(sys.argv) > 1:                                   # regenerate chosen fixtures only: generate.py f16 ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    f1_f2()
    f3()
    f4()
    f5()
    f6()
    f7()
    f8_f9()
    f10()
    f11()
    f12()
    f13()
    f14()
    f15()
    f16()
    f17()
    f18()
