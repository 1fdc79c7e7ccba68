"""Pins the CPU oracle (oracle/qs_oracle.py) to the golden vectors recorded from the real reference.

CPU-only.  Every comparison is bit-for-bit: the oracle restates the same ATen operator chain, so
on the same torch build nothing may differ.
"""
import pytest
import torch

from golden_io import Golden, same, tdtype
from oracle import qs_oracle as O


def _param(g, k, c):
    p = g.get(k + "param")
    if c["pkind"] == "pyfloat":
        return float(p)
    if c["pkind"] == "zerodim":
        return torch.tensor(p, dtype=torch.float32) if not isinstance(p, torch.Tensor) else p
    return p


def test_f1_scaler():
    g = Golden("f1_scaler")
    for c in g.cases:
        k = f"c{c['id']}_"
        x, p = g.get(k + "x"), _param(g, k, c)
        y = O.scaler_fwd(x, c["bits"], p, c["channel_index"])
        assert same(y, g.get(k + "y")), c
        assert same(O.scaler_codes(x, p, c["channel_index"]), g.get(k + "codes")), c
        gx = O.ste_bwd(g.get(k + "gout"), c["bits"], p, c["channel_index"], c["flip_axis"],
                       c["backward_passthrough"], x.dtype)
        assert same(gx, g.get(k + "gx")), c


def test_f2_decimal():
    g = Golden("f2_decimal")
    for c in g.cases:
        k = f"c{c['id']}_"
        x, p = g.get(k + "x"), _param(g, k, c)
        if c["pkind"] == "pyfloat":
            p = int(p)
        y = O.decimal_fwd(x, c["bits"], p, c["channel_index"])
        assert same(y, g.get(k + "y")), c
        assert same(O.decimal_codes(x, p, c["channel_index"]), g.get(k + "codes")), c
        gx = O.ste_bwd(g.get(k + "gout"), c["bits"], 2.0 ** -p, c["channel_index"], c["flip_axis"],
                       c["backward_passthrough"], x.dtype)
        assert same(gx, g.get(k + "gx")), c


def test_f3_line():
    g = Golden("f3_line")
    for c in g.cases:
        k = f"c{c['id']}_"
        y = O.line_fwd(g.get(k + "x"), c["bits"], g.get(k + "lines"), c["channel_index"], c["float_zero_point"])
        assert same(y, g.get(k + "y")), c


def test_f4_quantize_layer_trajectories():
    g = Golden("f4_quantize_layer")
    for c in g.cases:
        if c.get("kind") == "conv_weight":
            continue
        k = f"c{c['id']}_"
        sim = O.QuantizeSim(c["cb"], c["bits"], c["channelwise"], c["timeout"], batch_dimension=0)
        for s in range(c["total_steps"]):
            y = sim.step(g.get(k + f"s{s}_x"), training=s < c["steps"])
            assert same(y, g.get(k + f"s{s}_y")), (c, s)
            assert same(sim.weight, g.get(k + f"s{s}_weight")), (c, s)
            assert sim.n_updates == int(g.get(k + f"s{s}_n_updates")[0]), (c, s)


def test_f4_conv_weight_quantization():
    g = Golden("f4_quantize_layer")
    for c in g.cases:
        if c.get("kind") != "conv_weight":
            continue
        k = f"c{c['id']}_"
        w0, b0 = g.get(k + "w0"), g.get(k + "b0")
        shared = {"t": 0}      # weight and bias layers share one callback and its counter (quirk B8)
        wq = O.QuantizeSim(c["cb"], 8, 0, c["timeout"], batch_dimension=-1, shared=shared)
        bq = O.QuantizeSim(c["cb"], 8, 0, c["timeout"], batch_dimension=-1, shared=shared)
        for s in range(c["steps"]):
            x = g.get(k + f"s{s}_x")
            y = torch.nn.functional.conv2d(x, wq.step(w0), bq.step(b0))     # Conv2d.forward reads weight then bias
            assert same(y, g.get(k + f"s{s}_y")), (c, s)
            assert same(wq.step(w0), g.get(k + f"s{s}_qweight")), (c, s)    # the fixture read .weight/.bias once more
            assert same(bq.step(b0), g.get(k + f"s{s}_qbias")), (c, s)
            assert same(wq.weight, g.get(k + f"s{s}_wscale")), (c, s)
            assert same(bq.weight, g.get(k + f"s{s}_bscale")), (c, s)


def test_f5_squeeze():
    g = Golden("f5_squeeze")
    for c in g.cases:
        k = f"c{c['id']}_"
        out = O.squeeze_mean(g.get(k + "x").abs(), c["mask_shape"])
        assert same(out, g.get(k + "out")), c


def test_f6_mask():
    import numpy as np
    g = Golden("f6_mask")
    for c in g.cases:
        imp = g.get(c["imp_key"])
        want = np.unpackbits(g.z[f"c{c['id']}_mask"])[: imp.numel()].astype(bool).reshape(tuple(imp.shape))
        got = O.mask_from_importance(imp, c["sparsity"])
        assert (got.numpy() == want).all(), c


def test_f7_prune_layer_trajectories():
    g = Golden("f7_prune_layer")
    for c in g.cases:
        if c.get("kind") == "conv_weight":
            continue
        k = f"c{c['id']}_"
        sim = O.PruneSim(c["sparsity"], c["dims"], c["start"], c["interval"], c["repetition"], c["rampup"], **c["cb"])
        for s in range(c["total_steps"]):
            training = s < c["steps"]
            x = g.get(k + f"s{s}_x")
            n_before = sim.n_updates
            y = sim.step(x, training)
            assert same(y, g.get(k + f"s{s}_y")), (c, s)
            assert same(sim.mask, g.get(k + f"s{s}_mask")), (c, s)
            assert sim.n_updates == int(g.get(k + f"s{s}_n_updates")[0])
            assert sim.cur_sparsity == float(g.get(k + f"s{s}_cur_sparsity")[0])
            assert sim.t == int(g.get(k + f"s{s}_t")[0])
            if g.has(k + f"s{s}_magnitude"):
                assert same(sim.magnitude, g.get(k + f"s{s}_magnitude")), (c, s)
            active = (not training) or n_before >= c["start"]
            assert same(sim.grad(g.get(k + f"s{s}_gout"), active), g.get(k + f"s{s}_gx")), (c, s)


def test_f15_prune_layer_use_gradient_trajectories():
    """MagnitudePruningCallback(use_gradient=True), reference sparse.py:69-80: the magnitude follows the input's gradient"""
    g = Golden("f15_prune_use_gradient")
    for c in g.cases:
        k = f"c{c['id']}_"
        sim = O.PruneSim(c["sparsity"], c["dims"], c["start"], c["interval"], c["repetition"], c["rampup"], use_gradient=True,
                         **c["cb"])
        for s in range(c["total_steps"]):
            training = s < c["steps"]
            x, gout = g.get(k + f"s{s}_x"), g.get(k + f"s{s}_gout")
            needs_grad = s not in c.get("no_grad_steps", [])
            n_before = sim.n_updates
            y = sim.step(x, training, requires_grad=needs_grad)
            assert same(y, g.get(k + f"s{s}_y")), (c, s)
            assert same(sim.mask, g.get(k + f"s{s}_mask")), (c, s)
            if g.has(k + f"s{s}_gx"):          # a backward ran: the hook saw the input's total gradient
                active = (not training) or n_before >= c["start"]
                gx = sim.grad(gout, active)
                if c.get("residual"):
                    gx = gx + gout * 0.5
                assert same(gx, g.get(k + f"s{s}_gx")), (c, s)
                if training and n_before >= c["start"]:
                    sim.receive_grad(gx)
            assert sim.n_updates == int(g.get(k + f"s{s}_n_updates")[0])
            assert sim.cur_sparsity == float(g.get(k + f"s{s}_cur_sparsity")[0])
            assert sim.t == int(g.get(k + f"s{s}_t")[0])
            if g.has(k + f"s{s}_magnitude"):
                assert same(sim.magnitude, g.get(k + f"s{s}_magnitude")), (c, s)


def test_f7_conv_weight_pruning():
    g = Golden("f7_prune_layer")
    for c in g.cases:
        if c.get("kind") != "conv_weight":
            continue
        k = f"c{c['id']}_"
        w0, x = g.get(k + "w0"), g.get(k + "x")
        torch.manual_seed(11)
        bias = torch.nn.Conv2d(10, 12, 3).bias.detach()
        sim = O.PruneSim(c["sparsity"], c["dims"], c["start"], c["interval"], c["repetition"], False, **c["cb"])
        for s in range(c["steps"]):
            y = torch.nn.functional.conv2d(x, sim.step(w0), bias)
            assert same(y, g.get(k + f"s{s}_y")), (c, s)
            assert same(sim.mask, g.get(k + f"s{s}_mask")), (c, s)


def test_f17_pair_with_non_finite_values_on_pruned_channels():
    """the reference's own outputs for NaN / Inf / -Inf on pruned channels: f32(INT_MIN) * s in evaluation (quirk B15), a NaN
    scale once a live scale has seen them (x * mask is NaN there), NaN clamp bounds in the backward from then on"""
    from golden_io import same_up_to_nan_payload as eq
    g = Golden("f17_pair_non_finite")
    seen_nan_scale = seen_int_min = 0
    for c in g.cases:
        k = f"c{c['id']}_"
        stop = c["stop_mask_refresh"]
        ps = O.PruneSim(c["sparsity"], [1], c["start"], c["interval"], c["repetition"], False,
                        **({} if stop is None else {"stop_mask_refresh": stop}))
        qs = O.QuantizeSim(c["kind"], c["bits"], -1, c["timeout"])
        for s in range(c["total_steps"]):
            training = s < c["total_steps"] - 1
            x = g.get(k + f"s{s}_x")
            n_before = ps.n_updates
            h = ps.step(x, training)
            y = qs.step(h, training)
            want = g.get(k + f"s{s}_y")
            assert eq(y, want), (c, s)
            assert same(ps.mask, g.get(k + f"s{s}_mask")), (c, s)
            assert eq(qs.weight, g.get(k + f"s{s}_scale")), (c, s)
            if g.has(k + f"s{s}_magnitude"):
                assert eq(ps.magnitude, g.get(k + f"s{s}_magnitude")), (c, s)
            gh = qs.grad(g.get(k + f"s{s}_gout"), h.dtype)
            gx = ps.grad(gh, (not training) or n_before >= c["start"])
            assert eq(gx, g.get(k + f"s{s}_gx")), (c, s)
            if s >= c["inject_from"]:
                scale = g.get(k + f"s{s}_scale")
                seen_nan_scale += int(bool(scale.isnan().any()))
                if not bool(scale.isnan().any()) and c["kind"] == "scaler":
                    assert float((want.float() == float(-2 ** 31) * float(scale)).sum()) >= 2, (c, s)      # NaN and Inf
                    seen_int_min += 1
    assert seen_nan_scale >= 3 and seen_int_min >= 2


def test_f10_prune_quant_pair():
    g = Golden("f10_prune_quant_pair")
    for c in g.cases:
        k = f"c{c['id']}_"
        ps = O.PruneSim(c["sparsity"], [1], c["start"], c["interval"], c["repetition"], False)
        qs = O.QuantizeSim("scaler", c["bits"], -1, c["timeout"])
        for s in range(c["total_steps"]):
            training = s < c["total_steps"] - 1
            x = g.get(k + f"s{s}_x")
            n_before = ps.n_updates
            h = ps.step(x, training)
            y = qs.step(h, training)
            assert same(y, g.get(k + f"s{s}_y")), (c, s)
            assert same(ps.mask, g.get(k + f"s{s}_mask")), (c, s)
            assert same(qs.weight, g.get(k + f"s{s}_scale")), (c, s)
            if g.has(k + f"s{s}_magnitude"):
                assert same(ps.magnitude, g.get(k + f"s{s}_magnitude")), (c, s)
            gh = qs.grad(g.get(k + f"s{s}_gout"), h.dtype)
            gx = ps.grad(gh, (not training) or n_before >= c["start"])
            assert same(gx, g.get(k + f"s{s}_gx")), (c, s)
