"""The caller's activation in a site's backward (ABI v26, `qs_ste_relu_bwd_args::act_x` / `qs_site_bwd_args::act_x`; option
`act_backward`): `convert(..., activation_layers=[nn.GELU])` puts the operators behind an activation the kernels do not fold.  Its
forward stays ATen's pass; its backward, `gelu_backward(g_h, x)`, is evaluated by the site's backward kernel on the gradient it holds
in registers.  Everything here is bit for bit against ATen's own GPU kernels -- the module-by-module route of the same package,
which the other suites hold against the oracle and the reference:

  * every bf16 / fp16 input pattern and 2^22 float32 values through the kernel (the probe of tools/probes/probe_gelu_bits.py);
  * every channel mode of the element-wise kernels (tensor-wise, rows, ragged rows, last dim), every gradient form (fp32, x's dtype,
    fp32 + 2-byte, 2-byte alone), clamp bounds that bite, channel masks;
  * converted sites (token-major and NCHW, autocast and float32) trained with the option on and off: outputs, input gradients,
    consumers' weight gradients, operator state; the test asserts that the fused route really ran."""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from qsparse_amd import _hip, fused
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEV = "cuda"
INF = float("inf")


def gen(seed):
    return torch.Generator().manual_seed(seed)


def aten_gelu_backward(dy, x):
    """ATen's gelu_backward as a FULL block of its vectorised kernel evaluates it.  For bf16 and float32 that is the function
    everywhere; for float16 the tail block of the same kernel rounds differently (2 of 20,000 pairs: tools/probes/probe_gelu_tail.py,
    `test_atens_fp16_tail_block_is_another_function` below), which is why the host side leaves float16 sites to ATen's own pass."""
    if x.dtype != torch.float16:
        return torch.ops.aten.gelu_backward(dy, x)
    n = x.numel()
    pad = (-n) % 8192
    dyp = torch.cat([dy.reshape(-1), dy.new_zeros(pad)])
    xp = torch.cat([x.reshape(-1), x.new_zeros(pad)])
    return torch.ops.aten.gelu_backward(dyp, xp)[:n].view(x.shape)


def want_of(g, g2, x, step, lo, hi, mask, ci):
    """the unfused chain: autograd's sum of the two shares, the STE backward + mask in x's dtype, ATen's gelu_backward"""
    total = g2.float() if g is None else (g if g2 is None else g + g2.float())
    gh = _hip.ste_bwd(total, step, False, -1, lo, hi, False, x.dtype, chan_mask=mask, mask_channel_index=ci)
    return aten_gelu_backward(gh, x)


def test_atens_fp16_tail_block_is_another_function():
    """(documents the reason for the float16 exclusion; if a torch build ever makes the two agree this test says so)"""
    g = gen(0)
    bad = {torch.float16: 0, torch.bfloat16: 0}
    for dt in bad:
        for _ in range(20):
            x = (torch.randn(1 << 20, generator=g) * 2).to(dt).to(DEV)
            dy = torch.randn(1 << 20, generator=g).to(dt).to(DEV)
            full = torch.ops.aten.gelu_backward(dy, x)[:1000]
            tail = torch.ops.aten.gelu_backward(dy[:1000].clone(), x[:1000].clone())
            bad[dt] += int((full.view(torch.int16) != tail.view(torch.int16)).sum())
    assert bad[torch.bfloat16] == 0
    if bad[torch.float16] == 0:
        pytest.skip("this torch evaluates fp16 gelu_backward identically in full and tail blocks: the fp16 exclusion could be lifted")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_every_two_byte_input_pattern(dtype):
    x = torch.arange(65536, dtype=torch.int32, device=DEV).to(torch.int16).view(dtype).view(256, 256).contiguous()
    for trial, scale in enumerate((None, 0.01, 1.0, 100.0)):
        g = torch.ones(256, 256) if scale is None else torch.randn(256, 256, generator=gen(trial)) * scale
        for gd in (torch.float32, dtype):
            gg = g.to(gd).to(DEV)
            got = _hip.ste_act_bwd(gg, x, 1.0, False, -INF, INF)
            assert same(got.cpu(), aten_gelu_backward(gg.to(dtype), x).cpu()), (dtype, trial, gd)


def test_float32_values():
    x = (torch.randn(1 << 22, generator=gen(3)) * 3).to(DEV)
    x[:8] = torch.tensor([0.0, -0.0, INF, -INF, float("nan"), 1e-30, -1e-30, 40.0], device=DEV)
    g = torch.randn(1 << 22, generator=gen(4)).to(DEV)
    assert same(_hip.ste_act_bwd(g, x, 1.0, False, -INF, INF).cpu(), torch.ops.aten.gelu_backward(g, x).cpu())


SHAPES = [((6, 16, 8, 8), 1, "rows"), ((5, 12, 7, 3), 1, "ragged rows"), ((40, 64), 1, "last dim"), ((33, 24), 1, "last dim, ragged tail"),
          ((4, 9, 20), 2, "token-major"), ((3, 1000), None, "tensor-wise"), ((2, 16, 4, 4), 1, "rows of 16 (4 per lane in float32)")]


@pytest.mark.parametrize("shape,ci,what", SHAPES, ids=[s[2] for s in SHAPES])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("form", ["f32", "same", "f32+g2", "g2"])
def test_channel_modes_and_gradient_forms(shape, ci, what, dtype, form):
    x = (torch.randn(shape, generator=gen(11)) * 2).to(dtype).to(DEV)
    g = torch.randn(shape, generator=gen(12)) * 3
    g2dt = torch.float16 if dtype == torch.float16 else torch.bfloat16
    g2 = (torch.randn(shape, generator=gen(13)) * 2).to(g2dt).to(DEV) if "g2" in form else None
    gg = None if form == "g2" else g.to(dtype if form == "same" else torch.float32).to(DEV)
    mask = None
    if ci is not None:
        mask = (torch.rand(shape[ci], generator=gen(14)) > 0.4).to(DEV)
    step, lo, hi = 0.25, -8.0, 7.0                 # the clamp bites: |g| up to ~12, bounds at -2 / 1.75
    got = _hip.ste_act_bwd(gg, x, step, False, lo, hi, chan_mask=mask, mask_channel_index=ci if ci is not None else 1, g2=g2)
    want = want_of(gg, g2, x, step, lo, hi, mask, ci)
    assert got.dtype == x.dtype and same(got.cpu(), want.cpu())


def test_rejected_operands():
    lib = _hip.load()
    x = torch.randn(64, device=DEV)
    a = _hip.SteReluBwdArgs()
    a.struct_size = __import__("ctypes").sizeof(_hip.SteReluBwdArgs)
    a.g, a.gx, a.nstep, a.step_host, a.outer, a.C, a.inner = x.data_ptr(), x.data_ptr(), 1, 1.0, 1, 1, 64
    a.act_x_kind = 1                                # a kind without its operand
    assert lib.qs_quant_ste_relu_bwd_v(__import__("ctypes").byref(a)) == -2
    a.act_x, a.act_x_kind = x.data_ptr(), 7         # an unknown kind
    assert lib.qs_quant_ste_relu_bwd_v(__import__("ctypes").byref(a)) == -2
    a.act_x, a.act_x_kind = x.data_ptr() + 4, 1     # misaligned
    assert lib.qs_quant_ste_relu_bwd_v(__import__("ctypes").byref(a)) == -3


# ---- converted sites ---------------------------------------------------------------------------------------------------------
class TokenSite(nn.Module):
    """Linear -> GELU -> [prune(last dim) -> quantize] -> Linear: one MLP block of a token-major encoder"""

    def __init__(self, dim, hidden, dims):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim)
        self.dims = dims

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def build(kind):
    torch.manual_seed(0)
    if kind == "tokens_q":               # the quantize-only recipe: Sequential(nn.GELU, QuantizeLayer)
        net = TokenSite(32, 64, {2})
    elif kind == "tokens":
        net = TokenSite(32, 64, {2})
        net = qs.convert(net, qs.prune(sparsity=0.5, dimensions={2}, start=1, interval=1, repetition=1), activation_layers=[nn.GELU], log=False)
    else:
        net = nn.Sequential(nn.Conv2d(8, 16, 3, padding=1), nn.GELU(), nn.Conv2d(16, 8, 3, padding=1))
        net = qs.convert(net, qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=1), activation_layers=[nn.GELU], log=False)
    net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1), activation_layers=[nn.GELU], log=False)
    return net.to(DEV).train()


@pytest.mark.parametrize("kind", ["tokens", "nchw", "nchw_cl", "tokens_q"])
@pytest.mark.parametrize("autocast", [None, torch.bfloat16, torch.float16])
def test_converted_site_trains_to_the_same_bits(kind, autocast, monkeypatch):
    # (MIOpen's default weight-gradient algorithms are not run-to-run deterministic: ask for the deterministic ones, as fuzz_image.py does)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)
    runs = {}
    for on in (False, True):
        qs.set_qsparse_options(act_backward=on)
        before = fused.ROUTES["act_backward"]
        net = build(kind if kind.startswith("tokens") else "nchw")
        if kind == "nchw_cl":
            net = net.to(memory_format=torch.channels_last)
        opt = torch.optim.SGD(net.parameters(), lr=0.05)
        outs = []
        for step in range(8):
            shape = (6, 10, 32) if kind.startswith("tokens") else (4, 8, 12, 12)
            x = torch.randn(shape, generator=gen(100 + step)).to(DEV)
            if kind == "nchw_cl":
                x = x.contiguous(memory_format=torch.channels_last)
            x.requires_grad_(True)
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=autocast or torch.bfloat16, enabled=autocast is not None):
                y = net(x)
            y.float().square().mean().backward()
            outs.append([("y", y.detach().float().cpu()), ("gx", x.grad.cpu())] + [(n, p.grad.cpu()) for n, p in net.named_parameters() if p.grad is not None])
            opt.step()
        used = fused.ROUTES["act_backward"] - before
        assert (used > 0) == (on and autocast != torch.float16), (on, used)      # (float16 sites: ATen's own pass, see above)
        runs[on] = (outs, {k: v.cpu() for k, v in net.state_dict().items()})
    qs.set_qsparse_options(act_backward=True)
    for s, (a, b) in enumerate(zip(runs[False][0], runs[True][0])):
        assert len(a) == len(b)
        for (name, u), (_, v) in zip(a, b):
            assert same(u, v), (s, name, int((u != v).sum()), u.numel())
    for k in runs[False][1]:
        assert same(runs[False][1][k], runs[True][1][k]), k


def test_with_the_statistics_exchange_live_the_slow_route_carries_the_gelu_too():
    """a data-parallel step (`sync_statistics="always"` in a one-rank group) never takes the steady-state fast path: the full route
    (`fused_prune_quantize(act_in=)`: two calls around the record exchange) hands the GELU's backward to the site as well"""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        qs.set_qsparse_options(sync_statistics="always")
        runs = {}
        for on in (False, True):
            qs.set_qsparse_options(act_backward=on)
            before = fused.ROUTES["act_backward"]
            net = build("tokens")
            outs = []
            for step in range(6):
                x = torch.randn((6, 10, 32), generator=gen(300 + step)).to(DEV).requires_grad_(True)
                net.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = net(x)
                y.float().square().mean().backward()
                outs.append([("y", y.detach().float().cpu()), ("gx", x.grad.cpu())] + [(n, p.grad.cpu()) for n, p in net.named_parameters() if p.grad is not None])
            assert (fused.ROUTES["act_backward"] - before > 0) == on
            runs[on] = (outs, {k: v.cpu() for k, v in net.state_dict().items()})
        for s, (a, b) in enumerate(zip(runs[False][0], runs[True][0])):
            for (name, u), (_, v) in zip(a, b):
                assert same(u, v), (s, name)
        for k in runs[False][1]:
            assert same(runs[False][1][k], runs[True][1][k]), k
    finally:
        qs.set_qsparse_options(act_backward=True)
        from qsparse_amd import util
        util._options_["sync_statistics"] = None          # (set_qsparse_options(x=None) leaves x untouched)
        dist.destroy_process_group()
