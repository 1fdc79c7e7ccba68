"""More than 2^31 -- and more than 2^32 -- elements in one activation (the GPU has 288 GB: a 2052 x 256 x 64 x 64 bf16 tensor is
4.3 GB, a 4212 x 256 x 64 x 64 one 8.8 GB).

Every kernel of the fused ReLU -> prune -> quantize site -- statistics, select, apply forward with the gate bitmap, backward --
addresses such a tensor with 64-bit element offsets and 32-bit group / row counters; nothing in the BASELINE configurations
comes near the limits.  One live training step in NCHW and in channels_last, checked chunk by chunk against torch's own
element-wise arithmetic on the GPU (exact: correctly rounded division, half-to-even, one rounding per operator -- the chain
of reference quantize.py:87-131 / sparse.py:263), the scale against the order-independent abs-max, the running magnitude of
four channels against the oracle's staged mean of that slice."""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
BITS = 4


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("batch", [2052, 4212])      # x 256 x 64 x 64: 2,151,677,952 > 2^31 and 4,416,602,112 > 2^32 elements
def test_one_live_step_on_more_than_2_pow_31_elements(batch, channels_last):
    SHAPE = (batch, 256, 64, 64)
    free, _ = torch.cuda.mem_get_info()
    if free < 48 * 2 ** 30 * (batch / 2052):
        pytest.skip("needs 48 GiB (100 GiB) of free device memory")
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        N, C, H, W = SHAPE
        assert N * C * H * W > 2 ** 31
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.empty(SHAPE, dtype=torch.bfloat16, device="cuda")
        for i in range(0, N, 171):               # filled in pieces: no float32 copy of the whole tensor
            part = torch.randn((min(171, N - i), C, H, W), generator=g, device="cuda")
            x[i:i + 171] = (part * torch.linspace(0.25, 4.0, C, device="cuda").view(1, C, 1, 1)).to(torch.bfloat16)
        del part
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        site = fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.75, dimensions={1}, start=0, interval=1, repetition=1)),
            qs.quantize(bits=BITS, channelwise=-1, timeout=1)).cuda().train())
        small = x[:2].clone()
        site(small)                                                                # the identity step of the quantizer (timeout 1)
        xg = x.requires_grad_(True)
        y = site(xg)                                                               # the live step on the big tensor
        assert y.dtype == torch.float32 and y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last) == channels_last
        gout = torch.empty(SHAPE, dtype=torch.float32, device="cuda")
        if channels_last:
            gout = gout.contiguous(memory_format=torch.channels_last)
        for i in range(0, N, 171):
            gout[i:i + 171] = torch.randn((min(171, N - i), C, H, W), generator=g, device="cuda") * 2
        y.backward(gout)
        gx = xg.grad
        p, q = site[0][1], site[1]
        mask, s = p.mask.view(1, C, 1, 1), q.weight.detach().view(())
        assert int(mask.sum()) == C // 4 and int(p._n_updates) == 2 and int(q._n_updates) == 2
        lo, hi = O.ste_bounds(BITS, s.cpu().view(1, 1))
        lo, hi = float(lo), float(hi)
        amax = torch.zeros((), device="cuda")
        with torch.no_grad():
            for i in range(0, N, 108):
                sl = slice(i, min(i + 108, N))
                h = torch.relu(x[sl]).float() * mask
                amax = torch.maximum(amax, h.amax())
                assert torch.equal(y[sl], torch.round(h / s) * s), ("forward", i)
                want = torch.where(x[sl] > 0, torch.clamp(gout[sl], lo, hi) * mask, torch.zeros((), device="cuda")).to(torch.bfloat16)
                assert torch.equal(gx[sl], want), ("backward", i)
        # the first live statistics step of the quantizer: scale = max|relu(x) * mask| / 2^(bits-1), exactly
        assert torch.equal(s, amax / 2 ** (BITS - 1))
        # running magnitude of four channels (two steps: the small tensor, then the big one) against the oracle's staged mean
        m_small = O.squeeze_mean(torch.relu(small[:, :4]).cpu().contiguous(memory_format=torch.channels_last if channels_last
                                                                           else torch.contiguous_format).abs(), (1, 4, 1, 1)).float().view(-1)
        xs = torch.relu(x.detach()[:, :4]).cpu()
        xs = xs.contiguous(memory_format=torch.channels_last) if channels_last else xs.contiguous()
        m_big = O.squeeze_mean(xs.abs(), (1, 4, 1, 1)).float().view(-1)
        ref = (1 * m_small + m_big) / 2          # sparse.py:89 at t = 1
        assert same(p.callback.magnitude.view(-1)[:4].cpu(), ref)
    finally:
        torch.set_num_threads(threads)


def _filled(shape, seed, dtype=torch.bfloat16, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty(shape, dtype=dtype, device="cuda")
    for i in range(0, shape[0], 171):
        n = min(171, shape[0] - i)
        x[i:i + n] = (torch.randn((n,) + tuple(shape[1:]), generator=g, device="cuda") * scale).to(dtype)
    return x


def test_functional_operators_and_reductions_on_more_than_2_pow_31_elements():
    """the other entry points on such a tensor: scaler / decimal / line quantizers with per-channel parameters and their STE
    backward, broadcast mask apply, abs-max and min/max (tensor-wise and per channel), the plain staged mean"""
    from qsparse_amd import _hip
    from qsparse_amd.quantize import quantize_with_decimal, quantize_with_line, quantize_with_scaler
    from qsparse_amd.sparse import apply_mask
    from qsparse_amd.util import squeeze_tensor_to_shape
    free, _ = torch.cuda.mem_get_info()
    if free < 48 * 2 ** 30:
        pytest.skip("needs 48 GiB of free device memory")
    N, C, H, W = shape = (2052, 256, 64, 64)
    x = _filled(shape, 11, scale=2.0)
    gen = torch.Generator(device="cuda").manual_seed(12)
    chunks = [slice(i, min(i + 108, N)) for i in range(0, N, 108)]
    # per-channel scaler, 8 bits, forward + backward
    s = (torch.rand(C, 1, generator=gen, device="cuda") * 0.05 + 0.01)
    xg = x.requires_grad_(True)
    y = quantize_with_scaler(xg, 8, s, 1)
    gout = _filled(shape, 13, torch.float32, 3.0)
    y.backward(gout)
    sv = s.view(1, C, 1, 1)
    with torch.no_grad():
        for sl in chunks:
            assert torch.equal(y[sl], torch.round(x[sl].float() / sv) * sv), ("scaler", sl)
            assert torch.equal(xg.grad[sl], torch.minimum(torch.maximum(gout[sl], -128 * sv), 127 * sv).to(torch.bfloat16)), ("ste", sl)
    x = x.detach()
    del y, xg
    # per-channel decimal (truncation), tensor-wise line quantizer in evaluation form
    d = torch.randint(2, 7, (C, 1), generator=gen, device="cuda").float()
    y = quantize_with_decimal(x, 8, d, 1)
    dv = d.view(1, C, 1, 1)
    for sl in chunks:
        assert torch.equal(y[sl], torch.trunc(x[sl].float() * 2.0 ** dv) * 2.0 ** -dv), ("decimal", sl)
    del y
    lines = torch.tensor([[-1.5, 2.25]], device="cuda")
    y = quantize_with_line(x, 4, lines, -1, False, True)
    lo_t, hi_t = lines[0, 0], lines[0, 1]
    step = (hi_t - lo_t) / 16                     # a TENSOR divisor: torch divides by a Python scalar on the GPU through its reciprocal
    for sl in chunks:
        xc = torch.clamp(x[sl].float(), -1.5, 2.25)
        assert torch.equal(y[sl], ((xc - lo_t) / step).round().clamp(0, 15) * step + lo_t), ("line", sl)
    del y
    # mask apply with a (1, C, 1, W) mask
    mask = torch.rand(1, C, 1, W, generator=gen, device="cuda") > 0.4
    y = apply_mask(x, mask)
    for sl in chunks:
        assert torch.equal(y[sl], x[sl] * mask), ("mask", sl)
    del y
    # order-independent reductions
    assert torch.equal(_hip.absmax(x, -1).view(()), x.abs().amax().float())
    assert torch.equal(_hip.absmax(x, 1), torch.stack([x[sl].abs().amax(dim=(0, 2, 3)) for sl in chunks]).amax(0).float())
    lo, hi = _hip.minmax(x, -1)
    assert float(lo) == float(x.amin()) and float(hi) == float(x.amax())
    lo, hi = _hip.minmax(x, 1)
    assert torch.equal(lo.view(-1), torch.stack([x[sl].amin(dim=(0, 2, 3)) for sl in chunks]).amin(0).float())
    assert torch.equal(hi.view(-1), torch.stack([x[sl].amax(dim=(0, 2, 3)) for sl in chunks]).amax(0).float())
    # the staged mean of |x| down to (1, C, 1, 1): four channels against the oracle, all of them against float64 sums
    m = squeeze_tensor_to_shape(x.abs(), (1, C, 1, 1)).float().view(-1)
    ref = O.squeeze_mean(x[:, :4].abs().cpu().contiguous(), (1, 4, 1, 1)).float().view(-1)
    assert same(m[:4].cpu(), ref)
    exact = torch.stack([x[sl].abs().double().sum(dim=(0, 2, 3)) for sl in chunks]).sum(0) / (N * H * W)
    assert torch.allclose(m.double(), exact, rtol=2e-2)       # (bf16 stages: three roundings to 8 bits)


@pytest.mark.parametrize("ties", [False, True])
def test_mask_from_importance_of_150_million_entries(ties):
    """unstructured pruning of an embedding-sized weight (reference util.py:103-117: sort, threshold at rank k, `>=`): the
    multi-block radix select against torch's own order statistic on the GPU"""
    from qsparse_amd.util import calculate_mask_given_importance, threshold_rank
    n = 150_000_001
    g = torch.Generator(device="cuda").manual_seed(21)
    imp = torch.randn(n, generator=g, device="cuda").abs()
    if ties:
        imp = (imp * 64).round() / 64
    for sparsity in (0.6, 0.999):
        mask = calculate_mask_given_importance(imp, sparsity)
        k = threshold_rank(sparsity, n)
        thr = torch.kthvalue(imp, k + 1).values          # ascending position k (0-based) = the (k+1)-th smallest
        assert torch.equal(mask, imp >= thr), (ties, sparsity)
        assert int(mask.sum()) >= n - k


@pytest.mark.parametrize("shape", [(64, 100_003), (8, 70_001, 2, 4)])
def test_sites_with_a_hundred_thousand_channels(shape):
    """channel counts beyond the fused pair's 65,536-channel select (a wide nn.Linear's activation): the sites fall back to their
    multi-block kernels; prune -> quantize trajectories on the GPU equal the CPU path's (the op-by-op mirror of the reference)"""
    import copy
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        torch.manual_seed(0)
        cpu = fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.7, dimensions={1}, start=1, interval=1, repetition=2)),
            qs.quantize(bits=6, channelwise=-1, timeout=1)).train())
        gpu = copy.deepcopy(cpu).cuda()
        g = torch.Generator().manual_seed(3)
        scale = torch.linspace(0.2, 3.0, shape[1]).view([1, -1] + [1] * (len(shape) - 2))
        for step in range(4):
            x = (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)
            go = torch.randn(shape, generator=g) * 4
            outs = []
            for net, dev in ((cpu, "cpu"), (gpu, "cuda")):
                xd = x.clone().to(dev).requires_grad_(True)
                y = net(xd)
                y.backward(go.clone().to(dev).to(y.dtype))
                outs.append((y.detach().cpu(), xd.grad.cpu()))
            assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1]), step
            for (ka, va), (kb, vb) in zip(cpu.state_dict().items(), gpu.state_dict().items()):
                assert ka == kb and same(va, vb.cpu()), (step, ka)
    finally:
        torch.set_num_threads(threads)


def test_weight_path_with_a_layer_of_more_than_2_pow_31_weights():
    """the multi-tensor weight path (qs_multi_absmax / _scale_update / _quant_fwd, qs_multi_ste_bwd) on a 70,001 x 32,768 linear layer
    (2,293,792,768 weights) next to a small one: scale, quantized weight and weight gradient equal the layer-by-layer route's"""
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GiB of free device memory")
    res = []
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            torch.manual_seed(0)
            big = nn.Linear(32768, 70001, bias=False, device="cuda")
            with torch.no_grad():
                big.weight.copy_(_filled(tuple(big.weight.shape), 31, torch.float32, 0.02))
            net = nn.Sequential(nn.Linear(16, 32768, device="cuda"), big)
            net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1), weight_layers=[nn.Linear], log=False).train()
            assert (net.__dict__.get("_qs_weight_batcher") is not None) == batched
            g = torch.Generator().manual_seed(4)
            for _ in range(3):
                net.zero_grad(set_to_none=True)
                (net(torch.randn(2, 16, generator=g).cuda()) * 1e-3).sum().backward()
            net.eval()
            res.append((net[1].quantize.weight.detach().clone(), net[1]._parameters["weight"].grad[::4099].clone(),
                        net[1].weight[::4099].detach().clone(), net[0].quantize.weight.detach().clone(),
                        float(net[1]._parameters["weight"].grad.abs().amax())))
            del net, big
            torch.cuda.empty_cache()
        finally:
            qs.set_qsparse_options(batch_weights=True)
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert res[0][4] == res[1][4] and res[0][4] > 0
