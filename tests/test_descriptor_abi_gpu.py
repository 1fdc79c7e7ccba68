"""The descriptor entry points of ABI v25 (`qs_*_v`, include/qsparse_hip.h "ABI compatibility") on the GPU: the same results as
the positional entry points they stand next to -- and, for the quantizer forward, as the oracle -- plus the compatibility rule
itself: a caller compiled against an older header (smaller `struct_size`) is served, whatever lies behind its struct is not read."""
import ctypes

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd import _hip
from qsparse_amd.fused import fuse_prune_quantize_pairs

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("kind", ["scaler", "decimal"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quant_fwd_descriptor_equals_the_oracle(kind, dtype):
    lib = _hip.load()
    x = (torch.randn(4, 8, 7, 7, generator=gen(0)) * 2).to(dtype)
    p = (torch.rand(8, 1, generator=gen(1)) * 0.1 + 0.01) if kind == "scaler" else torch.randint(0, 6, (8, 1), generator=gen(1)).float()
    xd, pd = x.cuda(), p.cuda()
    y = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    a = _hip.QuantFwdArgs()
    a.struct_size = ctypes.sizeof(a)
    a.kind = 0 if kind == "scaler" else 1
    a.x, a.y, a.param, a.nparam = xd.data_ptr(), y.data_ptr(), pd.data_ptr(), 8
    a.outer, a.C, a.inner = 4, 8, 49
    a.xdt, a.ydt, a.qdt = _hip._DT[dtype], _hip.F32, _hip.F32
    a.stream = _hip._stream(xd)
    assert lib.qs_quant_fwd_v(ctypes.byref(a)) == 0
    ref = O.scaler_fwd(x, 8, p, 1) if kind == "scaler" else O.decimal_fwd(x, 8, p, 1)
    assert same(y.cpu(), ref)


def test_pq_select_and_quantize_step_descriptors_equal_the_positional_calls():
    lib = _hip.load()
    C = 48
    stream = None

    def state():
        mag = (torch.rand(C, generator=gen(3)) + 0.1).cuda()
        stage = torch.rand(C, generator=gen(4)).bfloat16().cuda()
        mask = torch.ones(C, dtype=torch.bool).cuda()
        amax = (torch.rand(C, generator=gen(5)) * 3).cuda()
        scale = torch.full((1,), 0.25).cuda()
        return mag, stage, mask, amax, scale

    a_state, b_state = state(), state()
    mag, stage, mask, amax, scale = a_state
    assert lib.qs_pq_select(mag.data_ptr(), stage.data_ptr(), _hip.BF16, C, 1, 3, 1, 20, mask.data_ptr(), amax.data_ptr(), 1, 1, 2, 4,
                            scale.data_ptr(), None, None, None, None, None, None, _hip.BF16, None, 1, None, stream) == 0
    mag, stage, mask, amax, scale = b_state
    d = _hip.PqSelectArgs()
    d.struct_size = ctypes.sizeof(d)
    d.magnitude, d.stage_mean, d.sdt, d.C = mag.data_ptr(), stage.data_ptr(), _hip.BF16, C
    d.update_magnitude, d.t_mag, d.refresh_mask, d.k, d.mask = 1, 3, 1, 20, mask.data_ptr()
    d.chan_absmax, d.chan_absmax_stride, d.update_scale, d.t_q, d.bits, d.scale = amax.data_ptr(), 1, 1, 2, 4, scale.data_ptr()
    d.stat_dt, d.world = _hip.BF16, 1
    assert lib.qs_pq_select_v(ctypes.byref(d)) == 0
    torch.cuda.synchronize()
    for u, v in zip(a_state, b_state):
        assert same(u.cpu(), v.cpu())
    assert 0 < int(a_state[2].sum()) < C                       # the mask really was rebuilt

    x = (torch.randn(6, 16, 8, 8, generator=gen(6)) * 2).bfloat16().cuda()
    outs = []
    for descriptor in (False, True):
        y = torch.empty(x.shape, dtype=torch.float32, device="cuda")
        lines = torch.zeros(4, 32, device="cuda")
        sc = torch.full((1,), 0.5).cuda()
        if descriptor:
            q = _hip.QuantizeStepArgs()
            q.struct_size = ctypes.sizeof(q)
            q.x, q.y, q.amax_lines, q.lines, q.scale, q.numel = x.data_ptr(), y.data_ptr(), lines.data_ptr(), 4, sc.data_ptr(), x.numel()
            q.xdt, q.ydt, q.bits, q.t, q.update = _hip.BF16, _hip.F32, 4, 2, _hip.QSTEP_ALL
            assert lib.qs_quantize_step_v(ctypes.byref(q)) == 0
        else:
            assert lib.qs_quantize_step(x.data_ptr(), y.data_ptr(), None, lines.data_ptr(), 4, sc.data_ptr(), x.numel(), _hip.BF16, _hip.F32, 4,
                                        2, None, None, 0, _hip.QSTEP_ALL, 0, 0, 0, None, None, 0, None) == 0
        outs.append((y.cpu(), sc.cpu()))
    assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1])
    ref_scale = O.running_mean_absmax(torch.full((1, 1), 0.5), O.absmax_scale(x.cpu(), 4, -1, True), 2).to(torch.float32)
    assert same(outs[0][1].view(1, 1), ref_scale)


def test_site_descriptors_equal_the_positional_calls_and_ignore_what_lies_behind_an_older_struct():
    lib = _hip.load()
    p = qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1)
    q = qs.quantize(bits=4, channelwise=-1, timeout=1)
    site = fuse_prune_quantize_pairs(nn.Sequential(nn.Sequential(nn.Sequential(nn.ReLU(), p), q)))[0].cuda().train()
    shape = (6, 16, 8, 8)
    for s in range(4):          # into the steady state: plan built, mask and scale live
        site((torch.randn(shape, generator=gen(10 + s)) * torch.linspace(0.3, 3, 16).view(1, -1, 1, 1)).cuda())
    plan = q.__dict__["_qs_site_plan"]
    x = (torch.randn(shape, generator=gen(20)) * 2).cuda()
    ys, gates = [], []
    for descriptor in (False, True):        # a pure apply step (no statistics): nothing of the state moves
        y = torch.empty_like(x)
        bits = torch.zeros((x.numel() + 7) // 8, dtype=torch.uint8, device="cuda")
        if descriptor:
            a = _hip.SiteFwdArgs()
            a.struct_size = ctypes.sizeof(a)
            a.flags, a.x, a.y, a.gate_out, a.world = _hip.SITE_PRE_RELU, x.data_ptr(), y.data_ptr(), bits.data_ptr(), 1
            assert lib.qs_site_fwd_v(plan.ref, ctypes.byref(a)) == 0
        else:
            _hip.site_fwd(plan.ref, x, y, bits, _hip.SITE_PRE_RELU, 0, 0, 0)
        ys.append(y.cpu()), gates.append(bits)
    assert same(ys[0], ys[1]) and torch.equal(gates[0], gates[1])
    g = torch.randn(shape, generator=gen(21)).cuda()
    ref = torch.empty_like(x)
    _hip.site_bwd(plan.ref, g, gates[0], ref, 0, -8.0, 7.0)
    b = _hip.SiteBwdArgs()
    b.struct_size = _hip.SiteBwdArgs.g3.offset          # a caller compiled against the v24-era field list
    b.g3, b.gx_image, b.gx_image_dt = 0xdead0, 0xbeef0, 7     # garbage behind that caller's struct
    gx = torch.empty_like(x)
    b.flags, b.gdt, b.g, b.gate, b.gx, b.lo_mul, b.hi_mul = 0, _hip.F32, g.data_ptr(), gates[0].data_ptr(), gx.data_ptr(), -8.0, 7.0
    assert lib.qs_site_bwd_v(plan.ref, ctypes.byref(b)) == 0
    torch.cuda.synchronize()
    assert same(gx.cpu(), ref.cpu())
