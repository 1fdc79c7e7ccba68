"""Hardening: the zero-on-entry scratch accumulators and streams other than the default one.

The per-channel abs-max of a fused site rides in the statistics launch as an atomic max into a PERSISTENT buffer that the
select launch re-zeroes -- nothing initialises it per step.  A step that dies between the two (a failed collective, an
exception in user code, KeyboardInterrupt) used to leave it dirty: the next step silently max-accumulated the stale values into
the running scale.  Every such window is now bracketed by an "armed" flag (fused.py::_arm_accumulators, QuantizeLayer's
`_qs_accumulator_armed`); a step that finds it set re-zeroes first.  The tests kill a step inside the window -- on the
fine-grained route (a failing select), on the data-parallel split-composite route (a failing all-gather; the exchange is
driven in-process with a stand-in collective) and on the lone quantizer's exchange route -- with an input 100x larger than
the real one, then check that the following steps equal an undisturbed twin bit for bit.
"""
import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd import _hip, fused
from qsparse_amd import distributed as qdist
from qsparse_amd.fused import fuse_prune_quantize_pairs
from qsparse_amd.quantize import DecimalQuantizer, ScalerQuantizer

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEV = "cuda"


def gen(seed):
    return torch.Generator().manual_seed(seed)


def same(a, b):
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.view(torch.uint8).view(-1), b.view(torch.uint8).view(-1))


def _pair(quantizer=ScalerQuantizer):
    pair = nn.Sequential(nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={1}, start=1, interval=1, repetition=2)),
                         qs.quantize(bits=4, channelwise=-1, timeout=1, callback=quantizer())).to(DEV).train()
    return fuse_prune_quantize_pairs(pair)


def _x(step, shape=(8, 16, 12, 12), scale=1.0):
    x = torch.randn(shape, generator=gen(100 + step)) * torch.linspace(0.3, 3, shape[1]).view([1, -1] + [1] * (len(shape) - 2))
    return (x * scale).bfloat16().to(DEV)


def _step(pair, x):
    xg = x.clone().requires_grad_(True)
    y = pair(xg)
    y.backward(torch.ones_like(y))
    return y.detach(), xg.grad


def _state(pair):
    cb = pair[0][1].callback
    return [pair[0][1].mask, getattr(cb, "magnitude", cb.t), pair[1].weight, pair[0][1]._n_updates, pair[1]._n_updates, cb.t]


def _check_twins(disturbed, twin, first, last, **kw):
    for s in range(first, last):
        (ya, ga), (yb, gb) = _step(disturbed, _x(s, **kw)), _step(twin, _x(s, **kw))
        assert same(ya, yb) and same(ga, gb), s
        for a, b in zip(_state(disturbed), _state(twin)):
            assert same(a, b), s


@pytest.mark.parametrize("quantizer,shape", [(DecimalQuantizer, (8, 16, 12, 12)), (ScalerQuantizer, (24, 16))])
def test_failing_select_on_the_fine_grained_route_leaves_no_stale_statistics(monkeypatch, quantizer, shape):
    """the fine-grained route (sites the composite call does not cover; forced here): statistics launch(es), then `qs_pq_select`"""
    monkeypatch.setattr(fused, "_site_plan", lambda *a, **k: None)
    disturbed, twin = _pair(quantizer), _pair(quantizer)
    _check_twins(disturbed, twin, 0, 4, shape=shape)
    real = _hip.pq_select
    calls = []

    def failing(*a, **k):
        calls.append(1)
        raise RuntimeError("injected: the step dies between the statistics launch and the select")

    monkeypatch.setattr(_hip, "pq_select", failing)
    with pytest.raises(RuntimeError, match="injected"):
        disturbed(_x(99, shape=shape, scale=100.0))          # statistics of a 100x larger input land in the accumulator ...
    assert calls
    monkeypatch.setattr(_hip, "pq_select", real)
    acc = disturbed[1].__dict__.get("_chan_absmax")
    if acc is None:
        acc = disturbed[1].__dict__["_chan_absmax_dense"]
    assert float(acc.max()) > 50.0                           # ... and are still there: the window the flag guards
    _check_twins(disturbed, twin, 4, 8, shape=shape)          # the next steps do not see them
    assert float(acc.max()) == 0.0


@pytest.mark.parametrize("fmt", [torch.contiguous_format, torch.channels_last])
def test_failing_collective_on_the_split_composite_route(monkeypatch, fmt):
    """data-parallel site step: qs_site_stats -> all-gather -> qs_site_fwd(QS_SITE_STATS_DONE).  The exchange is driven
    in-process (a one-rank stand-in for the collective: gathered <- record), which also pins that the two-call route equals
    the one-call composite bit for bit; then the collective fails once"""
    disturbed, twin = _pair(), _pair()
    log = []

    def gather(gathered, record):
        log.append("gather")
        gathered.copy_(record)

    real_stats = _hip.site_stats

    def stats(*a, **k):
        log.append("stats")
        return real_stats(*a, **k)

    def run(pair, x):
        return _step(pair, x.contiguous(memory_format=fmt))

    for s in range(3):                                       # schedule phase: both on the ordinary routes
        (ya, ga), (yb, gb) = run(disturbed, _x(s)), run(twin, _x(s))
        assert same(ya, yb)
    monkeypatch.setattr(qdist, "exchange_active", lambda world=None: True)
    monkeypatch.setattr(qdist, "all_gather_records", gather)
    monkeypatch.setattr(_hip, "site_stats", stats)
    for s in range(3, 6):                                    # `disturbed` exchanges, `twin` does not: same bits
        ya, ga = run(disturbed, _x(s))
        monkeypatch.setattr(qdist, "exchange_active", lambda world=None: False)
        yb, gb = run(twin, _x(s))
        monkeypatch.setattr(qdist, "exchange_active", lambda world=None: True)
        assert same(ya, yb) and same(ga, gb), s
        for a, b in zip(_state(disturbed), _state(twin)):
            assert same(a, b), s
    assert log == ["stats", "gather"] * 3                    # one statistics call + one collective per live step

    def broken(gathered, record):
        raise RuntimeError("injected: the all-gather failed")

    monkeypatch.setattr(qdist, "all_gather_records", broken)
    with pytest.raises(RuntimeError, match="injected"):
        disturbed(_x(99, scale=100.0).contiguous(memory_format=fmt))
    monkeypatch.setattr(qdist, "all_gather_records", gather)
    for s in range(6, 9):
        ya, ga = run(disturbed, _x(s))
        monkeypatch.setattr(qdist, "exchange_active", lambda world=None: False)
        yb, gb = run(twin, _x(s))
        monkeypatch.setattr(qdist, "exchange_active", lambda world=None: True)
        assert same(ya, yb) and same(ga, gb), s
        for a, b in zip(_state(disturbed), _state(twin)):
            assert same(a, b), s


def test_failing_all_reduce_of_a_lone_quantizer(monkeypatch):
    """quantize-only activation site with the exchange live: abs-max launch -> all-reduce (MAX) of the accumulator lines ->
    running scale + quantization"""
    a, b = (qs.quantize(bits=8, channelwise=-1, timeout=1).to(DEV).train() for _ in range(2))
    for s in range(3):
        assert same(a(_x(s)), b(_x(s)))
    monkeypatch.setattr(qdist, "exchange_active", lambda world=None: True)
    reduced = []
    monkeypatch.setattr(qdist, "allreduce_max_", lambda t, world=None: reduced.append(t.dtype) or t)
    ya = a(_x(3))
    monkeypatch.setattr(qdist, "exchange_active", lambda world=None: False)
    assert same(ya, b(_x(3))) and same(a.weight, b.weight)
    assert reduced == [torch.int32]                          # the lines travel as integers: a NaN maximum survives any MAX
    monkeypatch.setattr(qdist, "exchange_active", lambda world=None: True)

    def broken(t, world=None):
        raise RuntimeError("injected: the all-reduce failed")

    monkeypatch.setattr(qdist, "allreduce_max_", broken)
    with pytest.raises(RuntimeError, match="injected"):
        a(_x(99, scale=100.0))
    monkeypatch.setattr(qdist, "allreduce_max_", lambda t, world=None: t)
    for s in range(4, 7):
        ya = a(_x(s))
        monkeypatch.setattr(qdist, "exchange_active", lambda world=None: False)
        yb = b(_x(s))
        monkeypatch.setattr(qdist, "exchange_active", lambda world=None: True)
        assert same(ya, yb) and same(a.weight, b.weight) and same(a._n_updates, b._n_updates), s


@pytest.mark.parametrize("fmt", [torch.contiguous_format, torch.channels_last])
def test_sites_on_a_side_stream_equal_the_default_stream(fmt):
    """every launch goes to torch's CURRENT stream (`_hip._stream`): a training loop that runs its forward + backward under
    `torch.cuda.stream(side)` -- pipelined input staging, a second model on a stream of its own -- must see the same bits,
    and nothing may land on the default stream behind its back (checked by keeping the default stream busy with work that
    would race if a launch went there)"""
    def net():
        torch.manual_seed(0)
        m = nn.Sequential(nn.ReLU(), nn.Identity(), nn.ReLU())
        m = qs.convert(m, qs.prune(sparsity=0.5, start=2, interval=1, repetition=2), activation_layers=[nn.ReLU],
                       excluded_activation_layer_indexes=[(nn.ReLU, [1])], log=False)
        return qs.convert(m, qs.quantize(bits=4, channelwise=-1, timeout=2), activation_layers=[nn.ReLU, nn.Identity],
                          log=False).to(DEV).train()

    ref, sid = net(), net()
    side = torch.cuda.Stream()
    busy = torch.empty(64 * 1024 * 1024, device=DEV)
    for s in range(8):
        x = _x(s).contiguous(memory_format=fmt)
        ya, ga = _step(ref, x)
        torch.cuda.synchronize()
        for _ in range(4):
            busy.normal_()                                   # the default stream is occupied while the side stream works
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.default_stream())
            xs = x.clone()
            yb, gb = _step(sid, xs)
        side.synchronize()
        assert same(ya, yb) and same(ga, gb), s
        for (ka, va), (kb, vb) in zip(ref.state_dict().items(), sid.state_dict().items()):
            assert ka == kb and same(va, vb), (s, ka)
