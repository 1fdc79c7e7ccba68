"""The default-safe multi-tensor weight path (qsparse_amd/batch.py).

The reference evaluates a layer's weight operator when -- and only when -- the layer's weight is read
(qsparse/imitation.py:61-68 -> quantize.py:473-518).  `convert` now installs `WeightBatcher` by default, which evaluates all
weight quantizers at the start of the root's forward with three multi-tensor launches.  Every scenario below runs twice --
`batch_weights` on (the default) and off (layer by layer, the reference's order) -- and compares bit for bit everything the
weight path determines: every state_dict tensor (scales, counters, parameters), the quantizer callbacks' step counters and
`_quantized` flags, and the effective (quantized) weight every layer computes with at the end.  MIOpen's convolutions are
not run-to-run deterministic on this stack, so the parameters follow a seeded pseudo-gradient instead of the real one
(the real backward still runs, through the batcher's STE node, and is compared with a tolerance)."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from qsparse_amd.quantize import QuantizeLayer

pytestmark = pytest.mark.gpu
qs.set_qsparse_options(log_on_created=False, log_during_train=False)


class Branchy(nn.Module):
    """two branches, the forward takes one of them; optionally raises after the first layer; optionally reads a layer twice"""

    def __init__(self):
        super().__init__()
        self.stem = nn.Conv2d(3, 8, 3, padding=1)
        self.left = nn.Conv2d(8, 8, 3, padding=1)
        self.right = nn.Conv2d(8, 8, 3, padding=1)
        self.shared = nn.Conv2d(8, 8, 1)
        self.head = nn.Linear(8, 5)
        self.route = "left"
        self.fail = False
        self.twice = False

    def forward(self, x):
        h = F.relu(self.stem(x))
        if self.fail:
            raise RuntimeError("boom")
        h = F.relu(self.left(h) if self.route == "left" else self.right(h))
        h = self.shared(h)
        if self.twice:
            h = self.shared(F.relu(h))
        return self.head(h.mean(dim=(2, 3)))


def _state(model):
    out = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for name, m in model.named_modules():
        if isinstance(m, QuantizeLayer):
            out[name + ".<t>"] = torch.tensor(m.callback.t)
            out[name + ".<quantized>"] = torch.tensor(m._quantized)
    return out


# quantizer configurations the scenarios run with: name -> (callback kind, channelwise, bias_bits).  "default" is the
# reference's own default for quantize(): per channel along dim 1 (quantize.py:524); "bias" adds bias quantizers, which share
# their layer's callback and its running-mean count t (quantize.py:548,559-571)
KINDS = {"scaler": ("scaler", -1, -1), "decimal": ("decimal", -1, -1), "default": ("scaler", 1, -1),
         "decimal_dim0_bias": ("decimal", 0, 6), "scaler_dim1_bias": ("scaler", 1, 8)}


def _build(quantizer="scaler", timeout=2, channels_last=False):
    kind, channelwise, bias_bits = KINDS[quantizer]
    torch.manual_seed(0)
    cb = qs.DecimalQuantizer() if kind == "decimal" else None
    model = qs.convert(Branchy(), qs.quantize(bits=4, channelwise=channelwise, timeout=timeout, callback=cb, bias_bits=bias_bits),
                       weight_layers=[nn.Conv2d, nn.Linear], log=False).cuda().train()
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    return model, torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)


def _scenario(script, quantizer="scaler"):
    """script(model, step) drives the model; returns what to compare"""
    results = []
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            model, opt = _build(quantizer)
            assert (model.__dict__.get("_qs_weight_batcher") is not None) == batched
            if batched:      # every layer takes part, with its bias when that is quantized
                wb = model.__dict__["_qs_weight_batcher"]
                assert len(wb.layers) == 5 and len(wb.units) == (10 if KINDS[quantizer][2] > 0 else 5)
            g = torch.Generator().manual_seed(3)
            trace, loose = [], []

            def step(train=True, backward=True):
                x = torch.randn(4, 3, 10, 10, generator=g).cuda()
                y = torch.randint(0, 5, (4,), generator=g).cuda()
                if not train:
                    with torch.no_grad():
                        loose.append(model(x).detach().clone())
                    return
                opt.zero_grad(set_to_none=False)
                out = model(x)
                loose.append(out.detach().clone())
                if backward:
                    F.cross_entropy(out, y).backward()
                    loose.append(model.stem._parameters["weight"].grad.detach().clone())
                    with torch.no_grad():            # seeded pseudo-gradient step: identical in both runs by construction
                        for prm in model.parameters():
                            if prm.requires_grad:
                                prm.add_(torch.randn(prm.shape, generator=g).cuda() * 0.02)

            script(model, step)
            torch.cuda.synchronize()
            model.eval()
            for name, m in model.named_modules():       # the weight (and bias) every layer would compute with now
                if isinstance(getattr(m, "quantize", None), QuantizeLayer):
                    trace.append(m.weight.detach().clone())
                    if isinstance(getattr(m, "quantize_bias", None), QuantizeLayer):
                        trace.append(m.bias.detach().clone())
            results.append((trace, _state(model), loose))
        finally:
            qs.set_qsparse_options(batch_weights=True)
    (ta, sa, la), (tb, sb, lb) = results
    assert len(ta) == len(tb) and len(la) == len(lb) and sa.keys() == sb.keys()
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert torch.equal(a, b), ("effective weight", i)
    for i, (a, b) in enumerate(zip(la, lb)):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5), ("outputs / gradients", i)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    return sa


@pytest.mark.parametrize("quantizer", list(KINDS))
def test_a_branch_the_forward_skips_is_rolled_back(quantizer):
    def script(model, step):
        for i in range(9):
            model.route = "left" if i % 3 else "right"
            step()

    state = _scenario(script, quantizer)
    # the two branches really advanced differently (6 vs 3 reads, minus the 2 identity steps each saw)
    assert state["left.quantize._n_updates"].item() == 6 and state["right.quantize._n_updates"].item() == 3
    per_read = 2 if KINDS[quantizer][2] > 0 else 1       # a bias quantizer advances the shared count once more per forward
    assert state["left.quantize.<t>"].item() == 4 * per_read and state["right.quantize.<t>"].item() == 1 * per_read
    if per_read == 2:
        assert state["left.quantize_bias._n_updates"].item() == 6 and state["left.quantize_bias.<t>"].item() == 8


@pytest.mark.parametrize("quantizer", ["scaler", "default", "decimal_dim0_bias"])
def test_an_exception_in_the_forward_rolls_back_what_was_not_read(quantizer):
    def script(model, step):
        for i in range(8):
            model.fail = i in (3, 5)
            if model.fail:
                with pytest.raises(RuntimeError, match="boom"):
                    step()
                # right after the failed forward -- not at the next one -- the state is the layer-by-layer state
                if model.__dict__.get("_qs_weight_batcher") is not None:
                    assert not model.__dict__["_qs_weight_batcher"]._pending
            else:
                step()

    state = _scenario(script, quantizer)
    assert state["stem.quantize._n_updates"].item() == 8 and state["head.quantize._n_updates"].item() == 6


@pytest.mark.parametrize("quantizer", ["scaler", "default", "scaler_dim1_bias", "decimal_dim0_bias"])
def test_eval_train_switches_and_forwards_without_backward(quantizer):
    def script(model, step):
        for i in range(12):
            if i in (4, 5, 9):
                model.eval()
                step(train=False)
                step(train=False)                       # second evaluation call: quantized weights from the batcher's cache
                model.train()
            else:
                step(backward=i != 7)

    _scenario(script, quantizer)


@pytest.mark.parametrize("quantizer", ["scaler", "default", "decimal_dim0_bias"])
def test_a_layer_read_twice_and_a_weight_written_before_its_read(quantizer):
    def script(model, step):
        model.twice = True
        bump = []

        def touch_left(m, args, out):                    # the stem's forward hook writes `left`'s weight in place, i.e. AFTER the
            if bump:                                     # root's pre-hook precomputed it and before `left` reads it
                with torch.no_grad():
                    model.left._parameters["weight"].mul_(1.01)

        def touch_shared(m, args):                       # a pre-hook on the layer itself, writing through .data: not batched
            if bump:
                m._parameters["weight"].data.mul_(0.99)

        handles = [model.stem.register_forward_hook(touch_left), model.shared.register_forward_pre_hook(touch_shared)]
        for i in range(7):
            if i == 4:
                bump.append(1)                           # from now on both weights change between precomputation and read
            step()
        for h in handles:
            h.remove()
        step()

    state = _scenario(script, quantizer)
    assert state["shared.quantize._n_updates"].item() == 16       # two reads per forward, both counted


def test_hooked_quantizers_keep_their_inline_path_and_copies_are_independent():
    def script(model, step):
        seen = []
        model.right.quantize.register_forward_hook(lambda m, a, o: seen.append(1))
        model.route = "right"
        for _ in range(4):
            step()
        assert len(seen) == 4                             # the hook saw every call, batcher or not
        twin = copy.deepcopy(model)                       # a deep copy carries its own batcher (or none): training it ...
        twin.route = "left"
        x = torch.ones(2, 3, 10, 10).cuda()
        for _ in range(3):
            twin(x).sum().backward()
        assert twin.left.quantize._n_updates.item() == 3 and model.left.quantize.initted is False   # ... leaves the original alone
        step()

    _scenario(script)


def _ddp_worker(rank, world, port, batched, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)        # both ranks share cuda:0 on a 1-GPU box
    torch.cuda.set_device(0)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False, batch_weights=batched)
    model, _ = _build()
    net = nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
    g = torch.Generator().manual_seed(10 + rank)
    for i in range(6):
        model.route = "left" if i % 2 else "right"
        x, y = torch.randn(4, 3, 10, 10, generator=g).cuda(), torch.randint(0, 5, (4,), generator=g).cuda()
        net.zero_grad(set_to_none=False)
        F.cross_entropy(net(x), y).backward()
        gp = torch.Generator().manual_seed(100 + i)      # the same pseudo-gradient on every rank (DDP's all-reduce ran above)
        with torch.no_grad():
            for prm in model.parameters():
                if prm.requires_grad:
                    prm.add_(torch.randn(prm.shape, generator=gp).cuda() * 0.02)
    torch.cuda.synchronize()
    out[(batched, rank)] = {k: v.cpu() for k, v in _state(model).items()}
    dist.destroy_process_group()


def test_under_ddp_on_two_ranks():
    """DistributedDataParallel over gloo, two ranks on the shared GPU: weights are identical on every rank, so the batched
    statistics are too; both ranks and both modes end in the same state"""
    import socket
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    for batched in (True, False):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_ddp_worker, args=(2, port, batched, out), nprocs=2, join=True)
    ref = out[(False, 0)]
    for key in ((False, 1), (True, 0), (True, 1)):
        assert out[key].keys() == ref.keys()
        for k in ref:
            assert torch.equal(out[key][k], ref[k]), (key, k)


def test_deep_copy_of_a_converted_network_with_warm_caches():
    """site plans (fused.py) and launch plans (batch.py) cache raw device pointers in ctypes objects; a deep copy of the
    network must neither fail on them nor share them: the twin trains on, bit-identically to the original"""
    from examples.models import convert_pq, resnet18
    torch.manual_seed(0)
    model = convert_pq(resnet18(10, True, 8), sparsity=0.5, bits=4, prune_start=1, prune_interval=1, repetition=1,
                       quant_timeout=1).cuda().train()
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(8, 3, 32, 32, generator=g).cuda() for _ in range(6)]
    for x in xs[:4]:
        model(x).sum().backward()                       # caches are warm now
    twin = copy.deepcopy(model)
    with torch.no_grad():                               # (identical parameters from here on: no optimizer, grads unused)
        outs = [(model(x), twin(x)) for x in xs[4:]]
    torch.cuda.synchronize()
    sa, sb = _state(model), _state(twin)
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for a, b in outs:
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4)
    assert twin.__dict__["_qs_weight_batcher"] is not model.__dict__["_qs_weight_batcher"]


def test_a_layer_the_forward_never_reads_receives_no_gradient():
    """`.grad` of a skipped layer's weight stays None (zero_grad(set_to_none=True) world): an optimizer with weight decay must
    not start moving a parameter the forward did not use -- with the grouped hand-out node as with the layer-by-layer path"""
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            model, _ = _build()
            model.route = "left"
            opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=0.1)
            before = model.right._parameters["weight"].detach().clone()
            for _ in range(5):
                opt.zero_grad(set_to_none=True)
                x = torch.randn(4, 3, 10, 10, device="cuda")
                model(x).sum().backward()
                assert model.right._parameters["weight"].grad is None and model.left._parameters["weight"].grad is not None
                opt.step()
            assert torch.equal(model.right._parameters["weight"].detach(), before)
            assert not model.right.quantize.initted
        finally:
            qs.set_qsparse_options(batch_weights=True)
    # once `right` has been read (and is batched from then on), skipping it again still leaves its gradient alone
    model, _ = _build()
    for i in range(8):
        model.route = "right" if i < 4 else "left"
        model.zero_grad(set_to_none=True)
        model(torch.randn(4, 3, 10, 10, device="cuda")).sum().backward()
        if i >= 4:
            assert model.right._parameters["weight"].grad is None
    assert model.right.quantize._n_updates.item() == 4


class _Probe(nn.Linear):
    """records whether the weight its forward computes with requires grad"""

    def forward(self, x):
        w = self.weight
        self.__dict__["seen"] = w.requires_grad
        return F.linear(x, w, self.bias)


def test_a_frozen_weight_hands_out_a_weight_that_does_not_require_grad():
    """fine-tuning on a frozen backbone: the quantized weight of a `requires_grad=False` parameter does not require grad -- as
    layer by layer -- so the layer's backward skips the weight-gradient pass; its neighbours in the hand-out node train on"""
    res = []
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            torch.manual_seed(0)
            net = nn.Sequential(_Probe(8, 8), nn.Tanh(), _Probe(8, 8), nn.Tanh(), _Probe(8, 4))
            net = qs.convert(net, qs.quantize(bits=4, channelwise=-1, timeout=1), weight_layers=[_Probe], log=False).cuda().train()
            assert (net.__dict__.get("_qs_weight_batcher") is not None) == batched
            net[0]._parameters["weight"].requires_grad_(False)
            net[2]._parameters["weight"].requires_grad_(False)
            g = torch.Generator().manual_seed(5)
            for _ in range(4):
                net.zero_grad(set_to_none=True)
                x = torch.randn(6, 8, generator=g).cuda().requires_grad_()
                net(x).square().sum().backward()
                assert [net[i].__dict__["seen"] for i in (0, 2, 4)] == [False, False, True]
                assert net[0]._parameters["weight"].grad is None and net[2]._parameters["weight"].grad is None
            res.append((x.grad.clone(), net[4]._parameters["weight"].grad.clone(), net[0].bias.grad.clone(),
                        {k: v.detach().clone() for k, v in net.state_dict().items()}))
        finally:
            qs.set_qsparse_options(batch_weights=True)
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k


class BiasFirst(nn.Module):
    """a layer whose forward reads the bias BEFORE the weight (a fused bias-activation kernel would): the reference then updates
    the bias statistics with the callback's count t and the weight's with t + 1 -- the other way round from nn.Conv2d"""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 6)
        self.b = nn.Linear(6, 4)
        self.bias_first = True

    def forward(self, x):
        h = torch.relu(self.a(x))
        if self.bias_first:
            bias = self.b.bias
            return F.linear(h, self.b.weight) + bias
        return self.b(h)


@pytest.mark.parametrize("kind", ["scaler", "decimal"])
def test_a_bias_read_before_its_weight_follows_the_order_of_the_reads(kind):
    results = []
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            torch.manual_seed(0)
            cb = qs.DecimalQuantizer() if kind == "decimal" else None
            model = qs.convert(BiasFirst(), qs.quantize(bits=6, channelwise=0, timeout=1, bias_bits=6, callback=cb),
                               weight_layers=[nn.Linear], log=False).cuda().train()
            g = torch.Generator().manual_seed(5)
            outs = []
            for i in range(8):
                model.bias_first = i not in (5, 6)
                x = torch.randn(3, 6, generator=g).cuda()
                y = model(x)
                y.sum().backward()
                with torch.no_grad():
                    for prm in model.parameters():
                        if prm.requires_grad:
                            prm.add_(torch.randn(prm.shape, generator=g).cuda() * 0.05)
                            prm.grad = None
                outs.append(y.detach().clone())
            results.append((outs, _state(model)))
        finally:
            qs.set_qsparse_options(batch_weights=True)
    (oa, sa), (ob, sb) = results
    for a, b in zip(oa, ob):
        assert torch.equal(a, b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("channels_last", [False, True])
def test_reference_default_quantize_on_every_resnet50_weight_vs_the_oracle_in_a_handful_of_launches(channels_last, monkeypatch):
    """a user who writes `convert(model, quantize(bits=8), weight_layers=[...])` gets the reference's DEFAULT,
    a per-channel quantizer along dim 1 (quantize.py:524).  All 54 weights of a ResNet-50 take the multi-tensor path: three
    launches per forward (any number of tensors: the descriptor table lives on the device) and one per eight layers in the
    backward; every running scale follows the oracle's QuantizeSim fed with the raw parameter of that step, the effective
    weights equal its output, contiguous and channels_last (memory view [Cout*kh*kw, Cin, 1]) alike."""
    from examples.models import resnet50
    from oracle import qs_oracle as O
    from qsparse_amd import _hip
    torch.manual_seed(0)
    model = qs.convert(resnet50(10, False, width=16), qs.quantize(bits=8, timeout=1), weight_layers=[nn.Conv2d, nn.Linear],
                       log=False).cuda().train()
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    wb = model.__dict__["_qs_weight_batcher"]
    layers = {name: m for name, m in model.named_modules() if isinstance(getattr(m, "quantize", None), QuantizeLayer)}
    assert len(layers) == 54 and len(wb.units) == 54 and all(m.quantize.channelwise == 1 for m in layers.values())
    sims = {name: O.QuantizeSim("scaler", 8, 1, 1, batch_dimension=-1) for name in layers}
    calls = []
    for fn in ("multi_absmax", "multi_scale_update", "multi_quant_fwd", "multi_ste_bwd", "absmax", "scale_update", "quant_fwd", "ste_bwd"):
        real = getattr(_hip, fn)
        monkeypatch.setattr(_hip, fn, (lambda real, fn: (lambda *a, **k: (calls.append(fn), real(*a, **k))[1]))(real, fn))
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    g = torch.Generator().manual_seed(2)
    for step in range(5):
        raw = {name: m._parameters["weight"].detach().cpu().contiguous().clone() for name, m in layers.items()}
        x = torch.randn(4, 3, 64, 64, generator=g).cuda()
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        del calls[:]
        opt.zero_grad()
        model(x).square().mean().backward()
        if step >= 1:       # (step 0 is the identity step of timeout=1: nothing to launch)
            assert calls.count("multi_absmax") == 1 and calls.count("multi_scale_update") == 1 and calls.count("multi_quant_fwd") == 1
            assert calls.count("multi_ste_bwd") == 7 and not {"absmax", "scale_update", "quant_fwd", "ste_bwd"} & set(calls), calls
        opt.step()
        for name, m in layers.items():
            sims[name].step(raw[name], True)
            if sims[name].weight is not None and m.quantize.initted:
                assert torch.equal(m.quantize.weight.detach().cpu(), sims[name].weight), (step, name)
            assert m.quantize._n_updates.item() == sims[name].n_updates and m.quantize.callback.t == sims[name].shared["t"]
    model.eval()
    with torch.no_grad():
        model(x)                                   # evaluation hand-out (batched)
        for name, m in layers.items():
            w = m._parameters["weight"].detach().cpu().contiguous()
            assert torch.equal(m.weight.detach().cpu().contiguous(), sims[name].apply(w, False)), name


@pytest.mark.parametrize("quantizer", ["scaler", "decimal_dim0_bias", "default"])
def test_the_steady_state_fast_path_of_the_weight_path_equals_the_full_one(quantizer, monkeypatch):
    """once every tensor is quantized on every read the precomputation only compares identities and re-issues the cached launch
    table (`_Steady`): a run with it and a twin without (QS_NO_FAST_PATH) through route switches (roll-backs), an evaluation
    step, a hook that comes and goes, a configuration attribute and a parameter re-assigned -- bit for bit; and it did engage"""
    results, armed = [], []
    for fast in (True, False):
        if not fast:
            monkeypatch.setenv("QS_NO_FAST_PATH", "1")
        model, opt = _build(quantizer)
        wb = model.__dict__["_qs_weight_batcher"]
        g = torch.Generator().manual_seed(3)
        trace = []
        handle = None
        for i in range(20):
            model.route = "left" if i % 3 else "right"
            if i == 8:
                handle = model.left.quantize.register_forward_hook(lambda m, a, o: None)
            if i == 10:
                handle.remove()
            if i == 12:
                model.shared.quantize.bits = 6
            if i == 15:
                with torch.no_grad():
                    model.head._parameters["weight"] = nn.Parameter(model.head._parameters["weight"].detach().clone() * 1.5)
            x = torch.randn(4, 3, 10, 10, generator=g).cuda()
            if i == 6:
                model.eval()
                with torch.no_grad():
                    trace.append(model(x).detach().clone())
                model.train()
                continue
            if fast:
                armed.append((i, wb._steady is not None))
            for prm in model.parameters():
                prm.grad = None
            out = model(x)
            out.sum().backward()
            trace.append(out.detach().clone())
            with torch.no_grad():
                for prm in model.parameters():
                    if prm.requires_grad:
                        prm.add_(torch.randn(prm.shape, generator=g).cuda() * 0.02)
        model.eval()
        trace += [m.weight.detach().clone() for m in (model.stem, model.left, model.right, model.shared, model.head)]
        results.append((trace, _state(model)))
    (ta, sa), (tb, sb) = results
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for i, (a, b) in enumerate(zip(ta[-5:], tb[-5:])):
        assert torch.equal(a, b), ("effective weight", i)
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5), ("output", i)
    armed = dict(armed)
    # armed in the plain stretches; not on the first steps (timeouts: the skipped branch's layer sits AT its timeout for a
    # while), not after the evaluation step, not while the hook is there; a changed attribute / a re-assigned parameter is
    # noticed by the step that meets it, whose full path re-arms
    assert armed[5] and not armed[7] and not armed[9] and not armed[10] and all(armed[i] for i in (11, 13, 14, 16, 17, 18, 19)), \
        "".join("1" if v else "0" for _, v in sorted(armed.items()))
