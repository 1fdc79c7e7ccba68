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


def _build(quantizer="scaler", timeout=2):
    torch.manual_seed(0)
    cb = qs.DecimalQuantizer() if quantizer == "decimal" else None
    model = qs.convert(Branchy(), qs.quantize(bits=4, channelwise=-1, timeout=timeout, callback=cb),
                       weight_layers=[nn.Conv2d, nn.Linear], log=False).cuda().train()
    return model, torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)


def _scenario(script, quantizer="scaler"):
    """script(model, step) drives the model; returns what to compare"""
    results = []
    for batched in (True, False):
        qs.set_qsparse_options(batch_weights=batched)
        try:
            model, opt = _build(quantizer)
            assert (model.__dict__.get("_qs_weight_batcher") is not None) == batched
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
            for name, m in model.named_modules():       # the weight every layer would compute with now
                if isinstance(getattr(m, "quantize", None), QuantizeLayer):
                    trace.append(m.weight.detach().clone())
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


@pytest.mark.parametrize("quantizer", ["scaler", "decimal"])
def test_a_branch_the_forward_skips_is_rolled_back(quantizer):
    def script(model, step):
        for i in range(9):
            model.route = "left" if i % 3 else "right"
            step()

    state = _scenario(script, quantizer)
    # the two branches really advanced differently (6 vs 3 reads, minus the 2 identity steps each saw)
    assert state["left.quantize._n_updates"].item() == 6 and state["right.quantize._n_updates"].item() == 3
    assert state["left.quantize.<t>"].item() == 4 and state["right.quantize.<t>"].item() == 1


def test_an_exception_in_the_forward_rolls_back_what_was_not_read():
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

    state = _scenario(script)
    assert state["stem.quantize._n_updates"].item() == 8 and state["head.quantize._n_updates"].item() == 6


def test_eval_train_switches_and_forwards_without_backward():
    def script(model, step):
        for i in range(12):
            if i in (4, 5, 9):
                model.eval()
                step(train=False)
                step(train=False)                       # second evaluation call: quantized weights from the batcher's cache
                model.train()
            else:
                step(backward=i != 7)

    _scenario(script)


def test_a_layer_read_twice_and_a_weight_written_before_its_read():
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

    state = _scenario(script)
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
