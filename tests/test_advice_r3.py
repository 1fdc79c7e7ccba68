"""Regression tests for the round-3 advisor findings: each one reproduces the reported behaviour and pins the fix.

  * shared quantizer callbacks (reference quantize.py:548,559-571 allows one callback object on several layers) must not
    make a later `convert` fail -- such layers keep the inline weight path;
  * `export_integer` must not hand out codes for a layer that evaluates unquantized (`_quantized` False after a checkpoint
    load, quirk B7);
  * the batcher's evaluation cache must not serve stale weights to a loop that writes parameters through `.data`;
  * `QuantizeLayer.single_call_step` must leave layers alone whose callback carries backward hooks or when global module
    hooks are installed.
"""
import copy

import pytest
import torch
import torch.nn as nn

import qsparse_amd as qs
from qsparse_amd.batch import WeightBatcher, _batchable
from qsparse_amd.quantize import ScalerQuantizer

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _shared_net():
    torch.manual_seed(0)
    cb = ScalerQuantizer()
    return nn.Sequential(qs.quantize(nn.Conv2d(3, 8, 3, padding=1), bits=8, channelwise=-1, timeout=1, callback=cb), nn.ReLU(),
                         qs.quantize(nn.Conv2d(8, 8, 3, padding=1), bits=8, channelwise=-1, timeout=1, callback=cb), nn.ReLU(),
                         qs.quantize(nn.Conv2d(8, 4, 3, padding=1), bits=8, channelwise=-1, timeout=1))


@pytest.mark.parametrize("dev", DEVICES)
def test_convert_accepts_layers_that_share_one_quantizer_callback(dev):
    ref = _shared_net().to(dev)
    net = copy.deepcopy(ref)
    assert net[0].quantize.callback is net[2].quantize.callback
    # the reported failure: any later convert() installed a WeightBatcher whose constructor raised ValueError
    net = qs.convert(net, qs.prune(sparsity=0.5, start=2, interval=1, repetition=1), activation_layers=[nn.ReLU], log=False)
    layers = _batchable(net)
    assert [type(l).__name__ for l in layers] == ["Conv2d"] and layers[0] is net[4]       # only the layer that owns its callback
    WeightBatcher(net).remove()                                                            # by hand as well
    ref = qs.convert(ref, qs.prune(sparsity=0.5, start=2, interval=1, repetition=1), activation_layers=[nn.ReLU], log=False,
                     batch_weights=False)
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(1)).to(dev)
    for _ in range(4):          # batcher installed (shared layers inline) == no batcher at all, state for state
        a, b = net(x), ref(x)
        assert torch.equal(a, b)
    assert net[0].quantize.callback.t == ref[0].quantize.callback.t == 6       # the shared callback advanced once per read
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


@pytest.mark.parametrize("dev", DEVICES)
def test_export_skips_a_layer_that_evaluates_unquantized_after_a_plain_checkpoint_load(dev):
    torch.manual_seed(2)

    def build():
        return qs.quantize(nn.Linear(6, 5), bits=8, channelwise=-1, timeout=1).to(dev)

    layer = build()
    x = torch.randn(3, 6).to(dev)
    for _ in range(3):
        layer(x)
    ckpt, extra = copy.deepcopy(layer.state_dict()), qs.extra_state_dict(layer)
    assert len(qs.export_integer(layer)) == 1
    fresh = build()
    qs.preload_qsparse_state_dict(fresh, ckpt)
    fresh.load_state_dict(ckpt)
    fresh.eval()
    # quirk B7: counters say "quantizing", `_quantized` is False -> the eval forward uses the raw weight ...
    assert not fresh.quantize._quantized
    assert torch.equal(fresh.weight, fresh._parameters["weight"])
    # ... so there is nothing to export (before the fix: codes whose dequantize() != the effective weight)
    assert qs.export_integer(fresh) == {}
    qs.load_extra_state_dict(fresh, extra)
    ex = qs.export_integer(fresh)
    (rec,) = ex.values()
    assert torch.equal(rec.weight.dequantize(), fresh.weight)


@pytest.mark.gpu
def test_eval_cache_is_not_used_with_gradients_enabled_and_resync_invalidates_it():
    torch.manual_seed(3)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 4, 3, padding=1))
    net = qs.convert(net, qs.quantize(bits=8, channelwise=-1, timeout=1), weight_layers=[nn.Conv2d], log=False).cuda()
    wb = net.__dict__["_qs_weight_batcher"]
    x = torch.randn(2, 3, 8, 8, device="cuda")
    for _ in range(3):
        net(x).sum().backward()
    net.eval()
    # fine-tuning style loop under eval() WITH gradients: parameters written through `.data` must be seen on the next forward
    y0 = net(x)
    assert wb._eval_key is None
    net[0]._parameters["weight"].data.mul_(0.5)
    y1 = net(x)
    assert not torch.equal(y0, y1)
    inline = copy.deepcopy(net)
    inline.__dict__["_qs_weight_batcher"].remove()
    assert torch.equal(inline(x), y1)
    # serving (no_grad): the cache is used; a `.data` write is invisible to it by design, resync_host_state drops it
    with torch.no_grad():
        s0 = net(x)
        assert wb._eval_key is not None and torch.equal(net(x), s0)
        net[0]._parameters["weight"].data.mul_(2.0)
        qs.resync_host_state(net)
        assert wb._eval_key is None
        s1 = net(x)
        assert torch.equal(s1, y0) and not torch.equal(s1, s0)


@pytest.mark.gpu
def test_single_call_step_leaves_layers_with_backward_or_global_hooks_to_the_protocol_route():
    x = torch.randn(4, 8, 6, 6, device="cuda")

    def layer():
        q = qs.quantize(bits=8, channelwise=-1, timeout=1).cuda()
        q(x), q(x)
        return q

    q = layer()
    assert q.single_call_step(x, 2) is not None
    seen = []
    q = layer()
    q.callback.register_full_backward_hook(lambda m, gi, go: seen.append("bwd"))
    assert q.single_call_step(x, 2) is None
    xg = x.clone().requires_grad_(True)
    q(xg).sum().backward()
    assert seen == ["bwd"]
    q = layer()
    handle = nn.modules.module.register_module_forward_hook(lambda m, a, o: seen.append(type(m).__name__) if m is q.callback else None)
    try:
        assert q.single_call_step(x, 2) is None
        q(x)
    finally:
        handle.remove()
    assert seen[-1] == "ScalerQuantizer"
    assert q.single_call_step(x, 3) is not None
