"""Misuse scenarios for the error-behaviour fixture F18: each takes a namespace `ns` (the reference's `qsparse` when the fixture is
generated, `qsparse_amd` in the tests) and a device, and either raises or returns.  The fixture records, per scenario, what the
REFERENCE did -- ("ok",) or ("raised", exception type name, first line of the message) -- and tests/test_error_behaviour.py holds
the package to the same outcome on the CPU and on the GPU.  Only ARGUMENT errors live here (they fail for a contiguous input too);
layouts the reference cannot run at all (channels_last, permuted) are a documented superset and are not scenarios."""
import torch
import torch.nn as nn


def _steps(layer, x, n):
    for _ in range(n):
        y = layer(x)
    return y


def prune_1d_input(ns, dev):
    ns.prune(sparsity=0.5, start=0, interval=1, repetition=1).to(dev).train()(torch.randn(8, device=dev))


def prune_eval_full_mask_other_shape(ns, dev):
    p = ns.prune(sparsity=0.5, dimensions={0, 1, 2, 3}, start=0, interval=1, repetition=1).to(dev).train()
    _steps(p, torch.randn(2, 4, 6, 6, device=dev), 3)
    p.eval()(torch.randn(2, 4, 5, 5, device=dev))


def prune_eval_channel_mask_other_spatial(ns, dev):
    p = ns.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1).to(dev).train()
    _steps(p, torch.randn(2, 4, 6, 6, device=dev), 3)
    p.eval()(torch.randn(3, 4, 5, 7, device=dev))


def prune_train_other_channel_count(ns, dev):
    p = ns.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1).to(dev).train()
    _steps(p, torch.randn(2, 4, 6, 6, device=dev), 3)
    p(torch.randn(2, 8, 6, 6, device=dev))


def prune_train_other_rank(ns, dev):
    p = ns.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1).to(dev).train()
    _steps(p, torch.randn(2, 4, 6, 6, device=dev), 3)
    p(torch.randn(2, 4, 6, device=dev))


def quantize_batched_channelwise_scaler(ns, dev):
    q = ns.quantize(bits=8, timeout=1, channelwise=1).to(dev).train()
    _steps(q, torch.randn(4, 6, 5, 5, device=dev), 3)


def quantize_batched_channelwise_decimal(ns, dev):
    q = ns.quantize(bits=8, timeout=1, channelwise=1, callback=ns.DecimalQuantizer()).to(dev).train()
    _steps(q, torch.randn(4, 6, 5, 5, device=dev), 3)


def quantize_batch_of_one_channelwise_scaler(ns, dev):
    q = ns.quantize(bits=8, timeout=1, channelwise=1).to(dev).train()
    _steps(q, torch.randn(1, 6, 5, 5, device=dev), 3)


def adaptive_batched_last_dim(ns, dev):
    q = ns.quantize(bits=4, timeout=1, channelwise=2, callback=ns.AdaptiveQuantizer()).to(dev).train()
    _steps(q, torch.randn(4, 6, 8, device=dev), 3)


def adaptive_batched_dim_one(ns, dev):
    q = ns.quantize(bits=4, timeout=1, channelwise=1, callback=ns.AdaptiveQuantizer()).to(dev).train()
    _steps(q, torch.randn(4, 6, 8, device=dev), 3)


def adaptive_batch_of_one_last_dim(ns, dev):
    q = ns.quantize(bits=4, timeout=1, channelwise=2, callback=ns.AdaptiveQuantizer()).to(dev).train()
    _steps(q, torch.randn(1, 6, 8, device=dev), 3)


def quantize_invalid_argument(ns, dev):
    ns.quantize(3)


def prune_invalid_argument(ns, dev):
    ns.prune("not a module")


def callback_unsupported_combination(ns, dev):
    ns.MagnitudePruningCallback(use_gradient=True, running_average=False)


def mask_from_importance_rank_out_of_range(ns, dev):
    ns.calculate_mask_given_importance(torch.rand(4, device=dev), 1.0)


def mask_from_importance_full_range(ns, dev):
    ns.calculate_mask_given_importance(torch.rand(16, device=dev), 0.9)


def squeeze_target_not_broadcastable(ns, dev):
    from importlib import import_module
    util = import_module(ns.__name__ + ".util")
    util.squeeze_tensor_to_shape(torch.randn(2, 4, 6, device=dev), [1, 3, 1])


def squeeze_target_other_rank(ns, dev):
    from importlib import import_module
    util = import_module(ns.__name__ + ".util")
    util.squeeze_tensor_to_shape(torch.randn(2, 4, 6, device=dev), [1, 4])


def scaler_with_wrong_channel_count(ns, dev):
    from importlib import import_module
    Q = import_module(ns.__name__ + ".quantize")
    Q.quantize_with_scaler(torch.randn(2, 4, 6, device=dev), 8, torch.full((3, 1), 0.1, device=dev), 1)


def decimal_with_wrong_channel_count(ns, dev):
    from importlib import import_module
    Q = import_module(ns.__name__ + ".quantize")
    Q.quantize_with_decimal(torch.randn(2, 4, 6, device=dev), 8, torch.full((5, 1), 3.0, device=dev), 1)


def shared_activation_module_other_channel_count(ns, dev):
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.relu = nn.Conv2d(3, 4, 1), nn.Conv2d(4, 8, 1), nn.ReLU()

        def forward(self, x):
            return self.relu(self.b(self.relu(self.a(x))))

    net = ns.convert(Net(), ns.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1), activation_layers=[nn.ReLU], log=False).to(dev).train()
    _steps(net, torch.randn(2, 3, 5, 5, device=dev), 3)


def timeout_zero_never_quantizes(ns, dev):
    q = ns.quantize(bits=2, timeout=0).to(dev).train()
    x = torch.randn(2, 4, 3, 3, device=dev)
    assert torch.equal(_steps(q, x, 4), x)


def prune_weight_of_a_module_without_weight(ns, dev):
    m = ns.prune(nn.ReLU(), sparsity=0.5).to(dev)
    m(torch.randn(2, 3, device=dev))
    m.weight


def quantize_bias_of_a_layer_without_bias(ns, dev):
    m = ns.quantize(nn.Conv2d(3, 4, 1, bias=False), bits=8, bias_bits=8, timeout=1).to(dev).train()
    _steps(m, torch.randn(2, 3, 5, 5, device=dev), 3)


def layerwise_schedule_integer_interval_overflows(ns, dev):
    net = nn.Sequential(nn.Conv2d(3, 8, 1), nn.ReLU(), nn.Conv2d(8, 8, 1), nn.ReLU())
    net = ns.convert(net, ns.prune(sparsity=0.5, dimensions={1}), activation_layers=[nn.ReLU], log=False)
    net = ns.devise_layerwise_pruning_schedule(net, start=1, interval=2, mask_refresh_interval=1).to(dev).train()
    _steps(net, torch.randn(2, 3, 5, 5, device=dev), 12)


SCENARIOS = [v for k, v in list(globals().items()) if callable(v) and getattr(v, "__module__", None) == __name__ and not k.startswith("_")]
