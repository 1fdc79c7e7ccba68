"""hipGraph capture of training steps that contain prune/quantize operators.

A step of a converted network is launch-bound on the host: ~100 operator sites, each a handful of small
kernels issued from Python.  HIP graphs remove that cost, but a captured launch freezes its by-value
arguments, and the operators' running means depend on step counters.  With
``set_qsparse_options(graph_safe=True)`` every GPU layer hands those counters to its kernels through device
memory (`t_dev` arguments of the C ABI) and advances them with stream-ordered device operations, so a captured
step replays with live counters.  What stays frozen is the host-side *control flow*; a step is therefore only
capturable in steady state -- after the sparsity schedules have finished and the quantizers have switched on --
which :func:`steady_state` checks.

    qs.set_qsparse_options(graph_safe=True)
    for _ in range(warmup): train_step(static_x, static_y)          # eager, reaches steady state
    assert graphs.steady_state(model)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): train_step(static_x, static_y)        # standard whole-step capture
    for batch in loader: static_x.copy_(...); static_y.copy_(...); g.replay()
    graphs.resync_host_state(model)                                  # before going back to eager / checkpoints

`state_dict()` tensors (masks, scales, magnitudes, counters) are live device memory and stay correct under
replay; only the host mirrors need :func:`resync_host_state`.
"""
import torch
import torch.nn as nn

from qsparse_amd.quantize import QuantizeLayer
from qsparse_amd.sparse import MagnitudePruningCallback, PruneLayer, UniformPruningCallback
from qsparse_amd.util import get_option


def steady_state(model: nn.Module) -> bool:
    """True when another training step takes exactly the same host-side decisions as the last one."""
    if not get_option("graph_safe"):
        return False
    for m in model.modules():
        if isinstance(m, PruneLayer):
            if not m.initted:
                return False
            n = m._steps.read(m._n_updates)
            if m.mask.numel() != 1 and (n < m.start or n <= max(m.schedules)):
                return False
            cb = m.callback
            if isinstance(cb, UniformPruningCallback) or not isinstance(cb, MagnitudePruningCallback):
                return False
            if m.mask.numel() != 1:
                t = cb._t_host.read(cb.t)
                frozen = t > cb.stop_mask_refresh          # mask refresh has stopped for good
                every_step = cb.mask_refresh_interval == 1 and cb.stop_mask_refresh == float("inf") and t > 0
                if not (frozen or every_step) or cb.use_gradient or not cb.t.is_cuda:
                    return False
        elif isinstance(m, QuantizeLayer):
            if m.timeout > 0:
                if not m.initted or m._steps.read(m._n_updates) <= m.timeout or not m._quantized:
                    return False
                qc = m.callback
                if qc.group_num > 0 and qc.t <= qc.group_timeout + 1:
                    return False
    return True


def resync_host_state(model: nn.Module) -> nn.Module:
    """re-read every counter from device memory (one sync each); call after the last ``graph.replay()``."""
    for m in model.modules():
        if isinstance(m, PruneLayer):
            m._steps.invalidate()
            m._sparsity_host.invalidate()
            if hasattr(m.callback, "_t_host"):
                m.callback._t_host.invalidate()
        elif isinstance(m, QuantizeLayer):
            m._steps.invalidate()
            qc = m.callback
            buf = qc.__dict__.get("_t_dev")
            if buf is not None:
                qc.t = int(buf.item())
                qc.__dict__["_t_dev_value"] = qc.t
    return model
