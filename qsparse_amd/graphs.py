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
    from qsparse_amd import distributed as qdist
    if qdist.exchange_active():
        # a live statistics exchange puts an RCCL collective inside every site's step; capturing one into a hipGraph
        # crashes on this stack (torch 2.10 + ROCm 7.2, one-rank group: SIGSEGV during capture,
        # tests/test_distributed.py) -- data-parallel steps therefore always run eagerly
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
            if cb.l0 and m.mask.numel() != 1 and not (cb._t_host.read(cb.t) > cb.stop_mask_refresh):
                return False       # the L0 choice of 2-byte inputs is taken on the host each step (sparse._importance)
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
        batcher = m.__dict__.get("_qs_weight_batcher")
        if batcher is not None:
            batcher.invalidate()       # quantized weights kept for evaluation: whoever asks for a resync wrote behind our back
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


class GraphedStep:
    """runs ``step_fn(*tensors)`` eagerly until the network is in steady state, then captures it once into a
    hipGraph and replays it from there on:

        step = graphs.GraphedStep(model, train_step)      # train_step(x, y) -> loss (fwd + bwd + optimizer)
        for x, y in loader:
            loss = step(x, y)                              # eager first, graph replay later; same results
        step.finish()                                      # host mirrors back in sync (checkpoints, eval)

    Inputs are copied into static buffers (shapes and dtypes must not change once captured); tensors returned by
    ``step_fn`` are static too and are overwritten by the next call.  ``settle`` extra eager steps are run in
    steady state before capturing so that allocations (optimizer state, layer scratch) exist.  Sets the
    ``graph_safe`` option (required for the counters to stay live under replay)."""

    def __init__(self, model: nn.Module, step_fn, settle: int = 2):
        from qsparse_amd.util import set_options

        set_options(graph_safe=True)
        self.model, self.step_fn, self.settle = model, step_fn, settle
        self.graph = None
        self._steady_steps = 0
        self._static_in = None
        self._static_out = None

    @property
    def captured(self) -> bool:
        return self.graph is not None

    def __call__(self, *tensors):
        if self.graph is not None:
            for s, t in zip(self._static_in, tensors):
                s.copy_(t, non_blocking=True)
            self.graph.replay()
            return self._static_out
        if not (tensors and all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors)) or not steady_state(self.model):
            return self.step_fn(*tensors)
        if self._steady_steps < self.settle:
            self._steady_steps += 1
            return self.step_fn(*tensors)
        self._static_in = [t.detach().clone() for t in tensors]
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._static_out = self.step_fn(*self._static_in)
        self.graph = graph
        graph.replay()                     # the capture records the step without running it
        return self._static_out

    def finish(self) -> nn.Module:
        """drop the graph and re-read the host mirrors of every counter."""
        self.graph = None
        self._static_in = self._static_out = None
        self._steady_steps = 0
        return resync_host_state(self.model)
