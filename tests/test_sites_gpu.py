"""BASELINE configs 3 and 4 at their REAL width: operator sites of full-width ResNet-18 (CIFAR shape, 50 % channel
pruning) and ResNet-50 (ImageNet shape 224x224, 75 %, bf16 autocast), converted with the reference's --pq recipe
(reference qsparse/convert.py:199-229 builds the sites; sparse.py:215-273 and quantize.py:473-518 run them), checked
against the ORACLE while the network trains on the GPU.

Every selected site gets a forward hook and a full backward hook.  The hooks copy the tensors the site actually
received -- activations produced by MIOpen convolutions, batch norm, residual adds and the optimizer's weight updates,
in whatever dtype autocast gave them -- to the CPU and replay them through `oracle.PruneSim` / `oracle.QuantizeSim`,
asserting bit-for-bit equality of the site's output, input gradient, mask, running magnitude, scale and counters at
EVERY step of the schedule (inactive -> pruning starts -> quantization starts -> ramp -> steady state).  Nothing here
depends on the convolutions being deterministic: each site is compared with the oracle on the inputs it really saw.

Sites: the stem (64 x 112 x 112), the headline-shape site (256 x 56 x 56 behind a residual add), 28x28 / 14x14 / 7x7
maps (ragged rows: 196 and 49 elements), the quantize-only last ReLU, the input quantizer, a convolution weight and
the classifier weight.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50
from golden_io import same
from oracle import qs_oracle as O
from qsparse_amd.quantize import QuantizeLayer
from qsparse_amd.sparse import PruneLayer

qs.set_qsparse_options(log_on_created=False, log_during_train=False)

SCHEDULE = dict(prune_start=2, prune_interval=2, repetition=2, quant_timeout=3)


class SiteChecker:
    """oracle twin of one operator site; `fwd` / `bwd` are the hook bodies"""

    def __init__(self, name, module, sparsity, bits, log, schedule=None):
        schedule = schedule or SCHEDULE
        self.name, self.module, self.log = name, module, log
        self.relu = False
        self.p = self.q = None
        if isinstance(module, QuantizeLayer):                       # input / weight quantizer
            self.q = module
        else:
            first, second = module[0], module[1]
            if isinstance(second, QuantizeLayer) and isinstance(first, nn.Sequential):     # Sequential(Sequential(act, P), Q)
                self.relu, self.p, self.q = isinstance(first[0], nn.ReLU), first[1], second
            elif isinstance(second, QuantizeLayer):                                        # Sequential(act, Q)
                self.relu, self.q = isinstance(first, nn.ReLU), second
            else:
                raise AssertionError(f"unexpected site structure at {name}: {module}")
        self.psim = None
        if self.p is not None:
            assert isinstance(self.p, PruneLayer) and self.p.dimensions == {1}
            self.psim = O.PruneSim(sparsity, [1], schedule["prune_start"], schedule["prune_interval"], schedule["repetition"], False)
        self.qsim = O.QuantizeSim("scaler", bits, -1, schedule["quant_timeout"], batch_dimension=self.q.batch_dimension)
        self.steps = 0
        self.saved = None

    def fwd(self, module, inputs, output):
        if not module.training:
            return
        x = inputs[0].detach().cpu()                     # strides preserved: a channels_last activation stays channels_last
        h = torch.relu(x) if self.relu else x
        n_before = self.psim.n_updates if self.psim else 0
        if self.psim:
            h = self.psim.step(h, True)
        # the reference's tensor-wise abs-max calls .view(1, -1), which a channels_last tensor refuses (quantize.py:333);
        # the maximum is order-independent, so the quantizer twin is fed the same values in NCHW order
        y_ref = self.qsim.step(h.contiguous(), True)
        tag = (self.name, self.steps)
        y = output.detach().cpu()
        assert y.dtype == y_ref.dtype and same(y.contiguous(), y_ref.contiguous()), ("output", tag)
        if self.psim:
            assert same(self.p.mask.detach().cpu(), self.psim.mask), ("mask", tag)
            assert self.p._n_updates.item() == self.psim.n_updates and self.p.callback.t.item() == self.psim.t, ("counters", tag)
            assert abs(self.p._cur_sparsity.item() - self.psim.cur_sparsity) < 1e-7, ("sparsity", tag)
            if self.psim.magnitude is not None:
                assert same(self.p.callback.magnitude.detach().cpu(), self.psim.magnitude), ("magnitude", tag)
        assert same(self.q.weight.detach().cpu(), self.qsim.weight), ("scale", tag)
        assert self.q._n_updates.item() == self.qsim.n_updates, ("quantizer counter", tag)
        self.saved = (x, n_before)
        self.log.append((self.name, self.steps, tuple(x.shape), str(x.dtype), bool(self.qsim.quantized),
                         None if self.psim is None else float(self.psim.mask.float().mean())))
        self.steps += 1

    def bwd(self, module, grad_input, grad_output):
        if self.saved is None or grad_input[0] is None:
            return
        x, n_before = self.saved
        g = grad_output[0].detach().cpu().contiguous()
        gin = self.qsim.grad(g, x.dtype if (self.psim or self.relu or self.qsim.quantized) else g.dtype)
        if self.psim:
            gin = self.psim.grad(gin, n_before >= self.psim.start)
        if self.relu:
            gin = torch.where(x.contiguous() > 0, gin, torch.zeros_like(gin))      # ATen threshold_backward
        got = grad_input[0].detach().cpu().contiguous()
        assert got.dtype == gin.dtype and same(got, gin), ("input gradient", self.name, self.steps - 1)


def all_site_names(model):
    """every operator site `convert_pq` built: activation sites (`Sequential(Sequential(act, P), Q)`, `Sequential(act, Q)`) and
    every QuantizeLayer outside one (the input quantizer, the weight quantizers `<layer>.quantize`)"""
    sites, inside = [], set()
    for name, m in model.named_modules():
        if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[1], QuantizeLayer):
            sites.append(name)
            inside.update(id(c) for c in m.modules())
    for name, m in model.named_modules():
        if isinstance(m, QuantizeLayer) and id(m) not in inside:
            sites.append(name)
    return sites


def run_sites(arch, device, site_names, batch, width=64, channels_last=False, steps=7, autocast=True, schedule=None):
    """`site_names`: a list, or "all" for every site of the converted network"""
    schedule = schedule or SCHEDULE
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    torch.manual_seed(0)
    if arch == "resnet18":
        model, shape, classes, sparsity = resnet18(10, True, width), (batch, 3, 32, 32), 10, 0.5
    else:
        model, shape, classes, sparsity = resnet50(1000 if width == 64 else 10, False, width), (batch, 3, 224, 224), \
            (1000 if width == 64 else 10), 0.75
    model = convert_pq(model, sparsity=sparsity, bits=4, **schedule).to(device).train()
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    modules = dict(model.named_modules())
    if site_names == "all":
        site_names = all_site_names(model)
    log, checkers = [], []
    for name in site_names:
        ck = SiteChecker(name, modules[name], sparsity, 4, log, schedule)
        modules[name].register_forward_hook(ck.fwd)
        if not isinstance(modules[name], QuantizeLayer) or name != "0":
            modules[name].register_full_backward_hook(ck.bwd)
        checkers.append(ck)
    opt = torch.optim.SGD(model.parameters(), lr=0.02, momentum=0.9)
    g = torch.Generator().manual_seed(1)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                       # the staged mean's bits are ATen's with ONE intra-op thread
    try:
        for s in range(steps):
            x = torch.randn(shape, generator=g).to(device)
            if channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            y = torch.randint(0, classes, (batch,), generator=g).to(device)
            opt.zero_grad()
            with torch.autocast(device.split(":")[0], dtype=torch.bfloat16, enabled=autocast):
                loss = F.cross_entropy(model(x).float(), y)
            loss.backward()
            opt.step()
            assert torch.isfinite(loss).item()
    finally:
        torch.set_num_threads(threads)
    for ck in checkers:
        assert ck.steps == steps, (ck.name, ck.steps)
    return log, model


RN50_SITES = ["0", "1.stem.2", "1.stages.0.relu1", "1.stages.0.relu3", "1.stages.3.relu2", "1.stages.7.relu2", "1.stages.9.relu3",
              "1.stages.13.relu2", "1.stages.15.relu3", "1.stages.3.conv2.quantize", "1.fc.quantize"]
RN18_SITES = ["0", "1.stem.2", "1.stages.0.relu1", "1.stages.1.relu2", "1.stages.2.relu1", "1.stages.4.relu2", "1.stages.6.relu1",
              "1.stages.7.relu2", "1.stages.5.conv1.quantize", "1.fc.quantize"]


def _summary(log, steps):
    final = {}
    for name, s, shape, dtype, quantized, kept in log:
        if s == steps - 1:
            final[name] = (shape, dtype, quantized, kept)
    return final


def test_site_harness_on_cpu_miniature():
    """the harness itself, on the CPU path with narrow networks (no GPU needed): sites of every structure, fp32"""
    log, _ = run_sites("resnet18", "cpu", RN18_SITES, batch=4, width=8, autocast=False)
    final = _summary(log, 7)
    assert final["1.stem.2"][2] and abs(final["1.stages.0.relu1"][3] - 0.5) < 0.13
    log, _ = run_sites("resnet50", "cpu", ["0", "1.stem.2", "1.stages.0.relu3", "1.stages.15.relu3", "1.fc.quantize"], batch=2,
                       width=8, autocast=False, steps=6)


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last", [False, True])
def test_resnet50_imagenet_sites_vs_oracle(channels_last):
    """BASELINE config 4: full-width ResNet-50, 224x224, batch 16, bf16 autocast, 4-bit, 75 % channel pruning"""
    log, model = run_sites("resnet50", "cuda", RN50_SITES, batch=16, channels_last=channels_last)
    final = _summary(log, 7)
    assert final["1.stem.2"][0] == (16, 64, 112, 112) and final["1.stages.0.relu3"][0] == (16, 256, 56, 56)
    assert final["1.stages.7.relu2"][0][2:] == (14, 14) and final["1.stages.13.relu2"][0][2:] == (7, 7)
    assert final["1.stages.0.relu1"][1] == "torch.bfloat16"            # autocast really handed the sites bf16 activations
    for name in ("1.stem.2", "1.stages.0.relu3", "1.stages.7.relu2", "1.stages.13.relu2"):
        # quantizing, ~75 % pruned (ties in the bf16 magnitudes keep a few channels more: SURVEY quirk B18)
        assert final[name][2] and 0.2 <= final[name][3] <= 0.45, (name, final[name])
    assert final["1.stages.15.relu3"][3] is None and final["1.stages.15.relu3"][2]           # last ReLU: quantize only


@pytest.mark.gpu
@pytest.mark.parametrize("channels_last", [False, True])
def test_resnet18_cifar_sites_vs_oracle(channels_last):
    """BASELINE config 3: full-width ResNet-18, 32x32, batch 64, bf16 autocast, 4-bit, 50 % channel pruning"""
    log, model = run_sites("resnet18", "cuda", RN18_SITES, batch=64, channels_last=channels_last)
    final = _summary(log, 7)
    assert final["1.stem.2"][0] == (64, 64, 32, 32) and final["1.stages.6.relu1"][0] == (64, 512, 4, 4)
    for name in ("1.stem.2", "1.stages.2.relu1", "1.stages.6.relu1"):
        assert final[name][2] and 0.45 <= final[name][3] <= 0.7, (name, final[name])


FAST = dict(prune_start=1, prune_interval=1, repetition=1, quant_timeout=1)     # inactive -> live -> steady state in 3 steps


def test_all_sites_harness_on_cpu_miniature():
    log, model = run_sites("resnet50", "cpu", "all", batch=2, width=8, autocast=False, steps=3, schedule=FAST)
    names = {n for n, *_ in log}
    assert len(names) == 49 + 54 + 1              # activation sites, conv / fc weights, the input quantizer


@pytest.mark.gpu
def test_resnet50_every_operator_site_vs_oracle():
    """BASELINE config 4, EVERY operator site of the full-width network (49 activation sites, 54 weight quantizers, the
    input quantizer) hooked for three steps at batch 8, channels_last, bf16 autocast: each site's output, input
    gradient, mask, magnitude, scale and counters against the oracle on the tensors it really received
    (reference sparse.py:215-273, quantize.py:473-518, imitation.py:42-68)"""
    log, model = run_sites("resnet50", "cuda", "all", batch=8, channels_last=True, steps=3, schedule=FAST)
    final = _summary(log, 3)
    assert len(final) == 104
    live = [v for v in final.values() if v[3] is not None]
    assert len(live) == 48 and all(v[2] for v in final.values())          # every site quantizes; 48 of the 49 also prune
    # 75 % pruning keeps a quarter of the channels -- plus every tie at the threshold (`>=`, util.py:117): at batch 8 a deep
    # site can have enough dead (all-zero) channels for the threshold itself to be zero, and which ones are dead varies with
    # MIOpen's algorithm choice from run to run (0.53 seen once at 1.stages.5.relu2); parity is checked inside run_sites
    kept = sorted(v[3] for v in live)
    odd = {k: v[3] for k, v in final.items() if v[3] is not None and not 0.2 <= v[3] <= 0.8}
    assert not odd and kept[len(kept) // 2] <= 0.35, (odd, kept)


@pytest.mark.gpu
def test_resnet50_imagenet_sites_at_batch_64_vs_oracle():
    """the eleven chosen sites at batch 64 (the stem is a 51 M-element tensor): the grids a batch of 16 does not reach"""
    log, model = run_sites("resnet50", "cuda", RN50_SITES, batch=64, channels_last=True, steps=4, schedule=FAST)
    final = _summary(log, 4)
    assert final["1.stem.2"][0] == (64, 64, 112, 112) and final["1.stages.0.relu3"][0] == (64, 256, 56, 56)
