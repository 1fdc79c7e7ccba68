#!/usr/bin/env python3
"""Randomised differential test of the autocast image (qsparse_amd/fused.py; the default under torch.autocast since round 5):
every case builds a convert-style activation site in front of real autocast consumers, runs a few training / evaluation steps
under `torch.autocast` TWICE on the GPU -- with the image (default) and with `autocast_image=False` -- and compares everything the
user can observe bit for bit: the site's output, the consumers' outputs, input and parameter gradients, gradients seen by hooks
registered before / after the consumer ran, retained gradients, `torch.autograd.grad` with respect to the output, and the
operators' state.  (The reference has no such route: its values are those of the run without the image, which the other
harnesses hold against the CPU path and the reference itself.)  Random: shapes (ragged maps whose image is a cast of y, 2-d
activations), input dtype (bf16 / fp16 behind a convolution, fp32 behind a residual add), autocast dtype, layout, activation
module (in place or not), quantizer kind, saturation, schedules (frozen masks), consumer / observer mix, evaluation steps.

    python3 tests/fuzz/fuzz_image.py [cases=200] [seed=0]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs
from golden_io import same
from qsparse_amd.fused import AutocastImageTensor

DEV = "cuda"
# MIOpen is asked for deterministic algorithms and the consumers' weight gradients are compared bit for bit like everything else
# (QS_FUZZ_STRICT_GW=0: MIOpen's default algorithms, whose fp16 weight-gradient kernels are not run-to-run deterministic -- the
#  weight gradients are then held to a bound relative to their largest entry; round 6: every mismatch seen under that setting
#  disappeared under this one, 3,000 cases)
STRICT_GW = os.environ.get("QS_FUZZ_STRICT_GW", "1") == "1"
if STRICT_GW:
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
USED = [0]          # cases in which an image was really handed to a consumer


def build(rng):
    n = rng.choice([2, 3, 4, 8, 16, 33])
    c = rng.choice([8, 16, 24, 64, 96])
    hw = rng.choice([(8, 8), (7, 7), (14, 14), (5, 6), (16, 16), (3, 3), None])
    shape = (n, c) if hw is None else (n, c) + hw
    xdt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32, torch.float16])
    adt = torch.float16 if xdt == torch.float16 else rng.choice([torch.bfloat16, torch.bfloat16, torch.float16])
    # (gelu / identity: activations the kernels do not fold -- under autocast such a site folds the identity to hand out its image)
    act = rng.choice(["relu", "relu", "relu", "relu6", "leaky", "hardtanh", "gelu", "gelu", "identity"])
    inplace = rng.random() < 0.3
    kind = rng.choice(["scaler", "scaler", "decimal"])
    policy = rng.choice(["default", "default", "freeze", "no_avg"])
    desc = dict(shape=shape, xdt=str(xdt)[6:], adt=str(adt)[6:], act=act, inplace=inplace, quantizer=kind, policy=policy,
                bits=rng.choice([2, 4, 8]), sparsity=rng.choice([0.3, 0.5, 0.75]), start=rng.choice([0, 1, 2]), timeout=rng.choice([1, 2]),
                saturate=rng.random() < 0.2, cl=len(shape) == 4 and rng.random() < 0.5, steps=rng.choice([5, 7, 9]),
                consumers=rng.sample(["conv", "conv2", "residual", "cat", "view", "pool"], k=rng.choice([1, 1, 2, 3])),
                observer=rng.choice([None, None, "hook_before", "hook_after", "hook_replace", "retain_after", "grad_wrt_output"]),
                eval_at=rng.choice([None, None, 3]), consumer_first=rng.random() < 0.8,
                # the consumers' own weights read through quantizers: the weight path hands them out with THEIR images (batch.py)
                quant_weights=rng.choice([None, None, -1, 0, 1]),
                # a quantize-only site (convert's Sequential(act, QuantizeLayer)): its image comes from qs_quantize_step (ABI v21)
                site=rng.choice(["pair", "pair", "act_q"]))
    if desc["site"] == "act_q":
        desc["quantizer"] = "scaler"
    # a residual block in FRONT of the site: another site's float32 output y0, a convolution of it (bf16 under autocast) and the
    # type-promoting sum `conv(y0) + y0` as the site's input -- the sum goes through the library's add node, whose backward hands
    # the bf16 operand the gradient image the site's backward kernel wrote (fused.py, "Promoting add")
    desc["front"] = rng.choice([None, None, "add", "add_shared"])
    return desc


class Net(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.d = d
        C = d["shape"][1]
        a = {"relu": nn.ReLU, "relu6": nn.ReLU6, "leaky": lambda inplace: nn.LeakyReLU(0.1, inplace=inplace),
             "hardtanh": lambda inplace: nn.Hardtanh(-0.75, 1.5, inplace=inplace), "gelu": lambda inplace: nn.GELU(),
             "identity": lambda inplace: nn.Identity()}[d["act"]](inplace=d["inplace"])
        cbkw = {"default": {}, "freeze": dict(mask_refresh_interval=1, stop_mask_refresh=2), "no_avg": dict(running_average=False)}[d["policy"]]
        net = nn.Sequential(a)
        types = [type(a)]
        if d["site"] == "pair":
            net = qs.convert(net, qs.prune(sparsity=d["sparsity"], dimensions={1}, start=d["start"], interval=1, repetition=2,
                                           callback=qs.MagnitudePruningCallback(**cbkw)), activation_layers=types, log=False)
        qcb = qs.ScalerQuantizer() if d["quantizer"] == "scaler" else qs.DecimalQuantizer()
        if d["saturate"]:
            qcb.saturate = True
        self.site = qs.convert(net, qs.quantize(bits=d["bits"], channelwise=-1, timeout=d["timeout"], callback=qcb), activation_layers=types,
                               log=False)
        torch.manual_seed(7)
        flat = len(d["shape"]) == 2
        self.main = nn.Linear(C, 12, bias=False) if flat else nn.Conv2d(C, 12, 1, bias=False)
        self.second = nn.Linear(C, 12, bias=False) if flat else nn.Conv2d(C, 12, 1, bias=False)
        self.observed, self.kinds, self.held = [], [], None
        if d.get("front"):
            pre = nn.Sequential(nn.ReLU())
            pre = qs.convert(pre, qs.prune(sparsity=0.5, dimensions={1}, start=0, interval=1, repetition=1), activation_layers=[nn.ReLU], log=False)
            self.pre = qs.convert(pre, qs.quantize(bits=8, channelwise=-1, timeout=1), activation_layers=[nn.ReLU], log=False)
            self.front = nn.Linear(C, C, bias=False) if flat else nn.Conv2d(C, C, 1, bias=False)

    def forward(self, x):
        d = self.d
        h = x.clone() if d["inplace"] else x            # (an in-place activation needs a non-leaf input, as behind a convolution)
        extra = None
        if d.get("front"):
            y0 = self.pre(x)
            h = self.front(y0) + y0
            if d["front"] == "add_shared":              # a second consumer of the sum: the add's backward must cast for itself
                extra = (h * 0.5).float().mean(tuple(range(2, h.dim())))[:, :12] if h.dim() > 2 else (h * 0.5).float()[:, :12]
        y = self.site(h)
        self.kinds.append(type(y))
        flat = y.dim() == 2
        red = (lambda t: t.float()) if flat else (lambda t: t.float().mean((2, 3)))
        outs = []
        if d["observer"] == "hook_before" and y.requires_grad:
            y.register_hook(lambda g: self.observed.append(g.detach().clone()))
        cons = list(d["consumers"])
        if not d["consumer_first"]:
            cons = cons[::-1]
        took = False
        for k in cons:
            if k == "conv":
                outs.append(red(self.main(y)))
                took = True
            elif k == "conv2":
                outs.append(red(self.second(y)))
                took = True
            elif k == "residual":
                outs.append(red(y * 0.25)[:, :12])
            elif k == "cat":
                outs.append(red(torch.cat([y, y * 2.0], 1))[:, :12])
            elif k == "view":
                outs.append(red(y)[:, :12] + (y.flatten(1)[:, :12].float()))
            elif k == "pool" and not flat:
                outs.append(F.adaptive_avg_pool2d(y, 1).flatten(1)[:, :12].float())
            if took and d["observer"] in ("hook_after", "hook_replace", "retain_after") and y.requires_grad and self.held is not y:
                self.held = y
                if d["observer"] == "hook_after":
                    y.register_hook(lambda g: self.observed.append(g.detach().clone()))
                elif d["observer"] == "hook_replace":
                    y.register_hook(lambda g: g * 0.5)
                else:
                    y.retain_grad()
        if d["observer"] == "grad_wrt_output":
            self.held = y
        if not outs:
            outs.append(red(y)[:, :12])
        out = outs[0]
        for o in outs[1:]:
            out = out + o
        if extra is not None:
            out = out + extra
        return out


def run(d, image, seed):
    # (the run without the image is also the run without the fused backward of an nn.GELU in front of the site: ATen's own pass there)
    qs.set_qsparse_options(autocast_image=image, act_backward=image)
    try:
        net = Net(d)
        if d["quant_weights"] is not None:
            net = qs.convert(net, qs.quantize(bits=8, timeout=1, channelwise=d["quant_weights"]), weight_layers=[nn.Conv2d, nn.Linear], log=False)
        net = net.to(DEV).train()
        g = torch.Generator().manual_seed(seed)
        C = d["shape"][1]
        xdt = getattr(torch, d["xdt"])
        adt = getattr(torch, d["adt"])
        trace = []
        for s in range(d["steps"]):
            x = torch.randn(d["shape"], generator=g) * torch.linspace(0.3, 3, C).view([1, -1] + [1] * (len(d["shape"]) - 2))
            x = x.to(xdt).to(DEV)
            if d["cl"]:
                x = x.contiguous(memory_format=torch.channels_last)
            evaluating = d["eval_at"] is not None and s == d["eval_at"]
            net.train(not evaluating)
            net.held = None
            if evaluating:
                with torch.no_grad(), torch.autocast("cuda", dtype=adt):
                    trace.append(("eval", net(x).detach().clone()))
                continue
            x.requires_grad_(True)
            with torch.autocast("cuda", dtype=adt):
                out = net(x)
            w = torch.linspace(-1, 1, out.shape[1], device=DEV)
            loss = (out * w).sum()
            if d["observer"] == "grad_wrt_output" and net.held is not None and net.held.requires_grad:
                gy, gx = torch.autograd.grad(loss, [net.held, x], allow_unused=True)
                trace += [("gy", None if gy is None else gy.as_subclass(torch.Tensor).clone()), ("gx", gx.clone())]
            else:
                loss.backward()
                gw = net.main._parameters["weight"].grad
                trace += [("out", out.detach().clone()), ("gx", x.grad.clone()), ("gw", None if gw is None else gw.clone())]
                if d.get("front"):
                    gf = net.front._parameters["weight"].grad
                    trace.append(("gw", None if gf is None else gf.clone()))
                if d["observer"] == "retain_after" and net.held is not None:
                    gr = net.held.grad
                    trace.append(("retained", None if gr is None else gr.as_subclass(torch.Tensor).clone()))
            net.zero_grad()
        state = {k: v.detach().clone() for k, v in net.state_dict().items() if not k.endswith((".weight", ".bias")) or "quantize" in k or "site" in k}
        return trace, net.observed, net.kinds, state
    finally:
        qs.set_qsparse_options(autocast_image=True, act_backward=True)


def one_case(rng, idx):
    d = build(rng)
    d["i"] = idx
    res = {}
    for image in (False, True):
        try:
            res[image] = run(d, image, 5000 + idx)
        except Exception as e:      # noqa: BLE001 -- both runs must fail alike
            res[image] = ("raised", type(e).__name__, str(e)[:160])
    a, b = res[False], res[True]
    if len(a) == 3 or len(b) == 3:
        if len(a) == 3 and len(b) == 3 and a[1] == b[1]:
            return "ok"
        return dict(d, plain=a if len(a) == 3 else "ran", image=b if len(b) == 3 else "ran")
    (ta, oa, ka, sa), (tb, ob, kb, sb) = a, b
    if AutocastImageTensor in kb:
        USED[0] += 1
    if AutocastImageTensor in ka:
        return dict(d, mismatch="the run without the image produced one")
    if len(ta) != len(tb) or len(oa) != len(ob):
        return dict(d, mismatch="number of outputs / hook calls")
    for i, ((ka_, va), (kb_, vb)) in enumerate(zip(ta, tb)):
        if ka_ == kb_ == "gw" and va is not None and vb is not None and not STRICT_GW:
            # the CONSUMER's weight gradient: MIOpen's fp16 weight-gradient kernels are not run-to-run deterministic (two runs of
            # the plain route differ as well) -- the operands it gets are compared bit for bit through `out` and `gx`
            # (... and where large terms cancel, one fp16 ulp of a partial sum is large against the element itself: the bound is
            #  relative to the largest entry of the gradient)
            fa, fb = va.float().nan_to_num(0.0, 6e4, -6e4), vb.float().nan_to_num(0.0, 6e4, -6e4)
            if not (torch.equal(va.isnan(), vb.isnan()) and float((fa - fb).abs().max()) <= 8e-2 * float(fa.abs().max()) + 2e-3):
                if os.environ.get("QS_FUZZ_ONLY"):
                    print("gw", i, "max |a|", float(fa.abs().max()), "max |a - b|", float((fa - fb).abs().max()), "differing", int((fa != fb).sum()),
                          "of", fa.numel(), "nan", int(va.isnan().sum()), int(vb.isnan().sum()), "inf", int(va.isinf().sum()), int(vb.isinf().sum()), flush=True)
                return dict(d, mismatch=(i, ka_))
            continue
        if ka_ != kb_ or (va is None) != (vb is None) or (va is not None and not (va.dtype == vb.dtype and same(va.cpu(), vb.cpu()))):
            if os.environ.get("QS_FUZZ_ONLY") and va is not None and vb is not None and va.shape == vb.shape:
                fa, fb = va.float().cpu().flatten(), vb.float().cpu().flatten()
                bad = ((fa != fb) & ~(fa.isnan() & fb.isnan())).nonzero().flatten()
                print(ka_, i, "differing", bad.numel(), "of", fa.numel(), [(int(j), float(fa[j]), float(fb[j])) for j in bad[:8]], flush=True)
            return dict(d, mismatch=(i, ka_))
    for i, (va, vb) in enumerate(zip(oa, ob)):
        if not (va.dtype == vb.dtype and same(va.cpu(), vb.cpu())):
            return dict(d, mismatch=("hook", i))
    for k in sa:
        if not same(sa[k].cpu(), sb[k].cpu()):
            return dict(d, mismatch=("state", k))
    return "ok"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    only = os.environ.get("QS_FUZZ_ONLY")
    fails = ran = 0
    for i in range(cases):
        if only is not None and i != int(only):
            build(rng)
            continue
        ran += 1
        r = one_case(rng, i)
        if r != "ok":
            fails += 1
            print("FAIL", r, flush=True)
    from qsparse_amd.fused import ROUTES
    print(f"fuzz image-vs-plain: {ran} cases, {fails} failures (seed {seed}); an image was consumed in {USED[0]}; routes: {dict(ROUTES)}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
