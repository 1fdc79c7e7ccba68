#!/usr/bin/env python3
"""Per-call efficiency of the library's streaming kernels inside a converted ResNet training step (development tool):
every launch of the element-wise and statistics wrappers is bracketed with HIP events and listed per (kernel, shape,
dtype) with its algorithmic bytes, GB/s and the time above a 6 TB/s stream.  Shows which activation shapes of a real
network are far from the roofline.  In eager mode the rows of SMALL launches (below ~30 us) are dominated by the host:
the time between the two events includes waiting for the launch to be issued; trust them only for the large shapes and
use tools/profile_site.py --graph for the small ones.    site_efficiency.py [resnet50|resnet18] [batch] [channels_last]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import qsparse_amd as qs
from examples.models import convert_pq, resnet18, resnet50
from qsparse_amd import _hip

qs.set_qsparse_options(log_on_created=False, log_during_train=False)
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
if arch == "resnet18":
    base, shape, classes, sp = resnet18(10, True), (int(sys.argv[2]) if len(sys.argv) > 2 else 128, 3, 32, 32), 10, 0.5
else:
    base, shape, classes, sp = resnet50(1000, False), (int(sys.argv[2]) if len(sys.argv) > 2 else 64, 3, 224, 224), 1000, 0.75
m = convert_pq(base, sparsity=sp, bits=4, prune_start=1, prune_interval=1, repetition=1, quant_timeout=1).cuda().train()
x = torch.randn(shape, device="cuda")
y = torch.randint(0, classes, (shape[0],), device="cuda")
if "channels_last" in sys.argv:
    m = m.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
log = []
recording = [False]


def es(t):
    return t.element_size()


def wrap(name, bytes_of):
    orig = getattr(_hip, name)

    def timed(*a, **k):
        if not recording[0]:
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        t0 = a[1] if name in ("quant_fwd",) else a[0]
        log.append((name, tuple(t0.shape), str(t0.dtype)[6:], t0.is_contiguous(), bytes_of(a, k, out), e0, e1))
        return out

    setattr(_hip, name, timed)


first = lambda out: out[0] if isinstance(out, tuple) else out
wrap("quant_fwd", lambda a, k, out: a[1].numel() * (es(a[1]) + es(first(out))))
wrap("ste_bwd", lambda a, k, out: a[0].numel() * (es(a[0]) + es(out)))
wrap("ste_relu_bwd", lambda a, k, out: a[0].numel() * (es(a[0]) + es(a[1]) + es(out)))
wrap("mask_apply", lambda a, k, out: a[0].numel() * (es(a[0]) + es(out)))
wrap("absmax", lambda a, k, out: a[0].numel() * es(a[0]))
wrap("mean_dim", lambda a, k, out: a[0].numel() * es(a[0]))
wrap("mean_dim_cl", lambda a, k, out: a[0].numel() * es(a[0]))


def step():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.cross_entropy(m(x).float(), y)
    loss.backward()
    opt.step()


for _ in range(10):
    step()
recording[0] = True
for _ in range(4):
    step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for name, shp, dt, contig, nbytes, e0, e1 in log:
    a = agg[(name, shp, dt, contig)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
    a[2] = nbytes
rows = []
for (name, shp, dt, contig), (cnt, us, nbytes) in agg.items():
    per = us / cnt
    ideal = nbytes / 6e6            # us at 6 TB/s
    rows.append((max(per - ideal, 0) * cnt / 4, name, shp, dt, "nchw" if contig else "cl", cnt / 4, per, nbytes / per / 1e3, ideal))
rows.sort(reverse=True)
print(f"{'excess us/step':>14s} {'kernel':14s} {'shape':22s} {'dtype':8s} {'lay':4s} {'calls/step':>10s} {'us/call':>8s} {'GB/s':>7s} {'us@6TB/s':>9s}")
for ex, name, shp, dt, lay, cps, per, gbs, ideal in rows[:40]:
    print(f"{ex:14.1f} {name:14s} {str(shp):22s} {dt:8s} {lay:4s} {cps:10.1f} {per:8.1f} {gbs:7.0f} {ideal:9.1f}")
print(f"total measured {sum(r[6] * r[5] for r in rows):.0f} us/step, at 6 TB/s {sum(r[8] * r[5] for r in rows):.0f} us/step "
      "(event pairs add ~3 us per call)")
