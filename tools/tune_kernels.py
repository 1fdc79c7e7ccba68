#!/usr/bin/env python3
"""Kernel tuning sweep (development tool, runs on the GPU box).

Builds nothing: expects variant libraries build_variants/libqs_*.so (see tools/build_variants.sh) and
times the three headline kernels of each through the C ABI with HIP events:
    apply fwd  : qs_quant_scaler_fwd  bf16 -> fp32 with channel mask      (6 B/elem)
    apply bwd  : qs_quant_ste_bwd     fp32 -> bf16 with channel mask      (6 B/elem)
    stats      : qs_mean_dim          bf16 read, per-channel absmax fused (2 B/elem)
on the 256x256x56x56 headline tensor, for a few QS_MEAN_LANES settings (further knobs: QS_MAX_BLOCKS,
QS_REDUCE_BLOCKS, QS_MEAN_SPLIT, QS_EW_REVERSE).
"""
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from qsparse_amd import _hip

SHAPE = (256, 256, 56, 56)
PEAK = 8000.0
ELIDE_FWD = int(os.environ.get("QS_TUNE_ELIDE_FWD", "1"))   # mask-aware elision in the forward (library default)
ELIDE_BWD = int(os.environ.get("QS_TUNE_ELIDE_BWD", "0"))


def time_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def run_variant(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _hip.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    dev = "cuda"
    N, C, H, W = SHAPE
    numel = N * C * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.randn(SHAPE, generator=g, device=dev).relu_() * torch.linspace(0.25, 4, C, device=dev).view(1, C, 1, 1)).bfloat16()
    gout = torch.randn(SHAPE, generator=g, device=dev)
    y = torch.empty(SHAPE, device=dev)
    gx = torch.empty(SHAPE, device=dev, dtype=torch.bfloat16)
    mask = (torch.arange(C, device=dev) % 4 == 0).to(torch.uint8)
    scale = torch.tensor([0.37], device=dev)
    stage = torch.empty(C * H * W, device=dev, dtype=torch.bfloat16)
    amax = torch.empty(C, device=dev)
    st = 0

    def fwd():
        assert lib.qs_quant_scaler_fwd(x.data_ptr(), y.data_ptr(), None, scale.data_ptr(), 1, 0.0, mask.data_ptr(), N, C,
                                       H * W, 1, 0, 0, 0, 0, 0, 0, ELIDE_FWD, None, None, 0, None, None) == 0

    def bwd():
        assert lib.qs_quant_ste_bwd(gout.data_ptr(), gx.data_ptr(), scale.data_ptr(), 1, 0.0, 0, -8.0, 7.0, 0,
                                    mask.data_ptr(), N, C, H * W, 0, 1, ELIDE_BWD, None) == 0

    def stats():
        assert lib.qs_mean_dim(x.data_ptr(), stage.data_ptr(), 1, N, C * H * W, 1, 1, 1, None, amax.data_ptr(), 1, H * W, C,
                               None) == 0

    am1 = torch.empty(1, device=dev)

    def read_all():
        assert lib.qs_absmax(x.data_ptr(), am1.data_ptr(), 0, 1, 1, numel, 1, 0, 0, 1, None, 0, None) == 0

    def read_rows():
        assert lib.qs_absmax(x.data_ptr(), amax.data_ptr(), 1, N, C, H * W, 1, 0, 0, 1, None, 0, None) == 0

    imp = torch.empty(C, device=dev, dtype=torch.bfloat16)

    def last2():
        assert lib.qs_mean_last2(stage.data_ptr(), imp.data_ptr(), C, H, W, 1, 1, None, None, 1, None, None) == 0

    mag = torch.rand(C, device=dev)
    mk = torch.ones(C, device=dev, dtype=torch.uint8)
    sc = torch.ones(1, device=dev)

    def select():
        assert lib.qs_pq_select(mag.data_ptr(), imp.data_ptr(), 1, C, 1, 3, 1, 191, mk.data_ptr(), amax.data_ptr(), 1, 1, 3, 4,
                                sc.data_ptr(), None, None, None, None, None, None, 1, None, 1, None, None) == 0

    out = {}
    for name, fn, bpe in (("fwd", fwd, 6), ("bwd", bwd, 6), ("stats", stats, 2), ("read_all", read_all, 2),
                          ("read_rows", read_rows, 2), ("last2", last2, 0), ("select", select, 0)):
        med, best = time_ms(fn)
        out[name] = f"{med:.4f}ms/{bpe * numel / med / 1e6 / PEAK:.3f}" if bpe else f"{med * 1e3:.1f}us"
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        print(json.dumps(run_variant(sys.argv[2])))
        return
    # reference points: torch's own copy / cast kernels on the same tensors
    dev = "cuda"
    x = torch.randn(SHAPE, device=dev).bfloat16()
    y = torch.empty(SHAPE, device=dev)
    z = torch.empty(SHAPE, device=dev)
    numel = x.numel()
    for name, fn, bpe in (("torch bf16->f32 cast", lambda: y.copy_(x), 6), ("torch f32 copy", lambda: z.copy_(y), 8),
                          ("torch f32->bf16 cast", lambda: x.copy_(y), 6), ("torch bf16 abs-max", lambda: x.abs().amax(), 2)):
        med, best = time_ms(fn)
        print(f"{name:28s} {med:.4f} ms  {bpe * numel / med / 1e6:8.0f} GB/s  ({bpe * numel / med / 1e6 / PEAK:.3f} of 8 TB/s)",
              flush=True)
    del x, y, z
    torch.cuda.empty_cache()
    libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build_variants", "libqs_*.so")))
    envs = [{"QS_MEAN_LANES": bs} for bs in ("0",)]
    for lib in libs:
        for env in envs:
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, __file__, "--one", lib], env=e, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
            print(os.path.basename(lib), env, line, flush=True)


if __name__ == "__main__":
    main()
