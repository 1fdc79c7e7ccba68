#!/usr/bin/env python3
"""abs-max / min-max reductions (qs_absmax, qs_minmax) straight through the C ABI: tensor-wise and per-channel, on
the BASELINE shapes and batch-64 ResNet activations.  Development tool; QS_REDUCE_BLOCKS is read once per process."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(256, 256, 56, 56), (256, 64, 56, 56), (64, 256, 56, 56), (64, 64, 56, 56), (64, 512, 28, 28), (64, 1024, 14, 14),
          (64, 2048, 7, 7), (512, 512, 3, 3), (64, 64, 1, 1)]


def one():
    import torch

    from qsparse_amd import _hip

    lib = _hip.load()
    out = {}
    for shp in SHAPES:
        N, C, H, W = shp
        nrot = max(1, min(6, int(6e8 // (N * C * H * W * 2))))          # rotate inputs past the 256 MiB Infinity Cache
        xs = [torch.randn(shp, device="cuda").bfloat16() for _ in range(nrot)]
        turn = [0]

        class _Rot:                                                    # x.data_ptr() walks the ring
            def data_ptr(self):
                turn[0] += 1
                return xs[turn[0] % nrot].data_ptr()

            def numel(self):
                return xs[0].numel()
        x = _Rot()
        a1, ac = torch.zeros(1, device="cuda"), torch.zeros(C, device="cuda")
        mn, mx = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")

        def t_us(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            evs = []
            for _ in range(30):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            return round(ts[len(ts) // 2] * 1e3, 1)

        r = {}
        r["all"] = t_us(lambda: lib.qs_absmax(x.data_ptr(), a1.data_ptr(), 0, 1, 1, x.numel(), 1, 1, 0, 1, None, 0, None))
        a16 = torch.zeros(16, 32, device="cuda")
        r["all_16lines"] = t_us(lambda: lib.qs_absmax(x.data_ptr(), a16.data_ptr(), 0, 1, 1, x.numel(), 1, 1, 0, 16, None, 0, None))
        r["chan"] = t_us(lambda: lib.qs_absmax(x.data_ptr(), ac.data_ptr(), 1, N, C, H * W, 1, 1, 0, 1, None, 0, None))
        nb = lib.qs_workspace_bytes(2, C)
        if nb and C % 8 == 0:      # the same tensor taken as channels_last: [N*H*W, C], channel dim innermost (reduce_fewcols_kernel)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            r["chan_cl"] = t_us(lambda: lib.qs_absmax(x.data_ptr(), ac.data_ptr(), 1, N * H * W, C, 1, 1, 1, 0, 1, ws.data_ptr(), nb, None))
            r["chan_cl_minmax"] = t_us(lambda: lib.qs_minmax(x.data_ptr(), mn.data_ptr(), mx.data_ptr(), 1, N * H * W, C, 1, 1, 0, ws.data_ptr(), nb, None))
            kmn, kmx = _hip.minmax_key_buffers(C, "cuda")
            r["chan_cl_minmax_keys"] = t_us(lambda: lib.qs_minmax(x.data_ptr(), kmn.data_ptr(), kmx.data_ptr(), 1, N * H * W, C, 1, 1, 1, ws.data_ptr(), nb, None))
        r["chan_minmax"] = t_us(lambda: lib.qs_minmax(x.data_ptr(), mn.data_ptr(), mx.data_ptr(), 1, N, C, H * W, 1, 0, None, 0, None))
        kmn, kmx = _hip.minmax_key_buffers(C, "cuda")      # accumulate mode: ONE launch, keys left for qs_lines_update(from_keys)
        r["chan_minmax_keys"] = t_us(lambda: lib.qs_minmax(x.data_ptr(), kmn.data_ptr(), kmx.data_ptr(), 1, N, C, H * W, 1, 1, None, 0, None))
        k1n, k1x = _hip.minmax_key_buffers(1, "cuda")
        r["all_minmax"] = t_us(lambda: lib.qs_minmax(x.data_ptr(), mn.data_ptr(), mx.data_ptr(), 0, 1, 1, x.numel(), 1, 0, None, 0, None))
        r["all_minmax_keys"] = t_us(lambda: lib.qs_minmax(x.data_ptr(), k1n.data_ptr(), k1x.data_ptr(), 0, 1, 1, x.numel(), 1, 1, None, 0, None))
        r["ideal@6TB/s"] = round(x.numel() * 2 / 6e6, 1)
        out[str(shp)] = r
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one()
    else:
        for rb in sys.argv[1:] or ["256"]:
            r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, QS_REDUCE_BLOCKS=rb), capture_output=True,
                               text=True)
            print(f"QS_REDUCE_BLOCKS={rb}")
            if not r.stdout.strip():
                print(r.stderr[-600:])
                continue
            for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items():
                print(f"  {k:22s} {v}")
