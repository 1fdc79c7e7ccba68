#!/usr/bin/env python3
"""Statistics pass (qs_mean_dim over the batch dim with the fused per-channel abs-max) on the activation shapes
of a ResNet-50 step at batch 64, for every row-split width (QS_MEAN_SPLIT; 0 = the host's own choice).
Development tool: shows where the kernel is latency- rather than bandwidth-bound."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from qsparse_amd import _hip

SHAPES = [(64, 64, 56, 56), (64, 256, 56, 56), (64, 128, 28, 28), (64, 512, 28, 28), (64, 256, 14, 14), (64, 1024, 14, 14),
          (64, 512, 7, 7), (64, 2048, 7, 7), (128, 64, 32, 32), (128, 512, 4, 4), (256, 64, 56, 56), (128, 64, 56, 56),
          (256, 128, 28, 28), (256, 256, 56, 56)]


B256 = [(256, 64, 112, 112), (256, 64, 56, 56), (256, 256, 56, 56), (256, 128, 56, 56), (256, 128, 28, 28), (256, 512, 28, 28),
        (256, 256, 28, 28), (256, 256, 14, 14), (256, 1024, 14, 14), (256, 512, 14, 14), (256, 512, 7, 7), (256, 2048, 7, 7)]


def t_us(fn, iters=30, warm=5):
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
    return ts[len(ts) // 2] * 1e3


def main():
    lib = _hip.load()
    dev = "cuda"
    noabs = "--noabs" in sys.argv
    stride = 1 if "--dense" in sys.argv else 32     # abs-max accumulator: dense float[C] or one 128-byte line per channel
    channels_last = "--cl" in sys.argv
    tile_sweep = "--tile" in sys.argv            # channels_last one-wave kernel: linear columns (0) vs tiles of 8 / 4 lanes per position (QS_CL_TILE)
    depth_sweep = "--depth" in sys.argv          # sweep the rows in flight per wave (QS_MEAN_DEPTH) of the unsplit kernel instead
    # cells: R (waves per workgroup, 0 = the host's own choice) or R:lanes (channels_last: QS_CL_LANES)
    splits = [s for s in sys.argv[1:] if s.replace(":", "").isdigit()] or (["0", "16", "32"] if depth_sweep else ["0", "1", "2", "4", "8"])
    if tile_sweep:
        splits = ["0", "8", "4"]
    print(f"{'shape':24s} {'dtype':6s} " + " ".join(f"{('T=' if tile_sweep else 'D=' if depth_sweep else 'R=') + str(r):>14s}" for r in splits))
    for shp in (B256 if "--b256" in sys.argv else SHAPES):
        N, C, H, W = shp
        for dtype, code, nbytes in ((torch.bfloat16, 1, 2), (torch.float32, 0, 4)):
            nrot = max(1, min(6, int(6e8 // (N * C * H * W * nbytes))))     # rotate inputs past the 256 MiB Infinity Cache
            xs = [torch.randn(shp, device=dev).to(dtype) for _ in range(nrot)]
            x = xs[0]
            turn = [0]
            stage = torch.empty(C * H * W, device=dev, dtype=dtype)
            amax = torch.zeros(C * 32, device=dev)
            part = torch.empty(C * H * W, device=dev)
            cells = []
            for cell in splits:
                r = int(cell.split(":")[0])
                os.environ["QS_CL_LANES"] = cell.split(":")[1] if ":" in cell else "0"
                if tile_sweep:
                    os.environ["QS_MEAN_SPLIT"], os.environ["QS_CL_TILE"] = "0", str(r)
                elif depth_sweep:
                    os.environ["QS_MEAN_SPLIT"], os.environ["QS_MEAN_DEPTH"] = ("1" if r else "0"), str(r)
                else:
                    os.environ["QS_MEAN_SPLIT"] = str(r)

                def stats_cl():      # the same activation in NHWC memory order through qs_mean_dim_cl
                    turn[0] += 1
                    assert lib.qs_mean_dim_cl(xs[turn[0] % nrot].data_ptr(), stage.data_ptr(), N, H * W, C, code, code, 1 | 4,
                                              None, None if noabs else part.data_ptr(), None) == 0

                def stats():
                    turn[0] += 1
                    assert lib.qs_mean_dim(xs[turn[0] % nrot].data_ptr(), stage.data_ptr(), 1, N, C * H * W, code, code, 1 | 4, None,
                                           None if noabs else amax.data_ptr(), stride, H * W, C, None) == 0

                us = t_us(stats_cl if channels_last else stats)
                cells.append(f"{us:6.1f}us {x.numel() * nbytes / us / 1e3:5.0f}GB/s"[:14].rjust(14))
            print(f"{str(shp):24s} {str(dtype)[6:]:6s} " + " ".join(cells), flush=True)
    os.environ["QS_MEAN_SPLIT"] = os.environ["QS_MEAN_DEPTH"] = os.environ["QS_CL_LANES"] = os.environ["QS_CL_TILE"] = "0"


if __name__ == "__main__":
    main()
