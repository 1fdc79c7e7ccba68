#!/usr/bin/env python3
"""channels_last activations whose batch dim is not reduced (a batch of one, per-sample masks): the GPU staged mean against
ATen's one-thread CPU result for the same NHWC tensor, float32 and bf16 (development probe).
    python tools/probe_cl_n1.py        # on a GPU box: every line OK"""
import sys, torch
sys.path.insert(0, '.')
import qsparse_amd as qs
from qsparse_amd.util import squeeze_tensor_to_shape
torch.set_num_threads(1)
g = torch.Generator().manual_seed(0)
bad = tot = 0
for dtype in (torch.float32, torch.bfloat16):
    for (N, C, H, W) in [(1, 8, 7, 7), (1, 64, 14, 14), (1, 3, 28, 28), (1, 256, 56, 56), (1, 130, 5, 9), (4, 16, 7, 7), (3, 64, 14, 14)]:
        for target in ([1, C, 1, 1], [N, C, 1, 1]):
            x = torch.randn(N, C, H, W, generator=g).abs().to(dtype).contiguous(memory_format=torch.channels_last)
            want = squeeze_tensor_to_shape(x, target)
            got = squeeze_tensor_to_shape(x.cuda(), target).cpu()
            same = torch.equal(want, got)
            tot += 1; bad += (not same)
            print(dtype, (N, C, H, W), target, "OK" if same else f"DIFF {(want.float()-got.float()).abs().max().item():.3e} n={(want!=got).sum().item()}", want.stride(), got.stride())
print("bad", bad, "of", tot)
