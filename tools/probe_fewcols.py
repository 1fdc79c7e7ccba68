#!/usr/bin/env python3
"""development probe: per-channel abs-max of a channels_last activation ([N*H*W, C]); run under rocprofv3 --kernel-trace --stats
to split the time between reduce_fewcols_kernel and its finish kernel (QS_FEWCOLS_BLOCKS caps the partial rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qsparse_amd import _hip
lib = _hip.load()
N, C, H, W = 256, 256, 56, 56
xs = [torch.randn(N, H, W, C, device="cuda").bfloat16() for _ in range(2)]
ac = torch.zeros(C, device="cuda")
nb = lib.qs_workspace_bytes(2, C)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
for i in range(20):
    assert lib.qs_absmax(xs[i % 2].data_ptr(), ac.data_ptr(), 1, N * H * W, C, 1, 1, 1, 0, 1, ws.data_ptr(), nb, None) == 0
torch.cuda.synchronize()
