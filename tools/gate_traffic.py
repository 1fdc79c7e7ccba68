#!/usr/bin/env python3
"""Driver for the PMC passes of the ReLU gate bitmap (tools/refresh_profiles.py <tag> gate): the fused forward and backward
of a channels_last fp32 site behind a residual add (256x256x56x56, 50 % channel mask), each with and without the bitmap,
six launches per variant.  The kernels are told apart by name: GateOp<...> forward, ste_relu_bwd_kernel<..., GATE>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from qsparse_amd import _hip

SHAPE = (256, 256, 56, 56)


def main():
    torch.manual_seed(0)
    x = torch.randn(SHAPE, device="cuda").contiguous(memory_format=torch.channels_last)
    g = torch.randn(SHAPE, device="cuda").contiguous(memory_format=torch.channels_last)
    scale = torch.full((1, 1), 0.1, device="cuda")
    mask = torch.rand(SHAPE[1], device="cuda") > 0.5
    for _ in range(6):
        _hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1, pre_relu=True)
        _, _, gate = _hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1, pre_relu=True,
                                    want_gate=True)
        _hip.ste_relu_bwd(g, x, scale, False, -8.0, 7.0, mask, 1)
        _hip.ste_relu_bwd(g, None, scale, False, -8.0, 7.0, mask, 1, gate=gate)
    torch.cuda.synchronize()
    print("numel", x.numel())


if __name__ == "__main__":
    main()
