"""The numerical contract of the staged mean is "ATen's CPU result with ONE intra-op thread, torch 2.10" (INTEGRATION.md).
Two things keep it honest in the CPU suite: the golden fixtures (recorded from the reference under that contract:
tests/test_oracle_golden.py fails if a torch upgrade changes SumKernel's order) and this probe of how far ATen's own
result moves with the thread count (tools/probe_aten_threads.py) -- the facts the contract sentence rests on:
NCHW-contiguous inputs do not depend on the split up to 8 threads, channels_last inputs agree for 2 threads and start
to move from 4 on for a few shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import probe_aten_threads as probe


def test_aten_staged_mean_vs_intra_op_threads():
    before = torch.get_num_threads()
    try:
        total, stats, examples = probe.run(channels_last=False, seed=0, cases=120)
        assert total > 60 and stats == {2: 0, 4: 0, 8: 0}, (stats, examples)
        total, stats, examples = probe.run(channels_last=True, seed=1, cases=120)
        assert total > 60 and stats[2] == 0, (stats, examples)
        # 4 / 8 threads: ATen's own channels_last result moves for a handful of small-channel shapes (the reason the oracle and
        # every parity test run with one thread); if this ever exceeds a few per cent the contract sentence needs rewriting
        assert stats[4] <= 0.05 * total and stats[8] <= 0.1 * total, (stats, examples)
    finally:
        torch.set_num_threads(before)


def test_aten_reduces_h_of_a_channels_last_sample_like_the_batch_of_its_rows():
    """What the GPU route for channels_last activations whose batch dim is NOT reduced rests on (qsparse_amd/util.py,
    `_staged_mean_hip`: per-sample masks, and every batch of one): ATen's ``mean(2)`` of an NHWC tensor -- it reduces H in
    place, into an NCHW-contiguous result -- sums one sample exactly as its ``mean(0)`` sums the [H, C, 1, W] channels_last
    tensor that the sample's memory also is.  The latter order is the one ``qs_mean_dim_cl`` reproduces on the GPU."""
    before = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        g = torch.Generator().manual_seed(0)
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            for shape in [(1, 64, 14, 14), (2, 8, 7, 7), (1, 130, 5, 9), (1, 3, 28, 28), (1, 256, 56, 56), (3, 31, 3, 32), (2, 6, 28, 28),
                          (1, 7, 300, 5), (1, 16, 17, 70), (1, 2, 64, 64), (1, 1, 9, 9)]:
                N, C, H, W = shape
                x = torch.randn(shape, generator=g).abs().to(dtype).contiguous(memory_format=torch.channels_last)
                want = x.mean(2, keepdim=True)
                assert want.is_contiguous()
                rows = [x.permute(0, 2, 3, 1)[n].reshape(H, 1, W, C).permute(0, 3, 1, 2).mean(0, keepdim=True) for n in range(N)]
                assert torch.equal(want, torch.cat(rows, 0)), (dtype, shape)
                # ... and it is NOT the order of the NCHW copy (why a copy-and-reduce route is a last-bit off in float32)
                # the channel dim reduced first (batch kept): the inner reduction of the [N*H*W, C] matrix the memory is
                want = x.mean(1, keepdim=True)
                assert want.is_contiguous()
                assert torch.equal(want, x.permute(0, 2, 3, 1).reshape(N * H * W, C).mean(1).view(N, 1, H, W)), (dtype, shape)
            # channels_last_3d, batch reduced first: the batch reduction of the [N, C, 1, D*H*W] channels_last tensor
            for shape in [(8, 16, 3, 4, 8), (5, 32, 2, 7, 7), (16, 3, 4, 4, 4), (3, 130, 1, 5, 9)]:
                N, C, D, H, W = shape
                x = torch.randn(shape, generator=g).abs().to(dtype).contiguous(memory_format=torch.channels_last_3d)
                want = x.mean(0, keepdim=True)
                z = x.permute(0, 2, 3, 4, 1).reshape(N, 1, D * H * W, C).permute(0, 3, 1, 2)
                assert want.is_contiguous() and torch.equal(want, z.mean(0, keepdim=True).reshape(1, C, D, H, W)), (dtype, shape)
        x = torch.randn(1, 64, 14, 14, generator=g).abs().contiguous(memory_format=torch.channels_last)
        assert not torch.equal(x.mean(2, keepdim=True), x.contiguous().mean(2, keepdim=True))
        assert not torch.equal(x.mean(1, keepdim=True), x.contiguous().mean(1, keepdim=True))
    finally:
        torch.set_num_threads(before)
