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

import numpy as np


def multi_row(v):
    """SumKernel.cpp's cascade sum of the rows of v [n, columns], float32"""
    n = v.shape[0]
    lp = max(4, (0 if n <= 1 else (n - 1).bit_length()) // 4)
    step, lmask = 1 << lp, (1 << lp) - 1
    a = [np.zeros(v.shape[1], np.float32) for _ in range(4)]
    i = 0
    while i + step <= n:
        for j in range(step):
            a[0] = a[0] + v[i + j]
        i += step
        a[1] = a[1] + a[0]
        a[0] = np.zeros_like(a[0])
        if (i & (lmask << lp)) == 0:
            a[2] = a[2] + a[1]
            a[1] = np.zeros_like(a[1])
            if (i & (lmask << (2 * lp))) == 0:
                a[3] = a[3] + a[2]
                a[2] = np.zeros_like(a[2])
    while i < n:
        a[0] = a[0] + v[i]
        i += 1
    return ((a[0] + a[1]) + a[2]) + a[3]


def row_sum(v):
    """four interleaved cascade sums, the n % 4 last rows added to the first, ((p0 + p1) + p2) + p3"""
    n4 = v.shape[0] // 4
    p = [multi_row(v[k:n4 * 4:4]) if n4 else np.zeros(v.shape[1], np.float32) for k in range(4)]
    for i in range(n4 * 4, v.shape[0]):
        p[0] = p[0] + v[i]
    return ((p[0] + p[1]) + p[2]) + p[3]


def vector_inner(v):
    """the vectorised inner sum: 8 interleaved row-sums, the n % 8 tail, then the 8 lanes in turn"""
    n = v.shape[0]
    nv = n // 8
    fin = np.zeros(v.shape[1], np.float32)
    for i in range(nv * 8, n):
        fin = fin + v[i]
    for k in range(8):
        fin = fin + row_sum(v[k:nv * 8:8])
    return fin



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


def test_aten_reduces_w_of_a_channels_last_tensor_in_row_sum_order():
    """`x.mean(3, keepdim=True)` of a channels_last tensor (a mask that keeps N, C and H): ATen's scalar inner sum -- four
    interleaved cascade sums over w, the W % 4 last elements added to the first, ((p0 + p1) + p2) + p3 -- which is what
    qs_mean_cl_w restates (qsparse_amd/csrc/qs_reduce.h); NOT the order of the contiguous NCHW row"""
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        g = torch.Generator().manual_seed(0)
        differs_from_nchw = 0
        for N, C, H, W in [(2, 8, 5, 7), (4, 16, 14, 14), (3, 24, 7, 56), (2, 64, 28, 28), (8, 3, 9, 33), (1, 8, 5, 17), (2, 5, 3, 100),
                           (2, 8, 4, 300), (2, 8, 3, 3), (2, 16, 2, 9)]:
            x = (torch.randn(N, C, H, W, generator=g) * 3).contiguous(memory_format=torch.channels_last)
            a = x.abs().mean(3, keepdim=True)
            v = x.abs().permute(3, 0, 2, 1).contiguous().view(W, -1).numpy()          # [W, N*H*C]
            b = torch.from_numpy((row_sum(v) / np.float32(W)).astype(np.float32)).view(N, H, C).permute(0, 2, 1).unsqueeze(-1)
            assert torch.equal(a, b.contiguous()), (N, C, H, W)
            differs_from_nchw += int(not torch.equal(a, x.abs().contiguous().mean(3, keepdim=True)))
        assert differs_from_nchw >= 3          # (which is why the NCHW copy's order was a deviation)
    finally:
        torch.set_num_threads(threads)


def test_aten_reduce_plan_names_the_order_aten_takes_for_any_dense_layout():
    """`util.aten_reduce_plan` -- the host half of qs_mean_strided -- against Tensor.mean on the CPU: dense tensors in a random
    dim order (transposes, permutes, channels_last(_3d) among them), every reducible dim, float32 / bfloat16 / float16; the plan is
    executed here exactly as the kernel executes it (one output at a time, the order it names)"""
    import itertools
    import random

    from qsparse_amd.util import _dense_any_order, aten_reduce_plan, split_view_plan

    def execute(x, d):
        n, s0, kept, order, split_dim, split = aten_reduce_plan(list(x.shape), list(x.stride()), d)
        mem = torch.as_strided(x, (x.untyped_storage().nbytes() // x.element_size(),), (1,), 0).float().numpy()
        out_shape = [1 if i == d else s for i, s in enumerate(x.shape)]
        out = np.zeros(int(np.prod(out_shape)), np.float32)
        for coords in itertools.product(*[range(k[0]) for k in kept]):
            at = x.storage_offset() + sum(c * k[1] for c, k in zip(coords, kept))
            v = mem[at + np.arange(n) * s0][:, None]
            r = vector_inner(v) if order == 0 else (multi_row(v) if order == 2 and coords[split_dim] < split else row_sum(v))
            out[sum(c * k[2] for c, k in zip(coords, kept))] = r[0] / np.float32(n)
        return torch.from_numpy(out).view(out_shape).to(x.dtype), order

    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        rng, g = random.Random(0), torch.Generator().manual_seed(0)
        seen, differs, views = set(), 0, 0
        for it in range(260):
            nd = rng.choice([1, 2, 3, 4, 4, 5])
            shape = [rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 33, 40]) for _ in range(nd)]
            while int(np.prod(shape)) > 12000:
                shape[rng.randrange(nd)] = 2
            perm = list(range(nd))
            rng.shuffle(perm)
            dtype = rng.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
            x = (torch.randn([shape[p] for p in perm], generator=g) * 3).abs().to(dtype).permute([perm.index(i) for i in range(nd)])
            assert _dense_any_order(x) and list(x.shape) == shape
            reducible = [i for i in range(nd) if shape[i] > 1]
            if not reducible:
                continue
            d = rng.choice(reducible)
            want = x.mean(d, keepdim=True)
            got, order = execute(x, d)
            assert want.is_contiguous() and torch.equal(want, got), (shape, x.stride(), d, dtype)
            seen.add(order)
            # the same plan as a contiguous [pre, n, post] tensor in memory order with a cascade PREFIX (qs_mean_dim_split), where it
            # has that form: executed here as the stage kernels execute it
            view = split_view_plan(list(x.shape), list(x.stride()), d, aten_reduce_plan(list(x.shape), list(x.stride()), d))
            if view is not None:
                perm, pre, n, post, mr_cols = view
                mem = x.permute(perm)
                assert mem.is_contiguous() and mem.numel() == pre * n * post and 0 <= mr_cols <= post
                v = mem.float().reshape(pre, n, post).numpy()
                res = np.empty((pre, post), np.float32)
                for p in range(pre):
                    if mr_cols:
                        res[p, :mr_cols] = multi_row(v[p][:, :mr_cols]) / np.float32(n)
                    if mr_cols < post:
                        res[p, mr_cols:] = row_sum(v[p][:, mr_cols:]) / np.float32(n)
                shape_mem = list(mem.shape)
                shape_mem[perm.index(d)] = 1
                back = torch.from_numpy(res).view(shape_mem).permute([perm.index(i) for i in range(nd)]).contiguous().to(dtype)
                assert torch.equal(want, back), (shape, x.stride(), d, dtype, view)
                views += 1
            differs += int(not torch.equal(want, x.contiguous().mean(d, keepdim=True)))
        assert seen == {0, 1, 2} and differs >= 20       # (the contiguous copy's order is a different one for many of them)
        assert views >= 15, views
        assert not _dense_any_order(torch.zeros(4, 6)[:, ::2]) and not _dense_any_order(torch.zeros(4, 1).expand(4, 3))
    finally:
        torch.set_num_threads(threads)


def test_fixtures_record_the_torch_they_were_generated_with_and_it_is_the_pinned_one():
    """the golden fixtures carry `meta["torch"]` / `meta["intra_op_threads"]`, the package pins the same minor
    version (`PINNED_TORCH`, stated in include/qsparse_hip.h and csrc/qs_reduce.h) and warns once under another"""
    import glob
    import json
    import warnings

    import qsparse_amd as qs
    from qsparse_amd import util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tests", "golden", "*.npz")))
    assert len(files) >= 16
    for f in files:
        meta = json.loads(str(np.load(f)["meta"]))
        assert ".".join(meta["torch"].split("+")[0].split(".")[:2]) == qs.PINNED_TORCH, f
        assert meta["intra_op_threads"] == 1, f
    assert f"torch {qs.PINNED_TORCH}" in open(os.path.join(root, "include", "qsparse_hip.h")).read()
    assert f"torch {qs.PINNED_TORCH}" in open(os.path.join(root, "qsparse_amd", "csrc", "qs_reduce.h")).read()
    util._torch_pin_warned = False
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert util.check_torch_pin("2.12.1+rocm9") is False and util.check_torch_pin("2.12.1+rocm9") is False
        assert len(w) == 1 and "summation order" in str(w[0].message)
        assert util.check_torch_pin(qs.PINNED_TORCH + ".0") is True
    finally:
        util._torch_pin_warned = False
