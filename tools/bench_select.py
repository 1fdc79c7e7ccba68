#!/usr/bin/env python3
"""Latency of the C-sized select step (qs_pq_select) over channel counts, for several rank-counting /
radix-select crossovers (QS_RANK_SMALL).  Development tool; each setting runs in its own process because the
knob is read once.  Also checks the threshold against torch.sort on the same magnitudes."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import torch

    from qsparse_amd import _hip

    lib = _hip.load()
    dev = "cuda"
    out = {}
    for C in (64, 256, 512, 1024, 2048, 4096):
        torch.manual_seed(C)
        mag = torch.rand(C, device=dev)
        mag[::7] = mag[3]                                     # ties
        stage = torch.rand(C, device=dev).bfloat16()
        mk = torch.ones(C, device=dev, dtype=torch.uint8)
        amax = torch.rand(C, device=dev)
        sc = torch.ones(1, device=dev)
        k = max(int(0.75 * C - 1), 0) + 1

        def select():
            assert lib.qs_pq_select(mag.data_ptr(), stage.data_ptr(), 1, C, 0, 3, 1, k, mk.data_ptr(), amax.data_ptr(), 1, 1, 3,
                                    4, sc.data_ptr(), None, None, None, None, None, None, 1, None, 1, None, None) == 0

        select()
        torch.cuda.synchronize()
        want = mag >= mag.sort()[0][k]
        assert torch.equal(mk.bool(), want), f"mask mismatch at C={C}"
        for _ in range(5):
            select()
        evs = []
        for _ in range(50):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            select()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        out[C] = round(ts[len(ts) // 2] * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one()
    else:
        for rs in ("0", "256", "512", "1024", "2048"):
            r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, QS_RANK_SMALL=rs),
                               capture_output=True, text=True)
            print(f"QS_RANK_SMALL={rs:5s} us per call:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:],
                  flush=True)
