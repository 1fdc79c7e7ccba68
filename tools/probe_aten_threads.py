#!/usr/bin/env python3
"""How much do ATen's own CPU staged means (the reference's `squeeze_tensor_to_shape`, qsparse/util.py:92-99) depend on
the intra-op thread count?  Random activation shapes, NCHW-contiguous and channels_last, 1 thread against 2 / 4 / 8
(CPU only; this is where the "1 thread, equal up to 4" contract in INTEGRATION.md comes from)."""
import random

import torch


def run(channels_last, seed, cases=400):
    rng = random.Random(seed)
    stats, examples, total = {2: 0, 4: 0, 8: 0}, {2: [], 4: [], 8: []}, 0
    for _ in range(cases):
        N = rng.choice([2, 3, 4, 8, 16, 17, 32, 33, 64, 100, 256])
        C = rng.choice([2, 3, 5, 6, 8, 10, 12, 16, 20, 24, 31, 36, 64, 100, 130, 256, 300])
        H, W = rng.choice([(1, 1), (2, 2), (4, 4), (7, 7), (8, 8), (14, 14), (5, 9), (16, 16), (3, 32), (28, 28), (1, 7)])
        if N * C * H * W > 3_000_000:
            continue
        x = torch.randn(N, C, H, W)
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)

        def staged():
            return x.abs().mean(0, keepdim=True).mean(2, keepdim=True).mean(3, keepdim=True)

        torch.set_num_threads(1)
        ref = staged()
        total += 1
        for t in stats:
            torch.set_num_threads(t)
            if not torch.equal(ref, staged()):
                stats[t] += 1
                examples[t].append((N, C, H, W))
    return total, stats, examples


if __name__ == "__main__":
    for cl in (False, True):
        total, stats, examples = run(cl, seed=int(cl))
        print(("channels_last" if cl else "NCHW"), "shapes:", total, "differ from the 1-thread result at", stats)
        for t, ex in examples.items():
            if ex:
                print("   ", t, "threads:", ex[:8])
