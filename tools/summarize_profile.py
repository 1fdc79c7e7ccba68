#!/usr/bin/env python3
"""Per-kernel summary of the steady-state steps of a rocprofv3 --kernel-trace database (rocpd sqlite).
usage: summarize_profile.py results.db [marker-substring] [steps]   (marker: a kernel launched once per step)"""
import collections
import sqlite3
import sys

db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "max_pool_backward"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
sel = rows[marks[-steps - 1]:marks[-1]]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    a = agg[n]
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
qs_t = sum(a[1] for n, a in agg.items() if "qs::" in n)
qs_n = sum(a[0] for n, a in agg.items() if "qs::" in n)
print(f"steps {steps}: wall/step {(sel[-1][2] - sel[0][1]) / 1e6 / steps:.2f} ms, kernel busy/step {tot / steps / 1e3:.2f} ms, "
      f"launches/step {len(sel) / steps:.0f}; qs kernels {qs_t / steps / 1e3:.2f} ms/step in {qs_n / steps:.0f} launches")
only_qs = "--qs" in sys.argv
for n, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    if only_qs and "qs::" not in n:
        continue
    print(f"{us / steps:8.1f} us/step {cnt / steps:6.1f} calls {us / cnt:7.1f} us/call  {n[:110]}")
