#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv compactly: calls, average us, share, short kernel name
usage: kstats.py file.csv [substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = r["Name"]
    if sub and sub not in name:
        continue
    short = name.split("(")[0].replace("void ", "")[:100]
    print(f"{int(r['Calls']):7d} calls {float(r['AverageNs']) / 1e3:9.1f} us avg {float(r['Percentage']):6.2f} %  {short}")
