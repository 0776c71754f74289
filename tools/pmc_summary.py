#!/usr/bin/env python3
"""Per-kernel averages of the counters tools/pmc_cmd.sh collected: python tools/pmc_summary.py <outdir-under-gpurun_out> [kernel-substring ...]"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", sys.argv[1])
want = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want and not any(w in k for w in want):
            continue
        m = re.search(r"k_\w+(<[^>]*>)?", k)
        agg[m.group(0) if m else k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())}, "launches", max(len(x) for x in v.values()))
