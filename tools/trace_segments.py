#!/usr/bin/env python3
"""Per-segment kernel times from a rocprofv3 --kernel-trace CSV: a segment is a run of
library kernels (k_*) between two foreign (torch) kernels.  Used with tools/bench_configs.py
to see which kernel bounds each config."""
import collections
import csv
import glob
import re
import sys


def main(root):
    f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    segs, cur = [], collections.OrderedDict()
    for r in rows:
        name = r["Kernel_Name"]
        m = re.search(r"\bk_[a-z_]+(<[^>]*>)?", name)
        short = m.group(0) if m else ""
        if not m:
            if cur:
                segs.append(cur)
                cur = collections.OrderedDict()
            continue
        cur.setdefault(short, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if cur:
        segs.append(cur)
    for i, s in enumerate(segs):
        if sum(len(v) for v in s.values()) < 1:
            continue
        print("segment %d" % i)
        for k, v in s.items():
            print("  %-60s n=%3d avg=%9.1f us min=%9.1f us" % (k[:60], len(v), sum(v) / len(v), min(v)))


if __name__ == "__main__":
    main(sys.argv[1])
