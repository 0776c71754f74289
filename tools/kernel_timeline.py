#!/usr/bin/env python3
"""Kernel sequence of a rocprofv3 --kernel-trace run (rocpd database): the launches around the LAST occurrences of a kernel.
usage: python tools/kernel_timeline.py <results.db> <kernel-substring> [before] [after] [occurrences]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 14
after = int(sys.argv[4]) if len(sys.argv) > 4 else 3
occ = int(sys.argv[5]) if len(sys.argv) > 5 else 1
rows = list(db.execute("select name,start,end,grid_x,workgroup_x,vgpr_count,lds_size from kernels order by start"))


def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+|at::\w+[^<(]*)", n)
    return m.group(1) if m else n[:40]


idx = [i for i, r in enumerate(rows) if want in r[0]]
for i in idx[-occ:]:
    lo = max(0, i - before)
    t0 = rows[lo][1]
    for r in rows[lo:i + after + 1]:
        print("%-44s %8.1f us  at %8.1f  grid %s wg %s vgpr %s lds %s" % (short(r[0]), (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e3, r[3], r[4], r[5], r[6]))
    print()
