#!/bin/bash
# the reference's sub rows: kernel sequence per case (tools/bench_suite.py, MRX_SUITE_OPS=sub)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/suite_sub_prof; cd /tmp; export TMPDIR=/tmp
MRX_SUITE_OPS=sub rocprofv3 --kernel-trace -d $O/suite_sub_prof -o s -- python3 $R/tools/bench_suite.py > $O/suite_sub_prof.log 2>&1
echo "rc=$?"
grep '^{' $O/suite_sub_prof.log | cut -c1-220
python3 - <<'PY'
import sqlite3, re
db = sqlite3.connect("/root/repo/gpurun_out/r04/suite_sub_prof/s_results.db")
rows = list(db.execute("select name,start,end,grid_x from kernels order by start"))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+|at::\w+[^<(]*)", n)
    return m.group(1) if m else n[:40]
# a "call" ends with a D2H copy burst; print the sequence between consecutive k_pitch/k_max_len markers for the last call of each case
marks = [i for i, r in enumerate(rows) if "k_max_len" in r[0] or "k_csr_stats" in r[0]]
seen = 0
for a, b in zip(marks, marks[1:] + [len(rows)]):
    seen += 1
    if seen % 11 != 0: continue   # one call per case (WARM 6 + REPS 5)
    t0 = rows[a][1]
    print("--- call", seen)
    for r in rows[a:b]:
        print("%-44s %8.1f us at %8.1f" % (short(r[0]), (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e3))
PY
