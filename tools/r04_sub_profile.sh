#!/bin/bash
# where regex.sub's time goes on configs 4 and 2: per-kernel durations of tools/bench_configs.py sub
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/sub_prof -o sub -- python3 $R/tools/bench_configs.py sub > $O/sub_prof.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("/root/repo/gpurun_out/r04/sub_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
