#!/bin/bash
# where regex.sub's time goes on configs 4 and 2: the kernel sequence of tools/bench_configs.py sub
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/sub_prof; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/sub_prof -o sub -- python3 $R/tools/bench_configs.py sub > $O/sub_prof.log 2>&1
echo "rc=$?"
python3 $R/tools/kernel_timeline.py $O/sub_prof/sub_results.db k_subs_wave 16 3 21 > $O/sub_timeline.txt
python3 - <<'PY'
import re
blocks = open("/root/repo/gpurun_out/r04/sub_timeline.txt").read().strip().split("\n\n")
for b in (blocks[3], blocks[10], blocks[17]):
    print(b, "\n")
PY
