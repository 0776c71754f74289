#!/bin/bash
# ragged CSR findall: kernel sequence of tools/bench_configs.py ragged
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/ragged_prof; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/ragged_prof -o r -- python3 $R/tools/bench_configs.py ragged > $O/ragged_prof.log 2>&1
echo "rc=$?"
python3 $R/tools/kernel_timeline.py $O/ragged_prof/r_results.db "k_decode" 8 2 60 > $O/ragged_timeline.txt
python3 - <<'PY'
blocks = open("/root/repo/gpurun_out/r04/ragged_timeline.txt").read().strip().split("\n\n")
print(len(blocks))
for b in (blocks[2], blocks[-1]):
    print(b, "\n")
PY
