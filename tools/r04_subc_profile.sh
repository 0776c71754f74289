#!/bin/bash
# Round 4: kernel timeline of the reference's sub_group_word_swap row (sub with \1..\9 on a chain: spans + k_subc_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/subc_prof; cd /tmp; export TMPDIR=/tmp
MRX_SUITE_OPS=sub rocprofv3 --kernel-trace -d $O/subc_prof -o subc -- python3 $R/tools/bench_suite.py sub_group_word > $O/subc_prof.log 2>&1
echo "rc=$?"
python3 $R/tools/kernel_timeline.py $O/subc_prof/subc_results.db k_subc_emit 12 1 1
