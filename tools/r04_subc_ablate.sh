#!/bin/bash
# Round 4: where the chain sub's kernels spend their time (MRX_SUBC_DEBUG bits, results are wrong: k_subc_sizes 1 no walk,
# 2 no bitmaps, 4 no per-match store; k_subc_emit 8 no gap copies, 32 no replacement bytes, 64 no stores)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for dbg in ${DBGS:-0 8 32 64 104}; do
  rm -rf $O/subc_ab
  MRX_SUBC_DEBUG=$dbg MRX_SUITE_OPS=sub rocprofv3 --kernel-trace -d $O/subc_ab -o subc -- python3 $R/tools/bench_suite.py sub_group_word > $O/subc_ab.log 2>&1
  echo "dbg=$dbg $(python3 $R/tools/kernel_timeline.py $O/subc_ab/subc_results.db ${KERNEL:-k_subc_emit} 0 0 1 | head -1)"
done
