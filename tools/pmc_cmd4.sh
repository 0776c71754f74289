#!/bin/bash
# as tools/pmc_cmd.sh, plus FETCH_SIZE and WRITE_SIZE in passes of their own:
#   gpurun -- bash tools/pmc_cmd4.sh <outdir-under-gpurun_out> tools/bench_suite.py complex_number
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
shift
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/"$@" > $OUT/set$i.log 2>&1
  echo "set$i rc=$?"
done
