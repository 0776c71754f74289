#!/bin/bash
# PMC counters of an arbitrary python command, one rocprofv3 pass per counter set (no tracing domains besides the kernel trace):
#   gpurun -- bash tools/pmc_cmd.sh <outdir-under-gpurun_out> tools/bench_configs.py sub
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/"$@" > $OUT/set$i.log 2>&1
  echo "set$i rc=$?"
done
