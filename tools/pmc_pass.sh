#!/bin/bash
# Collect PMC counters for the bench kernels, one rocprofv3 pass per counter set
# (counter passes must not be combined with sys/hip tracing on this pool).
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc}
shift
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/set$i.log 2>&1
  echo "set$i ($set) rc=$?"
done
