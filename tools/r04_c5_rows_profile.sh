#!/bin/bash
# config 5 by event rows: kernel sequence and HBM counters of tools/bench_configs.py only "c5 (x"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/c5rows_prof $O/c5rows_pmc; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/c5rows_prof -o c5 -- python3 $R/tools/bench_configs.py only "c5 (x" > $O/c5rows_prof.log 2>&1
echo "rc=$?"
python3 $R/tools/kernel_timeline.py $O/c5rows_prof/c5_results.db k_decode_rows 6 1 2
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/c5rows_pmc/set$i -- python3 $R/tools/bench_configs.py only "c5 (x" > $O/c5rows_pmc_set$i.log 2>&1
  echo "set$i rc=$?"
done
python3 $R/tools/pmc_summary.py r04/c5rows_pmc k_decode_rows "k_stream_findall<5"
