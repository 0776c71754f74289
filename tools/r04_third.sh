#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_bits" > $O/t_bits.log 2>&1 || { tail -30 $O/t_bits.log; exit 1; }
tail -2 $O/t_bits.log
MRX_SB_VERBOSE=1 timeout -k 5 200 python tools/r04_trace.py 2>&1 | tee $O/trace.txt | cut -c1-400 | tail -24
echo "--- small LDS variant"
MRX_SB_VERBOSE=1 MRX_LIB=$R/tools/variant_libs/libmrx_sb11264.so timeout -k 5 200 python tools/r04_trace.py 2>&1 | tee $O/trace_small.txt | cut -c1-400 | grep -v "tasks"
