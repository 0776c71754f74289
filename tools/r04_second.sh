#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_bits" > $O/t_bits.log 2>&1 || { tail -30 $O/t_bits.log; exit 1; }
tail -2 $O/t_bits.log
timeout -k 5 300 python tools/r04_ablate.py 2>&1 | tee $O/ablate.jsonl | tail -20
