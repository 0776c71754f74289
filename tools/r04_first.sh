#!/bin/bash
# round 4, first GPU call: k_stream_bits parity + the headline with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_bits" > $O/t_bits.log 2>&1 || { tail -30 $O/t_bits.log; exit 1; }
tail -3 $O/t_bits.log
timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_bits.json 2> $O/bench_bits.err || { tail -5 $O/bench_bits.err; exit 1; }
MRX_STREAM_BITS=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_three.json 2> $O/bench_three.err || exit 1
python - <<'PY'
import json
for f in ("bench_bits","bench_three"):
    d=json.loads(open("/root/repo/gpurun_out/r04/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d.get("two_streams_overlapped",{}).get("ms_per_step"))
PY
