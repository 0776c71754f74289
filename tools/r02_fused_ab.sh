#!/bin/bash
# Fused findall (one launch) against the three-launch form: parity tests, then bench.py A/B and the
# launch-shape knob.  Output under gpurun_out/r02_fused/.
set -o pipefail
out=gpurun_out/r02_fused
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" > $out/pytest_fused.log 2>&1
rc=$?
tail -5 $out/pytest_fused.log
if [ $rc -ne 0 ]; then echo "fused parity failed rc=$rc"; exit $rc; fi
run() { # name, env...
  name=$1; shift
  for rep in 1 2; do
    env "$@" timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_$name.$rep.json 2> $out/bench_$name.$rep.err || { echo "bench $name failed"; tail -5 $out/bench_$name.$rep.err; return 1; }
    python - "$name" $out/bench_$name.$rep.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
print("%-14s ms/step %.4f  value %.0f GB/s  kernel %s %.4f ms  frac %.3f" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
if d["ms_per_step"] > 5: sys.exit(3)
if sys.argv[1] != "three" and d["roofline"]["kernel"] != "k_stream_findall_fused": sys.exit(4)   # (the leg did not run the fused form)
PY
    [ $? -eq 0 ] || return 1
  done
}
run fused MRX_FUSED=1 && run three MRX_FUSED=0 || exit 1
for dbg in 7 2 1 4; do run dbg$dbg MRX_FUSED=1 MRX_FUSED_DEBUG=$dbg || exit 1; done
for bpc in 3 8; do run bpc${bpc} MRX_FUSED=1 MRX_FUSED_BPC=$bpc || exit 1; done
