#!/bin/bash
set -o pipefail
out=gpurun_out/r02_fused_dbg
mkdir -p $out
run() { # name, env...
  name=$1; shift
  for rep in 1 2; do
    env "$@" timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_$name.$rep.json 2> $out/bench_$name.$rep.err || { echo "bench $name failed"; tail -5 $out/bench_$name.$rep.err; return 1; }
    python - "$name" $out/bench_$name.$rep.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
print("%-22s ms/step %.4f  value %.0f GB/s  kernel %s %.4f ms  frac %.3f" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
if d["ms_per_step"] > 5: sys.exit(3)
if sys.argv[1] != "three" and d["roofline"]["kernel"] != "k_stream_findall_fused": sys.exit(4)   # (the leg did not run the fused form)
PY
    [ $? -eq 0 ] || return 1
  done
}
run three MRX_FUSED=0 || exit 1
for st in 0 2 4 6 8; do run stagger$st MRX_FUSED=1 MRX_FUSED_DEBUG=$((st*256)) || exit 1; done
for st in 0 4; do run bpc3_stagger$st MRX_FUSED=1 MRX_FUSED_BPC=3 MRX_FUSED_DEBUG=$((st*256)) || exit 1; done
for st in 0 4; do run dbg7_stagger$st MRX_FUSED_DEBUG=$((st*256+7)) || exit 1; done
