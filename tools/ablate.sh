#!/bin/bash
# Ablation of the streaming scan kernel on the real bench workload: builds variants of
# libmrx_hip.so with pieces of k_stream_findall switched off (MRX_ABLATE, see mrx_kernels.hip)
# and prints the scan kernel's time for each.  Variant results are wrong by design; only the
# timings mean anything.  Build here (cross-compile), run on the GPU box:
#   tools/ablate.sh build            (in the container)
#   gpurun -- tools/ablate.sh run    (on the box)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/tools/ablate_libs
VARIANTS="0 1 2 3 4 5 7 16 18 23"
if [ "$1" = build ]; then
  mkdir -p $OUT
  for v in $VARIANTS; do
    (cd $R/mojo_regex_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared \
      -Wno-unused-function -DMRX_ABLATE=$v -o $OUT/libmrx_hip_$v.so mrx_frontend.cpp mrx_analysis.cpp \
      mrx_dfa_build.cpp mrx_nfa_build.cpp mrx_plan.cpp mrx_kernels.hip) &
  done
  wait
  ls -la $OUT
else
  for v in $VARIANTS; do
    line=$(MRX_LIB=$OUT/libmrx_hip_$v.so python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1)
    echo "ablate=$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel_ms=%.4f step_ms=%.4f" % (d["roofline"]["kernel_ms"], d["ms_per_step"]))')"
  done
fi
