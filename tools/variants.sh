#!/bin/bash
# Build variants of libmrx_hip.so with extra -D flags and time the bench step with each.
#   VARIANTS="base: b16:-DMRX_DECODE_BATCH=16" tools/variants.sh build     (in the container)
#   gpurun -- tools/variants.sh run                                          (on the box)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/tools/variant_libs
if [ "$1" = build ]; then
  rm -rf $OUT; mkdir -p $OUT
  for v in $VARIANTS; do
    name=${v%%:*}; flags=${v#*:}; flags=${flags//,/ }
    (cd $R/mojo_regex_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared \
      -Wno-unused-function $flags -o $OUT/libmrx_hip_$name.so mrx_frontend.cpp mrx_analysis.cpp \
      mrx_dfa_build.cpp mrx_nfa_build.cpp mrx_plan.cpp mrx_kernels.hip) &
  done
  wait
  ls -la $OUT
elif [ "$1" = cmd ]; then
  # tools/variants.sh cmd <command...>: run the command once per variant with MRX_LIB set
  shift
  for lib in $OUT/libmrx_hip_*.so; do
    echo "== $(basename $lib)"
    MRX_LIB=$lib "$@" 2>/dev/null | tail -3
  done
else
  for lib in $OUT/libmrx_hip_*.so; do
    line=$(MRX_LIB=$lib python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1)
    echo "$(basename $lib) $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel_ms=%.4f step_ms=%.4f" % (d["roofline"]["kernel_ms"], d["ms_per_step"]))')"
  done
fi
