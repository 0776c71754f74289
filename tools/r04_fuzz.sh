#!/bin/bash
# Round-4 fuzz campaign on the GPU box (tests/big_fuzz.py against the oracle); prints one summary line per mode.
# usage (under gpurun, two calls of at most 20 minutes): bash tools/r04_fuzz.sh 1|2 > gpurun_out/r04_fuzz_N.txt
run() {  # label, time limit, env...
  local label="$1" limit="$2"; shift 2
  echo "# $label"
  env PYTHONUNBUFFERED=1 "$@" timeout -k 10 "$limit" python tests/big_fuzz.py 2>&1 | grep -E "^seed|MISMATCH|Traceback|Error" | tail -1
}
if [ "${1:-1}" = "1" ]; then
run "default kernel choice, seeds ${S1:-171000}:16" 290 MRX_FUZZ_SEEDS=${S1:-171000}:16
run "MRX_FUZZ_NFA=1 (NFA route and bitset kernels forced), seeds ${S2:-172000}:12" 290 MRX_FUZZ_NFA=1 MRX_FUZZ_SEEDS=${S2:-172000}:12
run "MRX_FUZZ_EXTRA=1 (arbitrary bytes, fixed-pitch layouts, start, count), seeds ${S3:-173000}:8" 290 MRX_FUZZ_EXTRA=1 MRX_FUZZ_SEEDS=${S3:-173000}:8
else
run "MRX_LONG_TEXT_MODE=1 (pieces / wavefront-per-text forms), seeds 174000:3" 290 MRX_LONG_TEXT_MODE=1 MRX_FUZZ_SEEDS=174000:3
run "MRX_FUZZ_GROUPS=1 (capture groups), seeds 175000:30" 220 MRX_FUZZ_GROUPS=1 MRX_FUZZ_SEEDS=175000:30
run "MRX_FUZZ_GEN=2 (second generator), seeds 176000:8" 220 MRX_FUZZ_GEN=2 MRX_FUZZ_SEEDS=176000:8
run "mrx_debug_multiwalk:2 (no multi-walk / marks / fixed-length forms), seeds 177000:6" 160 MRX_FUZZ_DEBUG=mrx_debug_multiwalk:2 MRX_FUZZ_SEEDS=177000:6
run "mrx_debug_multiwalk:3 (multi-walk without the packed starts), seeds 178000:4" 120 MRX_FUZZ_DEBUG=mrx_debug_multiwalk:3 MRX_FUZZ_SEEDS=178000:4
fi
