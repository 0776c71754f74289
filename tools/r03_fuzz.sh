#!/bin/bash
# Round-3 fuzz campaign on the GPU box (tests/big_fuzz.py against the oracle); prints one summary line per mode.
# usage (under gpurun): bash tools/r03_fuzz.sh > gpurun_out/r03_fuzz.txt
run() {  # label, time limit, env...
  local label="$1" limit="$2"; shift 2
  echo "# $label"
  env "$@" timeout -k 10 "$limit" python tests/big_fuzz.py 2>&1 | grep -E "^seed|MISMATCH|Traceback|Error" | tail -3
}
run "default kernel choice, seeds 51000:16" 280 MRX_FUZZ_SEEDS=51000:16
run "MRX_FUZZ_NFA=1 (NFA route and bitset kernels forced), seeds 52000:12" 280 MRX_FUZZ_NFA=1 MRX_FUZZ_SEEDS=52000:12
run "MRX_FUZZ_EXTRA=1 (arbitrary bytes, fixed-pitch layouts, start, count), seeds 53000:8" 280 MRX_FUZZ_EXTRA=1 MRX_FUZZ_SEEDS=53000:8
run "MRX_LONG_TEXT_MODE=1 (pieces / wavefront-per-text forms), seeds 54000:3" 280 MRX_LONG_TEXT_MODE=1 MRX_FUZZ_SEEDS=54000:3
run "MRX_FUZZ_GROUPS=1 (capture groups), seeds 55000:30" 200 MRX_FUZZ_GROUPS=1 MRX_FUZZ_SEEDS=55000:30
run "MRX_FUZZ_GEN=2 (second generator), seeds 56000:8" 200 MRX_FUZZ_GEN=2 MRX_FUZZ_SEEDS=56000:8
run "mrx_debug_multiwalk:2 (no multi-walk / marks / fixed-length forms), seeds 57000:6" 150 MRX_FUZZ_DEBUG=mrx_debug_multiwalk:2 MRX_FUZZ_SEEDS=57000:6
