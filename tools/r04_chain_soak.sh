#!/bin/bash
# Round 4: generated chains with groups (tests/test_gpu_parity.py's generator) under other seeds, and the groups mode of the fuzz.
# Every step prints as it goes and has its own time limit; a step that fails or is killed ends the script.
set -e
for seed in ${SEEDS:-11 12 13 14 15 16}; do
  echo "# generated chains, seed $seed, ${N:-400} patterns"
  MRX_CHAIN_FUZZ_SEED=$seed MRX_CHAIN_FUZZ_N=${N:-400} timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k generated_chains --durations=1 2>&1 | grep -v "^$"
done
echo "# MRX_FUZZ_GROUPS=1 (capture groups), seeds ${S:-195000}:30"
env PYTHONUNBUFFERED=1 MRX_FUZZ_GROUPS=1 MRX_FUZZ_SEEDS=${S:-195000}:30 timeout -k 10 280 python tests/big_fuzz.py 2>&1 | grep --line-buffered -E "^seed|MISMATCH|Traceback|Error"
