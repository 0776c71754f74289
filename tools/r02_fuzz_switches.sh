#!/bin/bash
# The default fuzz (findall / search / match_first / sub against the oracle) under each kernel-choice switch of
# include/mrx_testing.h, a few seeds each.  Appends one line per switch to gpurun_out/fuzz_switches.log.
set -u
out=gpurun_out/fuzz_switches.log
: > $out
for sw in mrx_debug_dynamic_texts:1 mrx_debug_fused_findall:1 mrx_debug_split_findall:1 mrx_debug_subs_group:16 \
          mrx_debug_subs_group:64 mrx_debug_subs_group:256 mrx_debug_subs_group:0 mrx_debug_litscan_pieces:1 \
          mrx_debug_force_generic:1 mrx_debug_force_generic:2; do
  MRX_FUZZ_DEBUG=$sw MRX_FUZZ_SEEDS=${SEEDS:-44000:3} timeout -k 10 200 python tests/big_fuzz.py > gpurun_out/fuzz_sw.tmp 2>&1
  echo "$sw: $(grep '^seed' gpurun_out/fuzz_sw.tmp | tail -1)" >> $out
  grep MISMATCH gpurun_out/fuzz_sw.tmp | cut -c1-300 | head -5 >> $out
done
cat $out
