#!/bin/bash
# config 5 findall with the decode kernel's grid capped (partial output lines per wavefront in flight vs L2)
for g in 0 256 512 1024 2048; do
  echo "MRX_DECODE_GRID=$g"
  MRX_DECODE_GRID=$g python tools/bench_configs.py only c5 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-400
done
