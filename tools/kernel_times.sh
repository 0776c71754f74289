#!/bin/bash
# Per-kernel average durations of one bench.py run under rocprofv3 (kernel trace only), for every
# library variant built by tools/variants.sh (or the product library when there are none).
#   gpurun -- tools/kernel_times.sh
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
libs=$(ls $R/tools/variant_libs/libmrx_hip_*.so 2>/dev/null)
[ -z "$libs" ] && libs=$R/mojo_regex_amd/libmrx_hip.so
for lib in $libs; do
  out=$R/gpurun_out/kt_$(basename $lib .so)
  rm -rf $out; mkdir -p $out
  MRX_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $out/bench.log 2>&1
  echo "== $(basename $lib)"
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print("  %-60s calls=%s avg_us=%.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
