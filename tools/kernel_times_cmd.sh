#!/bin/bash
# Per-kernel average durations (rocprofv3 kernel trace) of an arbitrary python command:
#   gpurun -- tools/kernel_times_cmd.sh tools/bench_suite.py sub_char_class
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/ktc
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $R/"$@" > $out/cmd.log 2>&1
tail -2 $out/cmd.log | cut -c1-300
python3 - "$(find $out -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "anonymous" in r["Name"]:
        print("  %-80s calls=%s avg_us=%.1f total_ms=%.2f" % (r["Name"][26:106], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"])/1e6))
PY
