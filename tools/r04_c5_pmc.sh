#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04/c5_pmc
rm -rf $OUT; mkdir -p $OUT
python3 $R/tools/r04_c5_probe.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/r04_c5_probe.py > $OUT/trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/tools/r04_c5_probe.py > $OUT/set$i.log 2>&1
  echo "set$i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("/root/repo/gpurun_out/r04/c5_pmc/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stream_findall" in r["Name"] or "k_decode" in r["Name"] or "k_scan" in r["Name"]:
            print(r["Name"][:70], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/root/repo/gpurun_out/r04/c5_pmc/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_stream_findall" in k or "k_decode" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
