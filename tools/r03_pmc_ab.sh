#!/bin/bash
# PMC passes 1-2 (instruction counts, LDS conflicts) of the bench kernels, code columns on / off
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for variant in code cols4; do
  OUT=$R/gpurun_out/r03_pmc_$variant
  mkdir -p $OUT
  if [ $variant = cols4 ]; then export MRX_NO_CODE_COLUMNS=1; else unset MRX_NO_CODE_COLUMNS; fi
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 3 --warmup 1 --settle 4 --no-cpu-baseline > $OUT/set$i.log 2>&1
    echo "$variant set$i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for variant in ("code", "cols4"):
    for kn in ("k_stream_findall<0", "k_stream_findall<1", "k_decode"):
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(R, "gpurun_out", "r03_pmc_" + variant, "set*", "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if kn in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        s = {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}
        print(variant, kn, json.dumps(s))
PY
