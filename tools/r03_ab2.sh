#!/bin/bash
# A/B: decode order; PMC of the fused form
out=gpurun_out/r03_ab2; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/$name.json" 2> "$out/$name.err"; }
for rep in 1 2 3; do
  run fwd_$rep MRX_DECODE_REVERSE=0
  run rev_$rep MRX_DECODE_REVERSE=1
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-18s ms/step %.4f  kernel %s %.4f ms  overlapped %.4f" % (
            os.path.basename(f)[:-5], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"],
            d.get("two_streams_overlapped", {}).get("ms_per_step", 0)))
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_pmc_fused; mkdir -p $OUT
export MRX_FUSED=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 3 --warmup 1 --settle 4 --no-cpu-baseline > $OUT/set$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(R, "gpurun_out", "r03_pmc_fused", "set*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "k_stream" in k or "k_decode" in k or "k_fused" in k:
        print(k, json.dumps({c: round(sum(x) / len(x)) for c, x in sorted(v.items())}))
PY
