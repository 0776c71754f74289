#!/bin/bash
# A/B of the scan forms on the bench workload: 4-bit columns vs code columns, three launches vs fused.
# usage: tools/r03_ab.sh OUTDIR [steps]
out=${1:-gpurun_out/r03_ab}; steps=${2:-50}
mkdir -p "$out"
run() { name=$1; shift; env "$@" python bench.py --steps "$steps" --warmup 5 --no-cpu-baseline > "$out/$name.json" 2> "$out/$name.err"; }
for rep in 1 2; do
  run cols4_$rep MRX_NO_CODE_COLUMNS=1
  run code_$rep MRX_X=0
  run cols4_fused_$rep MRX_NO_CODE_COLUMNS=1 MRX_FUSED=1
  run code_fused_$rep MRX_FUSED=1
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-18s ms/step %.4f  kernel %s %.4f ms  overlapped %.4f  count %.0f search %.0f" % (
            os.path.basename(f)[:-5], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"],
            d.get("two_streams_overlapped", {}).get("ms_per_step", 0), d["other_ops"]["count_GBps"], d["other_ops"]["search_GBps"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
