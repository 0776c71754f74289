#!/bin/bash
# the reference's findall rows under the long-text switches: 0 default, 1 pieces always, 2 never, 3 wavefront kernel for stepper plans
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O; cd $R
for m in 0 1 2 3; do
  MRX_LONG_TEXT_MODE=$m MRX_SUITE_OPS=findall python tools/bench_suite.py > $O/suite_mode$m.jsonl 2> $O/suite_mode$m.err
  echo "mode $m rc=$?"
done
python - <<'PY'
import json
rows = {}
for m in range(4):
    for l in open("/root/repo/gpurun_out/r04/suite_mode%d.jsonl" % m):
        l = l.strip()
        if not l.startswith("{"): continue
        d = json.loads(l)
        if d.get("op") != "findall" or "GBps" not in d: continue
        rows.setdefault(d["case"], {})[m] = (d["GBps"], d.get("kernel"))
for c, v in rows.items():
    print(c, " | ".join("%d: %s %s" % (m, v[m][0], v[m][1]) for m in sorted(v)))
PY
