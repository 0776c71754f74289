#!/bin/bash
# k_subs_wave with phases left out (MRX_SUBS_DEBUG, outputs wrong on purpose): which phase the time is in
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for d in 0 1 2 3 4 8 16 24 28 31; do
  echo "MRX_SUBS_DEBUG=$d"
  MRX_SUBS_DEBUG=$d python3 tools/bench_configs.py sub 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('   ', r['config'], r['sub_ms'])"
done
