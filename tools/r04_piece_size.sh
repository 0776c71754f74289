#!/bin/bash
# pieces on the multi-walk kernel: piece size (MRX_PIECE_C) against the default on the suite's few-long-text rows
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in 0 256 512 1024; do
  echo "MRX_PIECE_C=$c"
  for row in complex_number simple_phone toll_free_simple dual_quantifiers alternation_quantifiers optimize_phone flexible_datetime dense_quantifiers; do
    if [ $c = 0 ]; then python3 tools/bench_suite.py $row 2>/dev/null; else MRX_PIECE_C=$c python3 tools/bench_suite.py $row 2>/dev/null; fi
  done | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)
        if r['op'] == 'findall': print('   ', r['case'], r['kernel'], r['ms'], r['GBps'])"
done
