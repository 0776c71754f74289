#!/bin/bash
# Round-2 profile set: bench line (two streams, and one for comparison), rocprofv3 kernel stats and PMC passes of
# bench.py, the reference's benchmark list, all configs, the moved-off-generic table.  Outputs under gpurun_out/r02/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err || exit 1
python bench.py --streams 1 --no-cpu-baseline > $O/bench_line_streams1.json 2>> $O/bench.err || exit 1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_line_under_rocprofv3.json 2> $O/prof.err) || exit 1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $O/bench_line_streams1_under_rocprofv3.json 2>> $O/prof.err) || exit 1
bash tools/pmc_pass.sh r02/pmc --streams 1 || exit 1
python tools/bench_suite.py > $O/suite.jsonl 2> $O/suite.err || exit 1
python tools/bench_configs.py > $O/cfg.jsonl 2> $O/cfg.err || exit 1
python tools/bench_configs.py sub >> $O/cfg.jsonl 2>> $O/cfg.err || exit 1
python tools/bench_configs.py ragged >> $O/cfg.jsonl 2>> $O/cfg.err || exit 1
python tools/bench_moved.py > $O/moved.jsonl 2> $O/moved.err || exit 1
MRX_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --gather --no-cpu-baseline --texts 262144 > $O/bench_gpus2_shared.json 2> $O/gpus2.err
echo done
