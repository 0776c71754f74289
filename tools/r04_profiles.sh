#!/bin/bash
# Round-4 profile set: bench line (serial headline), rocprofv3 kernel stats and PMC passes of bench.py, all configs,
# ragged / sub rows, the reference's benchmark list.  Outputs under gpurun_out/r04/ (tools/summarize_profiles.py turns
# them into the tracked files under profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err || exit 1
echo "bench done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-overlap-leg > $O/bench_line_under_rocprofv3.json 2> $O/prof.err) || exit 1
echo "kernel stats done"
bash tools/pmc_pass.sh r04/pmc || exit 1
python tools/bench_configs.py > $O/cfg.jsonl 2> $O/cfg.err || exit 1
python tools/bench_configs.py sub >> $O/cfg.jsonl 2>> $O/cfg.err || exit 1
python tools/bench_configs.py ragged >> $O/cfg.jsonl 2>> $O/cfg.err || exit 1
echo "configs done"
python tools/bench_suite.py > $O/suite.jsonl 2> $O/suite.err || exit 1
echo "suite done"
MRX_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --no-cpu-baseline --texts 262144 --c3-texts 262144 > $O/bench_gpus2_shared.json 2> $O/gpus2.err
echo done
