#!/bin/bash
# Round-3 profile set: bench line (serial headline), rocprofv3 kernel stats and PMC passes of bench.py, all configs,
# ragged / sub / bitset rows, the multi-walk probe.  Outputs under gpurun_out/r03/ (tools/summarize_profiles.py and
# tools/r03_collect.py turn them into the tracked files under profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err || exit 1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-overlap-leg > $O/bench_line_under_rocprofv3.json 2> $O/prof.err) || exit 1
bash tools/pmc_pass.sh r03/pmc || exit 1
python tools/bench_configs.py > $O/cfg.jsonl 2> $O/cfg.err || exit 1
python tools/bench_configs.py sub >> $O/cfg.jsonl 2>> $O/cfg.err || exit 1
python tools/bench_configs.py ragged >> $O/cfg.jsonl 2>> $O/cfg.err || exit 1
python tools/r03_mwalk_probe.py > $O/mwalk_probe.jsonl 2> $O/mwalk.err || exit 1
MRX_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --no-cpu-baseline --texts 262144 --c3-texts 262144 > $O/bench_gpus2_shared.json 2> $O/gpus2.err
echo done
