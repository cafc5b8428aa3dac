#!/bin/bash
# kernel trace + gap analysis of the bench command only (the last section of tools/run_profiles.sh)
TAG=${1:-r04_x3}; PREC=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktrace -- python $R/bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-bf16-extra > $OUT/${TAG}_bench_under_trace.json 2> /tmp/ktrace.log
cp $(ls /tmp/ktrace/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python $R/tools/gap_analysis.py /tmp/ktrace > $OUT/${TAG}_gap_analysis.json 2> $OUT/${TAG}_gap.err
