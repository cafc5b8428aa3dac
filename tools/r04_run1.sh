#!/bin/bash
# r04 first GPU pass: full GPU suite, headline bench (rewrites the pinned tile cache for this library), single-plane table
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
rm -f $OUT/e2e_error.json
(time python -m pytest tests -m gpu -q -s --durations=15 -x > $OUT/r04_gpu_tests_1.log 2>&1); tail -5 $OUT/r04_gpu_tests_1.log
cp $OUT/e2e_error.json $OUT/r04_e2e_error_1.json 2>/dev/null
rm -f profiles/tune_cache.json
python bench.py --retune --steps 3 --warmup 1 > $OUT/r04_bench_line_1.json 2> $OUT/r04_bench_line_1.err; tail -c 1500 $OUT/r04_bench_line_1.json
cp profiles/tune_cache.json $OUT/tune_cache.json
python tools/x3_single_plane_table.py > $OUT/r04_x3_single_plane_table.txt 2>&1; cat $OUT/r04_x3_single_plane_table.txt
