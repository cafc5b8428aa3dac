#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash tools/ab_env_tuned.sh FRIDO_GN_CONV 0 1 2>&1 | tee $OUT/r04_gnconv_ab_v3.txt
rm -f $OUT/e2e_error.json
(time python -m pytest tests -m gpu -q -s --durations=12 > $OUT/r04_gpu_tests_2.log 2>&1); tail -25 $OUT/r04_gpu_tests_2.log
cp $OUT/e2e_error.json $OUT/r04_e2e_error_2.json 2>/dev/null
