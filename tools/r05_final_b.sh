#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
bash tools/final_run.sh r05 full
FIND_SAT_REPEAT=2 python tools/find_saturation.py 16 200 2>&1 | grep -E "repeat|sha|pass 1|stage 1" | cut -c1-200 > $OUT/r05_find_saturation.txt; cat $OUT/r05_find_saturation.txt
