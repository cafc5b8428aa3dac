#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
bash tools/final_run.sh r05 full
python tools/debug_status.py 16 200 2>&1 | grep -v amdgpu.ids > $OUT/r05_status_walk_b16_ddim200.txt; tail -12 $OUT/r05_status_walk_b16_ddim200.txt
