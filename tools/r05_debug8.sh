#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export FRIDO_TUNE_CACHE=/tmp/x.json
for k in 1 2; do
  echo "== round $k"
  (python tools/debug_tworank.py 0 2 5 2>&1 | grep shard | cut -c1-70) & (python tools/debug_tworank.py 2 4 5 2>&1 | grep shard | cut -c1-70) & wait
done > $OUT/r05_debug_tworank5.txt 2>&1
cat $OUT/r05_debug_tworank5.txt
