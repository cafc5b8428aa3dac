#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export FRIDO_TUNE_CACHE=/tmp/x.json
for L in libfrido_ds_nokdb.so libfrido_ds_nocount.so; do
  echo "== $L"
  (FRIDO_LIB=$R/frido_amd/$L python tools/debug_tworank.py 0 2 3 2>&1 | grep shard | cut -c1-70) & (FRIDO_LIB=$R/frido_amd/$L python tools/debug_tworank.py 2 4 3 2>&1 | grep shard | cut -c1-70) & wait
done > $OUT/r05_debug_tworank4.txt 2>&1
cat $OUT/r05_debug_tworank4.txt
