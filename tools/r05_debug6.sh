#!/bin/bash
# which switch makes the sampler's result independent of a second process sharing the GPU?  (static tiles; alone: z sha 17ddb5000af0 / 9fa842b506b0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export FRIDO_TUNE_CACHE=/tmp/x.json
for e in "X=1" "FRIDO_FLASH_DSPLIT=0" "FRIDO_ATTN_FLASH=0" "FRIDO_LN_IN_ATTN=0" "FRIDO_GN_FUSED=0" "FRIDO_GN_EPI_STATS=0" "FRIDO_GN_CONV=0" "FRIDO_ATTN_SKIP_DEAD_STREAM=0" "FRIDO_UP2_PHASES=0" "FRIDO_CHAIN_FF=0"; do
  echo "== $e"
  (env $e python tools/debug_tworank.py 0 2 3 2>&1 | grep shard | cut -c1-70) & (env $e python tools/debug_tworank.py 2 4 3 2>&1 | grep shard | cut -c1-70) & wait
done > $OUT/r05_debug_tworank3.txt 2>&1
cat $OUT/r05_debug_tworank3.txt
