#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
echo "== alone, sequentially"; python tools/debug_tworank.py 0 2 3 2>&1 | grep shard; python tools/debug_tworank.py 0 2 1 2>&1 | grep shard
echo "== two processes concurrently"; (python tools/debug_tworank.py 0 2 3 2>&1 | grep shard) & (python tools/debug_tworank.py 2 4 3 2>&1 | grep shard) & wait
echo "== static tiles only (no cache), concurrently"; (FRIDO_TUNE_CACHE=/tmp/x.json python tools/debug_tworank.py 0 2 2 2>&1 | grep shard) & (FRIDO_TUNE_CACHE=/tmp/x.json python tools/debug_tworank.py 2 4 2 2>&1 | grep shard) & wait
echo "== static, alone"; FRIDO_TUNE_CACHE=/tmp/x.json python tools/debug_tworank.py 0 2 2 2>&1 | grep shard; FRIDO_TUNE_CACHE=/tmp/x.json python tools/debug_tworank.py 2 4 1 2>&1 | grep shard
} > $OUT/r05_debug_tworank2.txt 2>&1
cat $OUT/r05_debug_tworank2.txt
FRIDO_TUNE_ON_MISS=static python tools/debug_status.py 16 2>&1 | grep -v amdgpu.ids > $OUT/r05_debug_status16.txt; tail -25 $OUT/r05_debug_status16.txt
