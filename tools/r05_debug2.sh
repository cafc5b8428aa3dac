#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
FRIDO_TUNE_CACHE=/tmp/none.json python tools/debug_status.py 4 2>&1 | grep -v amdgpu.ids > $OUT/r05_debug_status.txt
tail -30 $OUT/r05_debug_status.txt
T="tests/test_model_gpu.py::test_other_configs_at_their_per_gpu_batch"
i=0
for e in "X=1" "FRIDO_SK_DEFER=0" "FRIDO_TUNE_BIG_SPLITK=0" "FRIDO_GN_FUSED_V4=0" "FRIDO_ATTN_SKIP_DEAD_STREAM=0"; do
  i=$((i+1))
  echo "== tuned, $e"; env $e FRIDO_TUNE_CACHE=/tmp/t_$i.json FRIDO_TUNE_ON_MISS=tune FRIDO_TUNE_CACHE_READONLY=0 timeout 900 python -m pytest "$T" -q -x -s -k config3 2>&1 | grep -E "rows .* vs B = 1|passed|failed" | tail -2
done > $OUT/r05_debug_config3_tuned.txt 2>&1
cat $OUT/r05_debug_config3_tuned.txt
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -s -k two_real 2>&1 | grep -E "two ranks|passed|failed" > $OUT/r05_debug_tworank.txt
cat $OUT/r05_debug_tworank.txt
