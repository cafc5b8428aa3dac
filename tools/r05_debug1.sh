#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export FRIDO_TUNE_CACHE=/tmp/none.json
python tools/debug_status.py 4 2>&1 | grep -v amdgpu.ids > $OUT/r05_debug_status.txt
tail -40 $OUT/r05_debug_status.txt
T="tests/test_model_gpu.py::test_other_configs_at_their_per_gpu_batch"
for e in "X=1" "FRIDO_ATTN_SKIP_DEAD_STREAM=0" "FRIDO_LN_IN_ATTN=0" "FRIDO_SK_DEFER=0" "FRIDO_GN_FUSED_V4=0" "FRIDO_CHAIN_FF=0" "FRIDO_GN_CONV=0"; do
  echo "== $e"; env $e timeout 600 python -m pytest "$T" -q -x -s -k config3 2>&1 | grep -E "rows .* vs B = 1|passed|failed" | tail -2
done > $OUT/r05_debug_config3.txt 2>&1
cat $OUT/r05_debug_config3.txt
