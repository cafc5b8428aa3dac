#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export FRIDO_TUNE_CACHE=/tmp/t_v.json FRIDO_TUNE_ON_MISS=tune FRIDO_TUNE_CACHE_READONLY=0
python tools/verify_deferred.py config3 32 2>&1 | grep -v amdgpu.ids > $OUT/r05_verify_deferred_c3_b32.txt
grep -E "MISMATCH|checked|WORST|Error|error" $OUT/r05_verify_deferred_c3_b32.txt | head -30
FRIDO_GN_FUSED_SK1024=0 python tools/verify_deferred.py config3 32 2>&1 | grep -v amdgpu.ids > $OUT/r05_verify_deferred_c3_b32_nt.txt
grep -E "MISMATCH|checked|WORST|Error|error" $OUT/r05_verify_deferred_c3_b32_nt.txt | head -30
