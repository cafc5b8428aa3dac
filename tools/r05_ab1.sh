#!/bin/bash
# r05 batch 1: kernel tests of the touched kernels, per-op profile, and an interleaved A/B of the r04 kernels (libfrido_hip_base.so =
# HEAD~ kernels built against the current header) against the current library, each with its own tuning.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export FRIDO_TUNE_CACHE=/tmp/tune_new.json
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "status_word or dead_stream or pack or vq_lookup or layernorm or attention_core or deferred or groupnorm or saturate" 2>&1 | tail -15) > $OUT/r05_ab1_tests.log
python tools/profile_forward.py --precision bf16x3 --top 70 > $OUT/r05_ab1_forward_per_op.txt 2>&1
tools/ab_lib_tuned.sh frido_amd/libfrido_hip_base.so frido_amd/libfrido_hip.so > $OUT/r05_ab1_ab.txt 2>&1
tail -8 $OUT/r05_ab1_ab.txt; tail -5 $OUT/r05_ab1_tests.log
