#!/bin/bash
# The last (short) GPU call of round 5, every leg under its own `timeout`, most valuable first:
#  (1) the depth-2 denoiser (transformer_depth = 2) against the fixture generated from the reference, next to the two old fixtures;
#  (2) DESIGN.md section 7 item 5, per launch: the GEGLU projections (multi-round 4-wave launches) on the shipped library and on
#      the stagger builds (tools/build_ablate.sh 1024 with FRIDO_STAGGER_US = 6 / 12);
#  (3) the same end to end, interleaved, all libraries on the PINNED tiles of the shipped one (FRIDO_TUNE_TAG = its hash).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_model_gpu.py -q -k "test_unet_forward_matches_reference_golden or test_unet_forward_ragged" > $OUT/r05_depth2_tests.log 2>&1
tail -3 $OUT/r05_depth2_tests.log
BASE=frido_amd/libfrido_hip.so; S6=tools/ablate/libfrido_abl_1024_s6.so; S12=tools/ablate/libfrido_abl_1024_s12.so
( for shape in "geglu 16384 1536 384" "geglu 4096 2304 576" "dense 16384 3072 384"; do
    for L in $BASE $S6 $S12; do
      echo "== $shape   $L"
      FRIDO_LIB=$PWD/$L timeout 120 python tools/gemm_bench.py $shape 2 2,1 2>&1 | grep -E "tile|rror"
    done
  done ) > $OUT/r05_stagger_per_launch.txt 2>&1
cat $OUT/r05_stagger_per_launch.txt
export FRIDO_TUNE_TAG=$(sha256sum $BASE | cut -c1-16) FRIDO_TUNE_CACHE_READONLY=1
( for i in 1 2; do
    for L in $BASE $S12 $S6; do
      FRIDO_LIB=$PWD/$L timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
    done
  done ) > $OUT/r05_stagger_end_to_end.txt 2>&1
cat $OUT/r05_stagger_end_to_end.txt
