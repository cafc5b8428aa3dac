#!/bin/bash
# Second stagger call of round 5: the delay swept per launch (3 / 4 / 6 / 8 / 12 us) incl. the batch-32 GEGLU shape, then 4 / 8 us end to end
# on the pinned tiles (6 / 12 us: profiles/r05_stagger_end_to_end.txt).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out; mkdir -p $OUT
BASE=frido_amd/libfrido_hip.so
( for shape in "geglu 16384 1536 384" "geglu 32768 1536 384" "geglu 4096 2304 576" "geglu 8192 2304 576"; do
    for L in $BASE tools/ablate/libfrido_abl_1024_s{3,4,6,8,12}.so; do
      echo "== $shape   $L"
      FRIDO_LIB=$PWD/$L timeout 120 python tools/gemm_bench.py $shape 2 2 2>&1 | grep -E "tile|rror"
    done
  done ) > $OUT/r05_stagger_sweep_per_launch.txt 2>&1
cat $OUT/r05_stagger_sweep_per_launch.txt
export FRIDO_TUNE_TAG=$(sha256sum $BASE | cut -c1-16) FRIDO_TUNE_CACHE_READONLY=1
( for i in 1 2; do
    for L in $BASE tools/ablate/libfrido_abl_1024_s4.so tools/ablate/libfrido_abl_1024_s8.so; do
      FRIDO_LIB=$PWD/$L timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
    done
  done ) > $OUT/r05_stagger_sweep_end_to_end.txt 2>&1
cat $OUT/r05_stagger_sweep_end_to_end.txt
