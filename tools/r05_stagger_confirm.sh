#!/bin/bash
# Third stagger call: does the 8-us build's end-to-end gain (+2.5 % on the second box) repeat on another box?  base / 8 us / 6 us interleaved, pinned tiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out; mkdir -p $OUT
BASE=frido_amd/libfrido_hip.so
export FRIDO_TUNE_TAG=$(sha256sum $BASE | cut -c1-16) FRIDO_TUNE_CACHE_READONLY=1
( for i in 1 2; do
    for L in $BASE tools/ablate/libfrido_abl_1024_s8.so tools/ablate/libfrido_abl_1024_s6.so; do
      FRIDO_LIB=$PWD/$L timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
    done
  done ) > $OUT/r05_stagger_confirm_end_to_end.txt 2>&1
cat $OUT/r05_stagger_confirm_end_to_end.txt
