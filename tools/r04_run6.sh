#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
for rep in 1 2; do
for m in base prio; do
  if [ $m = base ]; then unset FRIDO_LIB; else export FRIDO_LIB=$R/tools/ablate/libfrido_cg_$m.so; fi
  echo "#### $m"
  python tools/gnconv_bench.py 16 64 64 192 0 192 0 0 20 | grep -E "fused t.*total|=="
  python tools/gnconv_bench.py 16 64 64 192 0 192 1 0 20 | grep -E "fused t.*total|=="
  python tools/gnconv_bench.py 16 32 32 384 0 384 0 0 21 | grep -E "fused t.*total|=="
  python tools/gnconv_bench.py 16 64 64 384 192 192 1 576 20 | grep -E "fused t.*total|=="
done; done
} 2>&1 | grep -v amdgpu.ids > $OUT/r04_gnconv_prio_ab.txt
cat $OUT/r04_gnconv_prio_ab.txt
