#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
for m in 0 1 3 4 8 12 16 31; do
  if [ $m = 0 ]; then unset FRIDO_LIB; else export FRIDO_LIB=$R/tools/ablate/libfrido_cg_$m.so; fi
  echo "#### CG_ABLATE=$m"
  python tools/gnconv_bench.py 16 64 64 192 0 192 0 0 20 | grep -E "fused t|==|two-kernel  (total|GEMM)"
  python tools/gnconv_bench.py 16 64 64 192 0 192 1 0 20 | grep -E "fused t|==|two-kernel  (total|GEMM)"
  python tools/gnconv_bench.py 16 32 32 384 0 384 0 0 21 | grep -E "fused t|==|two-kernel  (total|GEMM)"
done
} 2>&1 | grep -v amdgpu.ids > $OUT/r04_gnconv_ablate_v2.txt
cat $OUT/r04_gnconv_ablate_v2.txt
