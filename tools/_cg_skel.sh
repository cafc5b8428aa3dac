#!/bin/bash
# skeleton ablation of the fused GroupNorm + conv kernel: what is left of a launch without loads / conversion / MFMAs / DMA (31),
# then also without the epilogue (+32), the statistics prologue (+64), the per-step barriers (+128)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 FRIDO_TUNE_TAG=f17366f344c550df
for m in 0 32 96 31 63 127 255; do
  echo "#### CG_ABLATE=$m"
  if [ $m = 0 ]; then L=$R/frido_amd/libfrido_hip.so; else L=$R/tools/ablate/libfrido_cg_$m.so; fi
  FRIDO_LIB=$L python tools/gnconv_bench.py 16 64 64 192 0 192 0 0 20 2>&1 | grep -v amdgpu.ids | grep "fused"
  FRIDO_LIB=$L python tools/gnconv_bench.py 16 32 32 384 0 384 0 0 21 2>&1 | grep -v amdgpu.ids | grep "fused"
done
