#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
python tools/gnconv_bench.py 16 64 64 192 0 192 0 0
python tools/gnconv_bench.py 16 64 64 192 0 192 1 0
python tools/gnconv_bench.py 16 64 64 384 192 192 0 576
python tools/gnconv_bench.py 16 64 64 384 192 192 1 576
python tools/gnconv_bench.py 16 32 32 384 0 384 0 0
python tools/gnconv_bench.py 16 32 32 384 0 384 1 0
python tools/gnconv_bench.py 16 32 32 576 384 384 1 960
python tools/gnconv_bench.py 16 32 32 192 0 384 0 192
} 2>&1 | grep -v amdgpu.ids > $OUT/r04_gnconv_bench_v1.txt
cat $OUT/r04_gnconv_bench_v1.txt
bash tools/ab_env_tuned.sh FRIDO_GN_CONV 0 1 2>&1 | tee $OUT/r04_gnconv_ab_v1.txt
