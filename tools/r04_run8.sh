#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
{
for sk in 3 5 6 8; do python tools/gnconv_bench.py 16 16 16 576 0 576 0 0 20 $sk; done
python tools/gnconv_bench.py 16 16 16 576 0 576 1 0 20 5
python tools/gnconv_bench.py 16 16 16 384 0 576 0 384 20 5
python tools/gnconv_bench.py 16 16 16 576 384 576 1 960 20 5
python tools/gnconv_bench.py 16 16 16 576 384 576 1 960 20 8
} 2>&1 | grep -v amdgpu.ids | grep -v "GN_STATS" > $OUT/r04_gnconv_bench_16x16.txt
cat $OUT/r04_gnconv_bench_16x16.txt
bash tools/ab_env_tuned.sh FRIDO_GN_CONV_SPLITK 0 1 2>&1 | tee $OUT/r04_gnconv_splitk_ab.txt
