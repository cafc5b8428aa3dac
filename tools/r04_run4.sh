#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -m pytest tests/test_kernels_gpu.py -k "gn_conv" -q 2>&1 | tail -5
{
python tools/gnconv_bench.py 16 64 64 192 0 192 0 0 20
python tools/gnconv_bench.py 16 64 64 192 0 192 1 0 20
python tools/gnconv_bench.py 16 64 64 384 192 192 0 576 20
python tools/gnconv_bench.py 16 64 64 384 192 192 1 576 20
python tools/gnconv_bench.py 16 32 32 384 0 384 0 0 21
python tools/gnconv_bench.py 16 32 32 384 0 384 1 0 21
python tools/gnconv_bench.py 16 32 32 576 384 384 1 960 21
} 2>&1 | grep -v amdgpu.ids > $OUT/r04_gnconv_bench_v3.txt
cat $OUT/r04_gnconv_bench_v3.txt
