#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for v in 256 128; do FRIDO_GN_CONV_PREFER=$v FRIDO_TUNE_CACHE=/tmp/tune_b32_$v.json python bench.py --batch 32 --retune --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1; done
for i in 1 2; do for v in 256 128; do
  FRIDO_GN_CONV_PREFER=$v FRIDO_TUNE_CACHE=/tmp/tune_b32_$v.json python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=32 prefer=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
done; done 2>&1 | tee $OUT/r04_gnconv_b32_prefer_ab.txt
