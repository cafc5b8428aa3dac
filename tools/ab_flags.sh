#!/bin/bash
# A/B of FridoGemm.flags values (or any env switch) on the full benchmark with pinned tiles, interleaved, on ONE box:
#   tools/ab_flags.sh FRIDO_GEMM_FLAGS 0 3
VAR=${1:?env var}; shift
export FRIDO_TUNE_CACHE=${FRIDO_TUNE_CACHE:-/tmp/frido_tune.json}
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > /dev/null 2>&1      # fills the tile cache
for i in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'fwd', d['roofline']['forward_ms'], 'gemm TF', d['roofline']['achieved'])"
  done
done
