#!/bin/bash
# A/B of one environment switch on the full benchmark with the autotuner's choices pinned (noise ~0.1 %):
#   tools/ab_bench.sh FRIDO_GN_FUSED        -> runs bench.py with VAR=1 / VAR=0, twice each
VAR=${1:?env var to flip}
export FRIDO_TUNE_CACHE=${FRIDO_TUNE_CACHE:-/tmp/frido_tune.json}
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > /dev/null 2>&1      # fills the tile cache
for i in 1 2; do
  for v in 1 0; do
    env $VAR=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
  done
done
