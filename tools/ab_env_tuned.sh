#!/bin/bash
# A/B of an env switch that changes the TUNER's candidate set (each value gets its own tile cache), interleaved on one box:
#   tools/ab_env_tuned.sh FRIDO_TUNE_BIG_SPLITK 0 1
VAR=${1:?env var}; shift
for v in "$@"; do env $VAR=$v FRIDO_TUNE_CACHE=/tmp/tune_${VAR}_$v.json python bench.py --retune --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs > /dev/null 2>&1; done
for i in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v FRIDO_TUNE_CACHE=/tmp/tune_${VAR}_$v.json python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
  done
done
