#!/bin/bash
# A/B/C...: several builds of the library on the full benchmark, interleaved on ONE box with the FIRST library's pinned tiles:
#   tools/ab_lib3.sh libA.so libB.so libC.so
export FRIDO_TUNE_TAG=ab FRIDO_TUNE_CACHE=/tmp/tune_ab.json
FRIDO_LIB=$PWD/$1 python bench.py --retune --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs > /dev/null 2>&1
for i in 1 2 3; do
  for L in "$@"; do
    FRIDO_LIB=$PWD/$L python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
  done
done
