#!/bin/bash
# A/B two builds of the library on the full benchmark, interleaved on ONE box, EACH with its own tile tuning (use when a
# change shifts which tiles win; tools/ab_lib.sh pins the first library's choices for both):
#   tools/ab_lib_tuned.sh frido_amd/libfrido_hip_old.so frido_amd/libfrido_hip.so
A=${1:?old lib}; B=${2:?new lib}
for L in $A $B; do
  FRIDO_LIB=$PWD/$L FRIDO_TUNE_CACHE=/tmp/tune_$(basename $L).json python bench.py --retune --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs > /dev/null 2>&1
done
for i in 1 2 3; do
  for L in $A $B; do
    FRIDO_LIB=$PWD/$L FRIDO_TUNE_CACHE=/tmp/tune_$(basename $L).json python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"
  done
done
