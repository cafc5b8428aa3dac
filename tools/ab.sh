#!/bin/bash
# ONE interleaved end-to-end A/B harness on the full benchmark (r06: replaces ab_bench / ab_flags / ab_env_tuned / ab_lib / ab_lib3 /
# ab_lib_tuned and the r05 / r06 stagger one-offs).  Every arm runs `bench.py --steps 2 --warmup 1` without the CPU / bf16 / other-config
# legs, arms interleaved ROUNDS times on one box (box-to-box spread is +-2 %, run-to-run on one box ~0.15 %).
#
#   tools/ab.sh env   VAR v1 v2 ...        one environment switch over values, pinned tiles (profiles/tune_cache.json or $FRIDO_TUNE_CACHE)
#   tools/ab.sh tuned VAR v1 v2 ...        the same, but every value gets ITS OWN tile tuning first (switches that change the tuner's candidates)
#   tools/ab.sh lib   A.so B.so [C.so ...] builds of the library, all on the FIRST one's pinned tiles (FRIDO_TUNE_TAG)
#   tools/ab.sh libtuned A.so B.so ...     builds of the library, each with its own tuning
#   ROUNDS=3 (default 2)   EXTRA="--batch 32" (more bench.py flags)
# Recipes of committed profiles: r06_stagger_*: `tools/ab.sh env FRIDO_STAGGER_US 0 4 8 12`; r06_kg2_end_to_end_ab: `tools/ab.sh tuned FRIDO_TUNE_KG2 1 0`;
# r05_gn_fused_v4_ab: `tools/ab.sh env FRIDO_GN_FUSED_V4 1 0`; r04_x3_side_stream_ab: `tools/ab.sh env FRIDO_SIDE_STREAM 0 1`.
MODE=${1:?env | tuned | lib | libtuned}; shift
ROUNDS=${ROUNDS:-2}
B="--steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs $EXTRA"
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'fwd', d['roofline']['forward_ms'], 'gemm TF', d['roofline']['achieved'])"; }
case $MODE in
  env|tuned)
    VAR=${1:?env var}; shift
    if [ $MODE = tuned ]; then for v in "$@"; do env $VAR=$v FRIDO_TUNE_CACHE=/tmp/tune_${VAR}_$v.json python bench.py --retune $B > /dev/null 2>&1; done; fi
    for i in $(seq $ROUNDS); do for v in "$@"; do
      if [ $MODE = tuned ]; then env $VAR=$v FRIDO_TUNE_CACHE=/tmp/tune_${VAR}_$v.json FRIDO_TUNE_CACHE_READONLY=1 python bench.py $B 2>&1 | line "$VAR=$v"
      else env $VAR=$v python bench.py $B 2>&1 | line "$VAR=$v"; fi
    done; done;;
  lib)
    export FRIDO_TUNE_TAG=ab FRIDO_TUNE_CACHE=/tmp/tune_ab.json
    FRIDO_LIB=$PWD/$1 python bench.py --retune $B > /dev/null 2>&1
    for i in $(seq $ROUNDS); do for L in "$@"; do FRIDO_LIB=$PWD/$L FRIDO_TUNE_CACHE_READONLY=1 python bench.py $B 2>&1 | line "$L"; done; done;;
  libtuned)
    for L in "$@"; do FRIDO_LIB=$PWD/$L FRIDO_TUNE_CACHE=/tmp/tune_$(basename $L).json python bench.py --retune $B > /dev/null 2>&1; done
    for i in $(seq $ROUNDS); do for L in "$@"; do
      FRIDO_LIB=$PWD/$L FRIDO_TUNE_CACHE=/tmp/tune_$(basename $L).json FRIDO_TUNE_CACHE_READONLY=1 python bench.py $B 2>&1 | line "$L"; done; done;;
  *) sed -n 2,14p "$0"; exit 2;;
esac
