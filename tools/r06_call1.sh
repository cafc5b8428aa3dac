#!/bin/bash
# Round-6 GPU call 1: (a) the new gate subset and the trimmed full suite, timed, on the r05 binary + pinned tiles; (b) the run-time stagger
# build (tools/build_stagger.sh beforehand): kernel-trace diff base vs 8 us, delay sweep, confirmation, 8-wave (one-workgroup-per-CU) leg.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
( time python -m pytest tests -m "gpu and gate" -q -x --durations=12 > $OUT/r06_gate.log 2>&1 ) 2> $OUT/r06_gate.time; tail -16 $OUT/r06_gate.log; cat $OUT/r06_gate.time
( time python -m pytest tests -m gpu -q --durations=60 > $OUT/r06_suite_call1.log 2>&1 ) 2> $OUT/r06_suite_call1.time; tail -70 $OUT/r06_suite_call1.log; cat $OUT/r06_suite_call1.time
python -m cProfile -o /tmp/ema.prof -m pytest tests/test_model_gpu.py -q -k ema_scope > /dev/null 2>&1
python -c "import pstats; pstats.Stats('/tmp/ema.prof').sort_stats('cumulative').print_stats(45)" 2>&1 | tail -60 > $OUT/r06_ema_profile.txt
export FRIDO_LIB=$R/tools/ablate/libfrido_stagger.so
export FRIDO_TUNE_TAG=$(sha256sum frido_amd/libfrido_hip.so | cut -c1-16) FRIDO_TUNE_CACHE_READONLY=1
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs"
( cd /tmp; export TMPDIR=/tmp
  for us in 0 8; do
    FRIDO_STAGGER_US=$us timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_s$us -- $B > /tmp/kt_s$us.json 2> /tmp/kt_s$us.log
  done )
python tools/trace_diff.py /tmp/kt_s0 /tmp/kt_s8 40 > $OUT/r06_stagger_trace_diff.txt 2>&1; head -40 $OUT/r06_stagger_trace_diff.txt
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"; }
run() { timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | line "$1"; }
( for us in 0 4 8 12; do FRIDO_STAGGER_US=$us run "stagger_us=$us"; done ) > $OUT/r06_stagger_delay_sweep.txt 2>&1; cat $OUT/r06_stagger_delay_sweep.txt
BEST=$(grep -v "stagger_us=0 " $OUT/r06_stagger_delay_sweep.txt | sort -k2 -n -r | head -1 | sed 's/stagger_us=\([0-9.]*\).*/\1/')
( for i in 1 2; do for us in 0 $BEST; do FRIDO_STAGGER_US=$us run "stagger_us=$us"; done; done ) > $OUT/r06_stagger_confirm.txt 2>&1; cat $OUT/r06_stagger_confirm.txt
( for wg in 512; do for us in 1 $BEST; do FRIDO_STAGGER_US=$us FRIDO_STAGGER_MIN_WG=$wg run "stagger_us=$us min_wg=$wg"; done; done ) > $OUT/r06_stagger_min_wg_sweep.txt 2>&1
cat $OUT/r06_stagger_min_wg_sweep.txt
( for us in 4 8 12; do FRIDO_STAGGER_US=$us FRIDO_STAGGER_8W=1 FRIDO_STAGGER_MIN_WG=4096 run "8w-only stagger_us=$us"; done
  FRIDO_STAGGER_US=$BEST FRIDO_STAGGER_8W=1 run "8w + 4-wave stagger_us=$BEST" ) > $OUT/r06_stagger_8w.txt 2>&1
cat $OUT/r06_stagger_8w.txt
