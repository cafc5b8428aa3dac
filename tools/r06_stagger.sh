#!/bin/bash
# Round-6 opener for DESIGN.md section 7 item 5 (one gpurun call, ~12 min; needs tools/build_stagger.sh run beforehand, the .so travels):
#  (1) kernel trace of the bench command with FRIDO_STAGGER_US = 0 and 8 on the run-time-stagger build -> tools/trace_diff.py: WHICH kernels moved;
#  (2) the delay swept end to end (0 4 6 8 10 12 16 us), then the smallest eligible grid (512 / 768 / 1024 workgroups) at the best delay;
#      all runs on the shipped library's pinned tiles (FRIDO_TUNE_TAG), interleaved twice.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
export FRIDO_LIB=$R/tools/ablate/libfrido_stagger.so
export FRIDO_TUNE_TAG=$(sha256sum frido_amd/libfrido_hip.so | cut -c1-16) FRIDO_TUNE_CACHE_READONLY=1
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs"
( cd /tmp; export TMPDIR=/tmp
  for us in 0 8; do
    FRIDO_STAGGER_US=$us timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_s$us -- $B > /tmp/kt_s$us.json 2> /tmp/kt_s$us.log
  done )
python tools/trace_diff.py /tmp/kt_s0 /tmp/kt_s8 40 > $OUT/r06_stagger_trace_diff.txt 2>&1; head -30 $OUT/r06_stagger_trace_diff.txt
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch')"; }
( for i in 1 2; do for us in 0 4 6 8 10 12 16; do
    FRIDO_STAGGER_US=$us timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | line "stagger_us=$us"
  done; done ) > $OUT/r06_stagger_delay_sweep.txt 2>&1; cat $OUT/r06_stagger_delay_sweep.txt
BEST=$(sort -k2 -n -r $OUT/r06_stagger_delay_sweep.txt | head -1 | sed 's/stagger_us=\([0-9]*\).*/\1/')
( for i in 1 2; do for wg in 512 768 1024; do
    FRIDO_STAGGER_US=$BEST FRIDO_STAGGER_MIN_WG=$wg timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | line "stagger_us=$BEST min_wg=$wg"
  done; done ) > $OUT/r06_stagger_min_wg_sweep.txt 2>&1; cat $OUT/r06_stagger_min_wg_sweep.txt
