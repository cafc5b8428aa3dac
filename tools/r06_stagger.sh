#!/bin/bash
# Round-6 opener for DESIGN.md section 7 item 5 (one gpurun call, ~12 min; needs tools/build_stagger.sh run beforehand, the .so travels):
#  (1) kernel trace of the bench command with FRIDO_STAGGER_US = 0 and 8 on the run-time-stagger build -> tools/trace_diff.py: WHICH kernels moved;
#  (2) the delay swept end to end (0.5 ... 16 us), the best two confirmed interleaved, then smaller eligible grids (512 / 320 workgroups) at
#      sub-k-step delays, and the placement-model control; all runs on the shipped library's pinned tiles (FRIDO_TUNE_TAG).  ~25 min.
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
run() { timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs 2>&1 | line "$1"; }
# (a) the delay, once over a wide range incl. sub-k-step offsets (a k-step of the 4-wave tiles is ~1 us)
( for us in 0 0.5 1 2 4 6 8 10 12 16; do FRIDO_STAGGER_US=$us run "stagger_us=$us"; done ) > $OUT/r06_stagger_delay_sweep.txt 2>&1; cat $OUT/r06_stagger_delay_sweep.txt
BEST=$(grep -v "stagger_us=0 " $OUT/r06_stagger_delay_sweep.txt | sort -k2 -n -r | head -1 | sed 's/stagger_us=\([0-9.]*\).*/\1/')
SECOND=$(grep -v "stagger_us=0 " $OUT/r06_stagger_delay_sweep.txt | sort -k2 -n -r | sed -n 2p | sed 's/stagger_us=\([0-9.]*\).*/\1/')
# (b) base / best / second best interleaved twice
( for i in 1 2; do for us in 0 $BEST $SECOND; do FRIDO_STAGGER_US=$us run "stagger_us=$us"; done; done ) > $OUT/r06_stagger_confirm.txt 2>&1; cat $OUT/r06_stagger_confirm.txt
# (c) smaller grids (one-round two-per-CU launches: 512 workgroups; 1.5 per CU: 384) at sub-k-step delays and at the best one
( for wg in 512 320; do for us in 0.5 1 $BEST; do FRIDO_STAGGER_US=$us FRIDO_STAGGER_MIN_WG=$wg run "stagger_us=$us min_wg=$wg"; done; done ) > $OUT/r06_stagger_min_wg_sweep.txt 2>&1
cat $OUT/r06_stagger_min_wg_sweep.txt
# (d) control: the other placement model (every other workgroup of an XCD waits) at the best delay
( FRIDO_STAGGER_US=$BEST FRIDO_STAGGER_MODE=1 run "stagger_us=$BEST mode=1"; FRIDO_STAGGER_US=$BEST run "stagger_us=$BEST mode=0" ) > $OUT/r06_stagger_mode_control.txt 2>&1
cat $OUT/r06_stagger_mode_control.txt
# (e) the one-workgroup-per-CU kernels (8-wave igemm tiles, fused GroupNorm + conv): odd XCDs start late (flags bit 26)
( for us in 0 4 8 12; do FRIDO_STAGGER_US=$us FRIDO_STAGGER_8W=1 FRIDO_STAGGER_MIN_WG=4096 run "8w stagger_us=$us"; done
  FRIDO_STAGGER_US=$BEST FRIDO_STAGGER_8W=1 run "8w + 4-wave stagger_us=$BEST" ) > $OUT/r06_stagger_8w.txt 2>&1
cat $OUT/r06_stagger_8w.txt
