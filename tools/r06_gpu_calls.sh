#!/bin/bash
# The GPU calls of round 6, one function per call (what produced profiles/r06_*): tools/r06_gpu_calls.sh call1 | call2 | call3

call1() {
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
}

call2() {
  # Round-6 GPU call 2: the new binary (status poll / auto planes, K split inside the workgroup, shipped stagger): gate, per-shape sweep of the
  # KG2 tiles, re-tune with and without them, interleaved end-to-end A/B, per-op listing.
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
  ( for sh in "dense 1024 960 960" "dense 4096 576 576" "dense 16384 384 384" "dense 1024 960 3840" "dense 4096 576 2304" "dense 256 576 576" "dense 1024 7680 960"; do
      echo "== $sh"; python tools/gemm_bench.py $sh 2 3,33,4,34,6,36,1,31,5,35 2>&1 | grep -E "tile|rror"; done ) > $OUT/r06_kg2_sweep.txt 2>&1; cat $OUT/r06_kg2_sweep.txt
  B="--no-cpu-baseline --no-bf16-extra --no-other-configs"
  line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'roofline', d['roofline']['frac'])"; }
  rm -f profiles/tune_cache.json
  python bench.py --retune --steps 2 --warmup 1 $B 2>&1 | line "retune kg2=1" | tee $OUT/r06_kg2_ab.txt
  cp profiles/tune_cache.json $OUT/tune_cache_kg2.json
  ( time python -m pytest tests -m "gpu and gate" -q -x --durations=8 > $OUT/r06c2_gate.log 2>&1 ) 2> $OUT/r06c2_gate.time; tail -14 $OUT/r06c2_gate.log; cat $OUT/r06c2_gate.time
  python -m pytest tests/test_model_gpu.py -q -x -k "heavy or ema_scope or status" > $OUT/r06c2_heavy.log 2>&1; tail -5 $OUT/r06c2_heavy.log
  FRIDO_TUNE_KG2=0 FRIDO_TUNE_CACHE=/tmp/nokg2.json python bench.py --retune --steps 2 --warmup 1 $B 2>&1 | line "retune kg2=0" | tee -a $OUT/r06_kg2_ab.txt
  for i in 1 2; do
    python bench.py --steps 2 --warmup 1 $B 2>&1 | line "kg2=1" | tee -a $OUT/r06_kg2_ab.txt
    FRIDO_TUNE_KG2=0 FRIDO_TUNE_CACHE=/tmp/nokg2.json python bench.py --steps 2 --warmup 1 $B 2>&1 | line "kg2=0" | tee -a $OUT/r06_kg2_ab.txt
  done
  FRIDO_STAGGER_US=0 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "kg2=1 stagger=0" | tee -a $OUT/r06_kg2_ab.txt
  FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 python tools/profile_forward.py --precision bf16x3 --top 70 > $OUT/r06c2_forward_per_op.txt 2>&1; head -75 $OUT/r06c2_forward_per_op.txt
  python -c "
  import json
  a=json.load(open('$OUT/tune_cache_kg2.json'))
  print('KG2 entries:', [(k, v) for k, v in a['entries'] if v[0] > 30])
  "
}

call3() {
  # Round-6 GPU call 3: (a) two half-batches on two streams vs one batch-16 chain in the parity arithmetic; (b) in-context tile selection
  # (tools/tune_in_context.py, incl. the KG2 tiles and per-signature start delays) and its end-to-end A/B.
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
  python tools/dual_stream_exp.py --precision bf16x3 --parts 2 2>&1 | grep -v amdgpu.ids > $OUT/r06_dual_stream_x3.txt; cat $OUT/r06_dual_stream_x3.txt
  FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 python tools/tune_in_context.py --reps 5 --min-gain 0.02 --stagger 4,8,12,16,24 --min-wg 512 \
      --out $OUT/r06_tune_in_context.json --write-cache $OUT/tune_cache_ctx.json > $OUT/r06_tune_in_context.log 2>&1; tail -40 $OUT/r06_tune_in_context.log
  B="--no-cpu-baseline --no-bf16-extra --no-other-configs"
  line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'roofline', d['roofline']['frac'])"; }
  for i in 1 2 3; do
    python bench.py --steps 2 --warmup 1 $B 2>&1 | line "pinned(microbench)" | tee -a $OUT/r06_ctx_ab.txt
    FRIDO_TUNE_CACHE=$OUT/tune_cache_ctx.json FRIDO_TUNE_CACHE_READONLY=1 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "in-context" | tee -a $OUT/r06_ctx_ab.txt
  done
}

call4() {
  # Round-6 GPU call 4 (second session): scalar-base staging loads of the fused GroupNorm + conv kernel and the pass-major MFMA order of the
  # plain-loop k-step (FRIDO_SLAB0; tools/build_variants.sh head "" slab0 -DFRIDO_SLAB0=0 slab1 -DFRIDO_SLAB0=1 slab2 -DFRIDO_SLAB0=2 beforehand):
  # gate on the new default build, per-launch times of the affected kernels per variant, interleaved end-to-end A/B on one set of pinned tiles.
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
  ( time python -m pytest tests -m "gpu and gate" -q -x --durations=5 > $OUT/r06c4_gate.log 2>&1 ) 2> $OUT/r06c4_gate.time; tail -5 $OUT/r06c4_gate.log; cat $OUT/r06c4_gate.time
  ( for v in head slab0 slab1 slab2; do
      export FRIDO_LIB=$R/tools/ablate/libfrido_$v.so
      echo "== $v"
      python tools/gnconv_bench.py 16 64 64 192 0 192 0 0 2>&1 | grep -E "fused.*total|^=="
      python tools/gnconv_bench.py 16 64 64 192 0 192 1 0 2>&1 | grep -E "fused.*total|^=="
      python tools/gnconv_bench.py 16 64 64 192 0 192 1 384 2>&1 | grep -E "fused.*total|^=="
      python tools/gnconv_bench.py 16 32 32 384 0 384 0 0 2>&1 | grep -E "fused.*total|^=="
      python tools/gemm_bench.py dense 16384 3072 384 2 2,19 2>&1 | grep -E "tile|rror"
      python tools/gemm_bench.py dense 4096 4608 576 2 2,5 2>&1 | grep -E "tile|rror"
    done ) > $OUT/r06_slab0_per_launch.txt 2>&1; cat $OUT/r06_slab0_per_launch.txt
  unset FRIDO_LIB
  ROUNDS=3 bash tools/ab.sh lib tools/ablate/libfrido_head.so tools/ablate/libfrido_slab0.so tools/ablate/libfrido_slab1.so tools/ablate/libfrido_slab2.so > $OUT/r06_slab0_end_to_end_ab.txt 2>&1
  cat $OUT/r06_slab0_end_to_end_ab.txt
}

call5() {
  # Round-6 GPU call 5: the fused GroupNorm + conv kernel with its step barrier INSIDE the k-step (CG_MIDBAR, convgn.hip;
  # tools/build_variants.sh mb1 "" mb0 -DCG_MIDBAR=0 beforehand): gate on the new default build, per-launch times, interleaved end-to-end A/B.
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
  ( time python -m pytest tests -m "gpu and gate" -q -x --durations=5 > $OUT/r06c5_gate.log 2>&1 ) 2> $OUT/r06c5_gate.time; tail -5 $OUT/r06c5_gate.log; cat $OUT/r06c5_gate.time
  python -m pytest tests/test_kernels_gpu.py -q -x -k "gn_conv or convgn or fused" > $OUT/r06c5_convgn_tests.log 2>&1; tail -3 $OUT/r06c5_convgn_tests.log
  ( for i in 1 2; do for v in mb0 mb1; do
      export FRIDO_LIB=$R/tools/ablate/libfrido_$v.so
      echo "== $v"
      python tools/gnconv_bench.py 16 64 64 192 0 192 0 0 20 2>&1 | grep -E "fused.*total|^=="
      python tools/gnconv_bench.py 16 64 64 384 0 192 1 384 20 2>&1 | grep -E "fused.*total|^=="
      python tools/gnconv_bench.py 16 32 32 384 0 384 0 0 21 2>&1 | grep -E "fused.*total|^=="
    done; done ) > $OUT/r06_midbar_per_launch.txt 2>&1; cat $OUT/r06_midbar_per_launch.txt
  unset FRIDO_LIB
  ROUNDS=3 bash tools/ab.sh lib tools/ablate/libfrido_mb0.so tools/ablate/libfrido_mb1.so > $OUT/r06_midbar_end_to_end_ab.txt 2>&1
  cat $OUT/r06_midbar_end_to_end_ab.txt
}

call6() {
  # Round-6 GPU call 6: per-wave cycle accounting of the fused GroupNorm + conv kernel (tools/cg_prof.py on the -DCG_PROF=1 build)
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
  export FRIDO_LIB=$R/tools/ablate/libfrido_prof.so
  ( python tools/cg_prof.py 16 64 64 192 192 0 0
    python tools/cg_prof.py 16 64 64 192 192 1 0
    python tools/cg_prof.py 16 64 64 384 192 0 0
    python tools/cg_prof.py 16 32 32 384 384 0 0 21
    python tools/cg_prof.py 16 64 64 192 192 1 384 ) 2>&1 | grep -v amdgpu.ids > $OUT/r06_cg_prof.txt
  cat $OUT/r06_cg_prof.txt
}

call7() {
  # Round-6 GPU call 7: the SPREAD form of the fused kernel (weight DMA pieces and staging loads inside the k-step, CG_SPREAD;
  # tools/build_variants.sh sp1 "-DCG_SPREAD=1 -DCG_MIDBAR=0" profsp "-DCG_PROF=1 -DCG_SPREAD=1 -DCG_MIDBAR=0" prof "-DCG_PROF=1 -DCG_MIDBAR=0" mb0 "-DCG_MIDBAR=0")
  R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
  FRIDO_LIB=$R/tools/ablate/libfrido_sp1.so python -m pytest tests/test_kernels_gpu.py -q -x -k "gn_conv" > $OUT/r06c7_convgn_tests.log 2>&1; tail -3 $OUT/r06c7_convgn_tests.log
  ( for v in prof profsp; do echo "#### $v"; export FRIDO_LIB=$R/tools/ablate/libfrido_$v.so
      python tools/cg_prof.py 16 64 64 192 192 0 0; python tools/cg_prof.py 16 64 64 192 192 1 0; python tools/cg_prof.py 16 32 32 384 384 0 0 21; done ) 2>&1 | grep -v amdgpu.ids > $OUT/r06_cg_prof_spread.txt
  cat $OUT/r06_cg_prof_spread.txt
  unset FRIDO_LIB
  ROUNDS=3 bash tools/ab.sh lib tools/ablate/libfrido_mb0.so tools/ablate/libfrido_sp1.so > $OUT/r06_spread_end_to_end_ab.txt 2>&1
  cat $OUT/r06_spread_end_to_end_ab.txt
}

"${1:?call1 | call2 | call3 | call4 | call5 | call6 | call7}"
