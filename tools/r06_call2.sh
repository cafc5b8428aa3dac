#!/bin/bash
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
