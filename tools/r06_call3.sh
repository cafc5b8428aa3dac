#!/bin/bash
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
