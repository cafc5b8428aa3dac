R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_model_gpu.py -q -x -k "several_steps or sampler_matches_reference_golden or sampler_philox or config2_step_count or benchmarked_batch_rows" 2>&1 | tail -3
B="--no-cpu-baseline --no-bf16-extra --no-other-configs"
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'loop', d['loop_only_value'])"; }
rm -f profiles/tune_cache.json
python bench.py --retune --steps 2 --warmup 1 $B 2>/dev/null | line "retune" | tee $OUT/r06_c9_ab.txt
for i in 1 2 3; do
  FRIDO_GRAPH_STEPS=10 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "graph_steps=10" | tee -a $OUT/r06_c9_ab.txt
  FRIDO_GRAPH_STEPS=1 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "graph_steps=1" | tee -a $OUT/r06_c9_ab.txt
done
FRIDO_GRAPH_STEPS=40 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "graph_steps=40" | tee -a $OUT/r06_c9_ab.txt
