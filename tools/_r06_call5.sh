R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "tiny or several_steps" 2>&1 | tail -4
B="--no-cpu-baseline --no-bf16-extra --no-other-configs"
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'roofline', d['roofline']['frac'], 'fwd', d['roofline']['forward_ms'])"; }
rm -f profiles/tune_cache.json
python bench.py --retune --steps 2 --warmup 1 $B 2>/dev/null | line "retune" | tee $OUT/r06_c5_ab.txt
cp profiles/tune_cache.json $OUT/tune_cache_c5.json
for i in 1 2; do
  python bench.py --steps 2 --warmup 1 $B 2>&1 | line "tiny=1 graph_steps=1" | tee -a $OUT/r06_c5_ab.txt
  FRIDO_GN_CONV_TINY=0 FRIDO_TUNE_ON_MISS=tune python bench.py --steps 2 --warmup 1 $B 2>&1 | line "tiny=0 graph_steps=1" | tee -a $OUT/r06_c5_ab.txt
  FRIDO_GRAPH_STEPS=8 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "tiny=1 graph_steps=8" | tee -a $OUT/r06_c5_ab.txt
done
FRIDO_GRAPH_STEPS=25 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "tiny=1 graph_steps=25" | tee -a $OUT/r06_c5_ab.txt
FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 python tools/profile_forward.py --precision bf16x3 --top 70 > $OUT/r06c5_forward_per_op.txt 2>&1; grep -E "forward:|t40" $OUT/r06c5_forward_per_op.txt | head
