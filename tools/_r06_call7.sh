R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -q -x -k "gn_conv or tiny" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -q -x -k "several_steps or unet_full_width_forward or benchmarked_batch_forward" 2>&1 | tail -3
B="--no-cpu-baseline --no-bf16-extra --no-other-configs"
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'roofline', d['roofline']['frac'], 'fwd', d['roofline']['forward_ms'])"; }
rm -f profiles/tune_cache.json
python bench.py --retune --steps 2 --warmup 1 $B 2>/dev/null | line "retune" | tee $OUT/r06_c7_ab.txt
cp profiles/tune_cache.json $OUT/tune_cache_c7.json
for i in 1 2 3; do
  python bench.py --steps 2 --warmup 1 $B 2>&1 | line "stats_in_conv=1" | tee -a $OUT/r06_c7_ab.txt
  FRIDO_GN_STATS_IN_CONV=0 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "stats_in_conv=0" | tee -a $OUT/r06_c7_ab.txt
done
FRIDO_GRAPH_STEPS=10 python bench.py --steps 2 --warmup 1 $B 2>&1 | line "stats_in_conv=1 graph_steps=10" | tee -a $OUT/r06_c7_ab.txt
FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 python tools/profile_forward.py --precision bf16x3 --top 30 > $OUT/r06c7_forward_per_op.txt 2>&1; head -16 $OUT/r06c7_forward_per_op.txt
