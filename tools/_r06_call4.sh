R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -q -x -k "tiny or k_split" 2>&1 | tail -5
B="--no-cpu-baseline --no-bf16-extra --no-other-configs"
line() { grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/batch', 'roofline', d['roofline']['frac'], 'fwd', d['roofline']['forward_ms'])"; }
rm -f profiles/tune_cache.json
python bench.py --retune --steps 2 --warmup 1 $B 2>$OUT/r06c4_retune.err | line "retune" | tee $OUT/r06_c4_ab.txt
grep -i "warn\|no candidate" $OUT/r06c4_retune.err | head -5
cp profiles/tune_cache.json $OUT/tune_cache_c4.json
( time python -m pytest tests -m "gpu and gate" -q -x --durations=5 > $OUT/r06c4_gate.log 2>&1 ) 2> $OUT/r06c4_gate.time; tail -8 $OUT/r06c4_gate.log; cat $OUT/r06c4_gate.time
for i in 1 2; do
  python bench.py --steps 2 --warmup 1 $B 2>&1 | line "tiny=1" | tee -a $OUT/r06_c4_ab.txt
  FRIDO_GN_CONV_TINY=0 FRIDO_TUNE_ON_MISS=tune python bench.py --steps 2 --warmup 1 $B 2>&1 | line "tiny=0" | tee -a $OUT/r06_c4_ab.txt
done
FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 python tools/profile_forward.py --precision bf16x3 --top 70 > $OUT/r06c4_forward_per_op.txt 2>&1; grep -E "forward:|conv2x2|t40|N=3 |N=4 " $OUT/r06c4_forward_per_op.txt | head -20
