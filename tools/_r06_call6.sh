R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -q -x -k "tiny" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py -q -x -k "several_steps" 2>&1 | grep -v "^$" | tail -40
FRIDO_TUNE_ON_MISS=tune python tools/profile_forward.py --precision bf16x3 --top 80 > $OUT/r06c6_forward_per_op.txt 2>&1; grep -E "forward:|t40" $OUT/r06c6_forward_per_op.txt | head
