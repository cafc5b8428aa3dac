# in-kernel split-K reduction (FridoGemm.sk_mode 1: the last workgroup of each tile adds the slices) vs the splitk_reduce launch
# (sk_mode 0), bf16x3, on the sampler's split-K shapes; the reported time includes the reduce launch where there is one
for shape in "conv 16 8 8 960 960" "conv 16 8 8 1920 960" "conv 16 16 16 576 576" "conv 16 16 16 1152 576" "dense 1024 960 3840" "dense 4096 576 2304"; do
  for sk in 2 4 8; do
    for m in 0 1; do
      echo "== $shape splitk=$sk sk_mode=$m"; SKMODE=$m SPLITK=$sk python tools/gemm_bench.py $shape 2 1,3,4,6,7,18 2>&1 | grep -E "tile|Error"
    done
  done
done
