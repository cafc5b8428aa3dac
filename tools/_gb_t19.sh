# 256 x 192 eight-wave bf16x3 tile (19: wave tile 64 x 96, plain loop, 2-slot ring) against the tiles it competes with
for shape in "conv 16 64 64 192 192" "conv 16 64 64 384 192" "conv 16 64 64 576 192" "conv 16 32 32 384 384" "conv 16 32 32 768 384" "dense 65536 192 576" "dense 16384 3072 384" "conv 4 128 128 256 256"; do
  echo "== $shape"; python tools/gemm_bench.py $shape 2 2,7,18,19 2>&1 | grep -E "tile|Error"
done
