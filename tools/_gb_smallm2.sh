for sk in 4 8; do
  echo "== conv 16x8x8 960->960 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 1 1,7,8,9,11,17 2>&1 | grep -E "tile|Error"
done
for sk in 2 3 4; do
  echo "== conv 16x16x16 576->576 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 16 16 576 576 1 1,2,7,9,10,12,17 2>&1 | grep -E "tile|Error"
done
for sk in 2 4; do
echo "== dense 1024x960x3840 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py dense 1024 960 3840 1 1,3,4,11,13,14,17 2>&1 | grep -E "tile|Error"
done
