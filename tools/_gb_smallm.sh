# tile x split-K sweep on the small-M conv shapes (8x8 and 16x16 planes)
for sk in 1 2 4 8 12 16; do
  echo "== conv 16x8x8 960->960 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 1 1,2,4,6,7,8,9,10,11,12,17 2>&1 | grep -E "tile|Error"
done
for sk in 4 8; do
  echo "== conv 16x8x8 960->960 splitk=$sk NOEPI"; NOEPI=1 SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 1 1,2,7,8,9,10 2>&1 | grep -E "tile|Error"
done
for sk in 1 2 3 4 6; do
  echo "== conv 16x16x16 576->576 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 16 16 576 576 1 1,2,4,7,8,9,10,11,12,17 2>&1 | grep -E "tile|Error"
done
