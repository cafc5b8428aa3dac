for sk in 4 8; do
  echo "== conv 16x8x8 960->960 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 1 17,18 2>&1 | grep -E "tile|Error"
done
for sk in 2 3 4; do
  echo "== conv 16x16x16 576->576 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 16 16 576 576 1 17,18,10 2>&1 | grep -E "tile|Error"
done
echo "== conv 16x32x32 384->384"; python tools/gemm_bench.py conv 16 32 32 384 384 1 17,18,10 2>&1 | grep -E "tile|Error"
echo "== conv 16x64x64 192->192"; python tools/gemm_bench.py conv 16 64 64 192 192 1 17,18,10 2>&1 | grep -E "tile|Error"
echo "== dense 16384x384x1536"; python tools/gemm_bench.py dense 16384 384 1536 1 12,17,18 2>&1 | grep -E "tile|Error"
echo "== dense 4096x576x2304"; python tools/gemm_bench.py dense 4096 576 2304 1 11,17,18 2>&1 | grep -E "tile|Error"
echo "== dense 16384x3072x384 (geglu shape, plain)"; python tools/gemm_bench.py dense 16384 3072 384 1 8,17,18 2>&1 | grep -E "tile|Error"
