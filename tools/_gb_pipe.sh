# per-tile record that decided where the software-pipelined loop is used -> profiles/r02_pipelined_loop_per_tile.txt
bash tools/_gb_direct.sh
for sk in 4 8; do
  echo "== conv 16x8x8 960->960 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 1 1,2,7,8,11,12,17 2>&1 | grep -E "tile|Error"
done
for sk in 1 2 3; do
  echo "== conv 16x16x16 576->576 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 16 16 576 576 1 1,2,4,7,8,11,12,17 2>&1 | grep -E "tile|Error"
done
echo "== dense 16384x3072x384"; python tools/gemm_bench.py dense 16384 3072 384 1 2,7,8,12,17 2>&1 | grep -E "tile|Error"
echo "== dense 4096x576x2304"; python tools/gemm_bench.py dense 4096 576 2304 1 1,2,3,4,6,11,12,13,14,16 2>&1 | grep -E "tile|Error"
echo "== dense 1024x960x960"; python tools/gemm_bench.py dense 1024 960 960 1 1,3,4,6,11,13,14,16 2>&1 | grep -E "tile|Error"
