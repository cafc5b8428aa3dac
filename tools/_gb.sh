for sk in 3 5 6 10; do echo "== conv 16x8x8 960->960 sk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 1 10,9,1,2 2>&1 | grep -E "tile|Error" | head -5; done
for sk in 1 2 3; do echo "== conv 16x16x16 576->576 sk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 16 16 576 576 1 10,6,1 2>&1 | grep -E "tile|Error" | head -4; done
