# K sweep of the 8x8-plane conv (M = 1024, split-K 8): loop rate vs fixed cost -> profiles/r02_gemm_k_sweep_small_m.txt
for cin in 192 480 960 1920 3840; do
  for ne in 0 1; do
    echo "== conv 16x8x8 $cin->960 splitk=8 NOEPI=$ne"; NOEPI=$ne SPLITK=8 python tools/gemm_bench.py conv 16 8 8 $cin 960 1 17,7,1 2>&1 | grep -E "tile|Error"
  done
done
for cin in 960 3840; do
    echo "== conv 16x8x8 $cin->960 splitk=1"; SPLITK=1 python tools/gemm_bench.py conv 16 8 8 $cin 960 1 17,7,1,3,13 2>&1 | grep -E "tile|Error"
done
