# K sweep with / without epilogue (NOEPI=0/1): per-launch fixed cost, epilogue cost and main-loop rate -> profiles/r02_gemm_k_sweep.txt
for cin in 32 64 128 192 384 576; do
  for ne in 0 1; do
    echo "== conv 16x64x64 $cin->192 NOEPI=$ne"; NOEPI=$ne python tools/gemm_bench.py conv 16 64 64 $cin 192 1 9,2 2>&1 | grep -E "tile|Error"
  done
done
for cin in 64 192 384 768; do
  for ne in 0 1; do
    echo "== conv 16x32x32 $cin->384 NOEPI=$ne"; NOEPI=$ne python tools/gemm_bench.py conv 16 32 32 $cin 384 1 10,2,9 2>&1 | grep -E "tile|Error"
  done
done
for k in 128 384 768 1536 3072; do
  for ne in 0 1; do
    echo "== dense 16384x384x$k NOEPI=$ne"; NOEPI=$ne python tools/gemm_bench.py dense 16384 384 $k 1 2,1,12 2>&1 | grep -E "tile|Error"
  done
done
