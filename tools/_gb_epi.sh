# epilogue anatomy: NOEPI=0 full | 2 everything but the stores | 4 only the barrier | 1 nothing
for ne in 0 2 4 1; do
  echo "== conv 16x64x64 192->192 NOEPI=$ne"; NOEPI=$ne python tools/gemm_bench.py conv 16 64 64 192 192 1 9,10,2 2>&1 | grep -E "tile|Error"
done
for ne in 0 2 4 1; do
  echo "== dense 16384x384x768 NOEPI=$ne"; NOEPI=$ne python tools/gemm_bench.py dense 16384 384 768 1 2,1 2>&1 | grep -E "tile|Error"
done
for ne in 0 2 4 1; do
  echo "== dense 4096x576x576 NOEPI=$ne"; NOEPI=$ne python tools/gemm_bench.py dense 4096 576 576 1 3,6,4,13 2>&1 | grep -E "tile|Error"
done
