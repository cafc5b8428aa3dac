# per-tile launch times (tools/gemm_bench.py) of the 64^2 / 32^2 convs and the 16384-row dense GEMMs: before / after record of an epilogue or main-loop change
for cin in 64 192 576; do
  echo "== conv 16x64x64 $cin->192"; python tools/gemm_bench.py conv 16 64 64 $cin 192 1 9,10,2 2>&1 | grep -E "tile|Error"
done
for cin in 192 768; do
  echo "== conv 16x32x32 $cin->384"; python tools/gemm_bench.py conv 16 32 32 $cin 384 1 10,2,9 2>&1 | grep -E "tile|Error"
done
for k in 384 768 1536; do
  echo "== dense 16384x384x$k"; python tools/gemm_bench.py dense 16384 384 $k 1 2,1,12 2>&1 | grep -E "tile|Error"
done
