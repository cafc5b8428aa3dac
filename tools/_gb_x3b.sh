# bf16x3 conv shapes per tile, chunk-major vs tap-major K order
T=${TILES:-1,2,7}
for ko in 1 0; do
  export FRIDO_KORDER=$ko
  for shape in "16 64 64 192 192" "16 64 64 384 192" "16 32 32 384 384" "16 32 32 768 384" "16 16 16 576 576"; do
    echo "== conv $shape korder=$ko"; python tools/gemm_bench.py conv $shape 2 $T 2>&1 | grep -E "tile|rror"
  done
done
