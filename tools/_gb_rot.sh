# rotated k-walk A/B (FridoGemm.flags bit 7 = rotation off), bf16x3 and bf16 -> profiles/r03_ab_kwalk_rotation.txt
for ns in 2 1; do
  T=7,1,2; [ $ns = 1 ] && T=17,11,12,9
  for shape in "conv 16 64 64 192 192" "conv 16 32 32 384 384" "conv 16 16 16 576 576" "dense 16384 384 1536" "dense 16384 384 384" "dense 4096 576 2304" "geglu 16384 1536 384" "dense 1024 960 3840"; do
    for f in 0 128; do
      echo "== $shape nsplit=$ns flags=$f (128 = rotation off)"
      FRIDO_GEMM_FLAGS=$f python tools/gemm_bench.py $shape $ns $T 2>&1 | grep -E "tile|rror"
    done
  done
done
