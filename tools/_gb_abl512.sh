for shape in "conv 16 64 64 192 192" "conv 16 32 32 384 384" "conv 16 16 16 576 576" "dense 16384 384 1536" "dense 4096 576 2304" "geglu 16384 1536 384" "dense 1024 960 3840"; do
  for m in 0 512; do
    L=frido_amd/libfrido_hip.so; [ $m != 0 ] && L=tools/ablate/libfrido_abl_$m.so
    echo "== $shape variant=$m (512: weight pieces of 8 rows x 128 B instead of 16 rows x 64 B, timing only)"
    FRIDO_LIB=$PWD/$L python tools/gemm_bench.py $shape 2 7,1,2 2>&1 | grep -E "tile|rror"
  done
done
