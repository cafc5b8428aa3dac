for shape in "dense 16384 384 1536" "dense 4096 576 2304" "geglu 16384 1536 384"; do
  for m in 0 16 80 1; do
    L=frido_amd/libfrido_hip.so; [ $m != 0 ] && L=tools/ablate/libfrido_abl_$m.so
    echo "== $shape ablate=$m (16: planes interleaved per 32-element chunk; 80: + DMA pieces of 8 rows x 128 B; 1: no DMA)"
    FRIDO_LIB=$PWD/$L python tools/gemm_bench.py $shape 2 7,1 2>&1 | grep -E "tile|rror"
  done
done
