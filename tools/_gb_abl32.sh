for shape in "conv 16 32 32 384 384" "dense 16384 384 1536"; do
  for m in 0 32 1; do
    L=frido_amd/libfrido_hip.so; [ $m != 0 ] && L=tools/ablate/libfrido_abl_$m.so
    echo "== $shape ablate=$m (32: every DMA piece reads the same 64 bytes; 1: no DMA)"
    FRIDO_LIB=$PWD/$L python tools/gemm_bench.py $shape 2 7,1 2>&1 | grep -E "tile|rror"
  done
done
