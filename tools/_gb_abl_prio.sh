for shape in "conv 16 32 32 384 384" "conv 16 64 64 192 192" "dense 16384 384 1536" "geglu 16384 1536 384"; do
  for m in 0 128 256 384; do
    L=frido_amd/libfrido_hip.so; [ $m != 0 ] && L=tools/ablate/libfrido_abl_$m.so
    echo "== $shape variant=$m (128: s_setprio 1 around every MFMA group; 256: waves 4-7 issue their DMA half a window later)"
    FRIDO_LIB=$PWD/$L python tools/gemm_bench.py $shape 2 7,1 2>&1 | grep -E "tile|rror"
  done
done
