# ablation ladder of the bf16x3 main loop (tools/build_ablate.sh 1 2 4 8 3 first) -> profiles/r03_x3_ablation*.txt
for shape in "conv 16 32 32 384 384" "conv 16 64 64 192 192" "dense 16384 384 1536"; do
  for m in 0 1 2 3 4 8; do
    L=frido_amd/libfrido_hip.so; [ $m != 0 ] && L=tools/ablate/libfrido_abl_$m.so
    echo "== $shape ablate=$m (1 no-DMA 2 no-reads 4 no-barrier 8 no-MFMA)"
    FRIDO_LIB=$PWD/$L python tools/gemm_bench.py $shape 2 ${TILES:-7,1} 2>&1 | grep -E "tile|rror"
  done
done
