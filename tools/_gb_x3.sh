# per-tile timing of the bf16x3 (parity arithmetic) GEMM family on the denoiser's top shapes -> profiles/r03_x3_tiles_*.txt
T=${TILES:-1,2,4,5,6,3,7}
run() { echo "== $*"; python tools/gemm_bench.py "$@" 2 $T 2>&1 | grep -E "tile|Error|error"; }
run conv 16 64 64 192 192
run conv 16 64 64 384 192
run conv 16 32 32 384 384
run conv 16 16 16 576 576
for sk in 4 8; do echo "== conv 16 8 8 960 960 splitk=$sk"; SPLITK=$sk python tools/gemm_bench.py conv 16 8 8 960 960 2 $T 2>&1 | grep -E "tile|Error"; done
run geglu 16384 1536 384
run geglu 4096 2304 576
run dense 16384 384 384
run dense 16384 384 1536
run dense 4096 576 576
run dense 4096 576 2304
run dense 1024 960 960
run dense 1024 960 3840
