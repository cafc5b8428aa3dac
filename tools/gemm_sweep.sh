#!/bin/bash
# ONE parameterised per-tile GEMM / conv sweep over tools/gemm_bench.py (r05: replaces the seventeen tools/_gb_*.sh one-offs of rounds 2-4).
#
#   tools/gemm_sweep.sh [-n NSPLIT] [-t TILES] [-v "ENVVAR v1 v2 ..."]... [-a "MASK ..."] (-p PRESET | "shape" ...)
#     -n  1 = bf16, 2 = two-plane arithmetic (default 2)          -t  comma-separated tile ids (default 1,2,3,4,5,6,7)
#     -v  sweep an environment variable of gemm_bench.py / the library over values (may repeat: nested loops), e.g.
#         "SPLITK 1 2 4 8", "NOEPI 0 2 4 1", "SKMODE 0 1", "FRIDO_GEMM_FLAGS 0 128", "FRIDO_KORDER 1 0"
#     -a  ablation builds (tools/build_ablate.sh MASK ... first): every shape is timed on frido_amd/libfrido_hip.so ("0") and on
#         tools/ablate/libfrido_abl_MASK.so for each MASK
#     a shape is gemm_bench.py's own spelling: "conv B H W Cin Cout" | "dense M N K" | "geglu M N K"
# Presets (the shape lists the old scripts carried):
#   denoiser  the top shapes of a layout2i forward at B = 16        dense     the dense / GEGLU shapes of the 32^2 and 16^2 planes
#   conv      the 64^2 / 32^2 / 16^2 3x3 convs                       smallm    the 8^2 / 16^2 convs that run under split-K
#   t19       the shapes the 256 x 192 eight-wave tile competes on
# Recipes of the committed profiles:  r03_x3_tiles_*: -t 1,2,4,5,6,3,7 -p denoiser   |   r03_x3_ablation_ring: -t 7,1 -a "1 2 3 4 8" "conv 16 32 32 384 384"
#   "conv 16 64 64 192 192" "dense 16384 384 1536"   |   r03_x3_128B_pieces_ablation: -t 7,1 -a "16 80 1" -p dense   |   r03_ab_kwalk_rotation: -t 7,1,2
#   -v "FRIDO_GEMM_FLAGS 0 128" -p denoiser   |   r03_x3_korder_ab: -t 1,2,7 -v "FRIDO_KORDER 1 0" -p conv   |   r03_x3_splitk_inkernel_ab: -t 1,3,4,6,7,18
#   -v "SPLITK 2 4 8" -v "SKMODE 0 1" -p smallm   |   r03_x3_tile19_ab: -t 2,7,18,19 -p t19   |   r02_gemm_k_sweep*: -n 1 -v "NOEPI 0 1" with explicit K ladders
NS=2; TILES=1,2,3,4,5,6,7; VARS=(); ABL=""; PRESET=""
while getopts "n:t:v:a:p:" o; do
  case $o in n) NS=$OPTARG;; t) TILES=$OPTARG;; v) VARS+=("$OPTARG");; a) ABL=$OPTARG;; p) PRESET=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
case "$PRESET" in
  denoiser) SHAPES=("conv 16 64 64 192 192" "conv 16 64 64 384 192" "conv 16 32 32 384 384" "conv 16 16 16 576 576" "conv 16 8 8 960 960" "geglu 16384 1536 384"
                    "geglu 4096 2304 576" "dense 16384 384 384" "dense 16384 384 1536" "dense 4096 576 576" "dense 4096 576 2304" "dense 1024 960 960" "dense 1024 960 3840");;
  dense)    SHAPES=("dense 16384 384 1536" "dense 16384 384 384" "dense 4096 576 2304" "geglu 16384 1536 384" "dense 1024 960 3840");;
  conv)     SHAPES=("conv 16 64 64 192 192" "conv 16 64 64 384 192" "conv 16 32 32 384 384" "conv 16 32 32 768 384" "conv 16 16 16 576 576");;
  smallm)   SHAPES=("conv 16 8 8 960 960" "conv 16 8 8 1920 960" "conv 16 16 16 576 576" "conv 16 16 16 1152 576" "dense 1024 960 3840" "dense 4096 576 2304");;
  t19)      SHAPES=("conv 16 64 64 192 192" "conv 16 64 64 384 192" "conv 16 64 64 576 192" "conv 16 32 32 384 384" "conv 16 32 32 768 384" "dense 65536 192 576"
                    "dense 16384 3072 384" "conv 4 128 128 256 256");;
  "")       SHAPES=("$@");;
  *)        echo "unknown preset $PRESET" >&2; exit 2;;
esac
[ ${#SHAPES[@]} -eq 0 ] && { sed -n 2,20p "$0"; exit 2; }

one() {   # $1 = label suffix; environment already set
  for L in 0 $ABL; do
    lib=frido_amd/libfrido_hip.so; [ "$L" != 0 ] && lib=tools/ablate/libfrido_abl_$L.so
    echo "== $shape$1${ABL:+ ablate=$L}"
    FRIDO_LIB=$PWD/$lib python tools/gemm_bench.py $shape $NS $TILES 2>&1 | grep -E "tile|rror"
  done
}
sweep() {  # recursive nested loops over the -v axes: $1 = axis index, $2 = label so far
  if [ "$1" -ge ${#VARS[@]} ]; then one "$2"; return; fi
  set -- "$1" "$2" ${VARS[$1]}
  local idx=$1 lab=$2 name=$3; shift 3
  for val in "$@"; do export "$name=$val"; sweep $((idx + 1)) "$lab $name=$val"; done
  unset "$name"
}
for shape in "${SHAPES[@]}"; do sweep 0 ""; done
