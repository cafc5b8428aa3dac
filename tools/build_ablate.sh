#!/bin/bash
# Ablation builds of the GEMM library (timing experiments only: results are garbage): tools/ablate/libfrido_abl_<mask>.so =
# the shipped sources with igemm.hip compiled under -DFRIDO_ABLATE=<mask> (see the macro's comment there).
#   tools/build_ablate.sh 1 2 4 8 ...   (extra flags through ABLATE_FLAGS, output suffix through ABLATE_SUFFIX:
#   ABLATE_FLAGS=-DFRIDO_STAGGER_US=12 ABLATE_SUFFIX=_s12 tools/build_ablate.sh 1024)     then     FRIDO_LIB=$PWD/tools/ablate/libfrido_abl_1.so python tools/gemm_bench.py ...
cd "$(dirname "$0")/.." && mkdir -p tools/ablate
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ifrido_amd/csrc -Wno-unused-result -ffp-contract=on"
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $FL -DFRIDO_ABLATE=$m $ABLATE_FLAGS -c frido_amd/csrc/igemm.hip -o tools/ablate/igemm_$m$ABLATE_SUFFIX.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ablate/igemm_$m$ABLATE_SUFFIX.o frido_amd/csrc/{convgn,norm,misc,attn,flash,runtime}.o -o tools/ablate/libfrido_abl_$m$ABLATE_SUFFIX.so ) &
done
wait
ls -la tools/ablate/*.so
