#!/bin/bash
# Library with the RUN-TIME stagger (FridoGemm.flags bits 8..23; -DFRIDO_STAGGER_RT=1) next to the shipped one, for A/B sweeps of
# FRIDO_STAGGER_US / FRIDO_STAGGER_MIN_WG on one build:   tools/build_stagger.sh   then
#   FRIDO_LIB=$PWD/tools/ablate/libfrido_stagger.so tools/ab_env_tuned.sh FRIDO_STAGGER_US 0 8     (profiles/r05_stagger_*.txt: the compile-time form)
cd "$(dirname "$0")/.." && mkdir -p tools/ablate
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ifrido_amd/csrc -Wno-unused-result -ffp-contract=on"
( /opt/rocm/bin/hipcc $FL -DFRIDO_STAGGER_RT=1 -c frido_amd/csrc/igemm.hip -o tools/ablate/igemm_stagger.o ) &
( /opt/rocm/bin/hipcc $FL -DFRIDO_STAGGER_RT=1 -c frido_amd/csrc/convgn.hip -o tools/ablate/convgn_stagger.o ) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ablate/igemm_stagger.o tools/ablate/convgn_stagger.o frido_amd/csrc/{norm,misc,attn,flash,runtime}.o -o tools/ablate/libfrido_stagger.so
ls -la tools/ablate/libfrido_stagger.so
