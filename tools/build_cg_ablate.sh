#!/bin/bash
# Ablation builds of the fused GroupNorm + conv kernel (timing only): tools/ablate/libfrido_cg_<mask>.so = convgn.hip under -DCG_ABLATE=<mask>
cd "$(dirname "$0")/.." && mkdir -p tools/ablate
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ifrido_amd/csrc -Wno-unused-result -ffp-contract=on"
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $FL ${CG_FLAGS} -DCG_ABLATE=$m -c frido_amd/csrc/convgn.hip -o tools/ablate/convgn_$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ablate/convgn_$m.o frido_amd/csrc/{igemm,norm,misc,attn,flash,runtime}.o -o tools/ablate/libfrido_cg_$m.so ) &
done
wait
ls tools/ablate/libfrido_cg_*.so
