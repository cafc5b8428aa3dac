#!/bin/bash
# Variant builds of the library for interleaved A/B runs (tools/ab.sh lib ...): igemm.hip + convgn.hip compiled under extra flags, linked
# with the default build's other objects (run `make -C frido_amd/csrc` first).   tools/build_variants.sh name "flags" [name "flags" ...]
#   name = head : the two sources and igemm_shared.h as committed at HEAD (the baseline of an uncommitted kernel change)
cd "$(dirname "$0")/.." && mkdir -p tools/ablate
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Wno-unused-result -ffp-contract=on -DFRIDO_STAGGER_RT=1"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    src=frido_amd/csrc
    rest="frido_amd/csrc/norm.o frido_amd/csrc/misc.o frido_amd/csrc/attn.o frido_amd/csrc/flash.o frido_amd/csrc/runtime.o"
    if [ "$name" = head ]; then      # EVERY source as committed at HEAD
      src=/tmp/variants_head; rm -rf $src; mkdir -p $src
      for f in igemm.hip convgn.hip igemm_shared.h common.h norm.hip misc.hip attn.hip flash.hip runtime.hip; do git show HEAD:frido_amd/csrc/$f > $src/$f; done
      rest=""
      for f in norm misc attn flash runtime; do /opt/rocm/bin/hipcc $FL -I$src -c $src/$f.hip -o tools/ablate/${f}_head.o; rest="$rest tools/ablate/${f}_head.o"; done
    fi
    /opt/rocm/bin/hipcc $FL -I$src $flags -c $src/igemm.hip -o tools/ablate/igemm_$name.o &&
    /opt/rocm/bin/hipcc $FL -I$src $flags -c $src/convgn.hip -o tools/ablate/convgn_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ablate/igemm_$name.o tools/ablate/convgn_$name.o $rest -o tools/ablate/libfrido_$name.so
  ) &
done
wait
ls -la tools/ablate/libfrido_*.so
