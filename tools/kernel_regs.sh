#!/bin/bash
# compact register / scratch report for one HIP source of the library: tools/kernel_regs.sh frido_amd/csrc/igemm.hip [filter]
src=$1; filt=${2:-.}
cd "$(dirname "$src")"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-result -ffp-contract=on \
    -c "$(basename "$src")" -o /tmp/kernel_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -e 's/.*Name: //' -e 's/.*VGPRs: /v=/' -e 's/.*lane\]: /scratch=/' -e 's/.*SIMD\]: /occ=/' -e 's/ \[-Rpass.*//' |
    paste - - - - | while read n v s o; do echo "$(echo $n | c++filt | sed 's/(anonymous namespace):://; s/(FridoGemm)//; s/void //') $v $s $o"; done | grep -E "$filt"
