#!/bin/bash
# r05: d-split flash kernel with double-buffered K; 16x16-plane self-attention on it (FRIDO_ATTN_FLASH_MIN_KEYS=256)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or attention" 2>&1 | tail -5) > $OUT/r05_flash2_tests.log
tail -3 $OUT/r05_flash2_tests.log
for v in 512 256; do echo "== FRIDO_ATTN_FLASH_MIN_KEYS=$v"; FRIDO_ATTN_FLASH_MIN_KEYS=$v python tools/attn_bench.py 2>&1 | grep -E "Nk=1024|Nk=256"; done > $OUT/r05_flash2_attn_bench.txt
cat $OUT/r05_flash2_attn_bench.txt
tools/ab_env_tuned.sh FRIDO_ATTN_FLASH_MIN_KEYS 512 256 > $OUT/r05_flash2_ab.txt 2>&1
cat $OUT/r05_flash2_ab.txt
