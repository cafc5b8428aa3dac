#!/bin/bash
# r05: d-split flash kernel -- kernel tests, standalone timing against the 4-wave form (FRIDO_FLASH_DSPLIT=0), end-to-end A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash or attention" 2>&1 | tail -15) > $OUT/r05_flash_tests.log
tail -3 $OUT/r05_flash_tests.log
for v in 0 1; do echo "== FRIDO_FLASH_DSPLIT=$v"; FRIDO_FLASH_DSPLIT=$v python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > $OUT/r05_flash_attn_bench.txt
cat $OUT/r05_flash_attn_bench.txt
tools/ab_env_tuned.sh FRIDO_FLASH_DSPLIT 0 1 > $OUT/r05_flash_ab.txt 2>&1
cat $OUT/r05_flash_ab.txt
