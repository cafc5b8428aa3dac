#!/bin/bash
# In-graph cost of each launch bucket: bench time with the bucket dropped from the captured step body (results garbage,
# timing valid) vs the full body.  Pinned tiles.  Output: one line per variant.
export FRIDO_TUNE_CACHE=${FRIDO_TUNE_CACHE:-/tmp/frido_tune.json}
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
for v in "" GN_STATS GN_APPLY GN_FUSED LAYERNORM ATTN_SMALL ATTN_FLASH GEMM:dense GEMM:conv "GN_STATS,GN_APPLY,GN_FUSED,LAYERNORM,ATTN_SMALL,ATTN_FLASH" ""; do
  FRIDO_DEBUG_SKIP="$v" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode 2>&1 | grep -v amdgpu.ids | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=[$v]', d['ms_per_step'], 'ms/batch', round(d['ms_per_step']/400,3), 'ms/forward')"
done
