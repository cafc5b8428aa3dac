#!/bin/bash
# End-of-round GPU run.  FIRST the bench lines that (re)write the pinned tile cache for this library (profiles/tune_cache.json, keyed by
# the library's content hash: B = 16 headline incl. the config-3 / config-5 passes of `extra.other_configs`, B = 32 = config 4's per-GPU
# shard, B = 1 at full width), THEN the full GPU suite, which reads that cache (tests/conftest.py): the parity tests run on the tiles the
# benchmark runs on.  Then the repeat / torchrun lines and smoke().   tools/final_run.sh r05 [quick]
TAG=${1:-r05}; MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
rm -f profiles/tune_cache.json $OUT/e2e_error.json
python bench.py --retune --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_line_tune.json 2> $OUT/${TAG}_bench_line.err      # two-plane headline + bf16 extra + configs 3 / 5
python bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline --no-bf16-extra --retune > $OUT/${TAG}_bench_line_batch32.json 2>/dev/null   # config 4's per-GPU shard (the cache gains its signatures)
python bench.py --steps 1 --warmup 0 --batch 1 --ddim-steps 4 --no-cpu-baseline --no-bf16-extra --retune > /dev/null 2>&1   # B = 1 at full width: the shapes of the config-1 / config-2-step goldens and of the rows-vs-B=1 tests
cp profiles/tune_cache.json $OUT/tune_cache.json
(time python -m pytest tests -m gpu -q -s --durations=15 > $OUT/final_gpu_tests.log 2>&1); tail -6 $OUT/final_gpu_tests.log
cp $OUT/e2e_error.json $OUT/${TAG}_e2e_error.json 2>/dev/null
if [ "$MODE" = "full" ]; then
python bench.py --steps 5 --warmup 1 --no-bf16-extra --no-other-configs > $OUT/${TAG}_bench_line.json 2>/dev/null     # second process: pinned tiles, no tuning, WITH the CPU baseline
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > $OUT/${TAG}_bench_line_torchrun_n1.json 2>/dev/null   # the RCCL path (uint8 all-gather) at N = 1
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*line*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    oc = (d.get("extra") or {}).get("other_configs") or {}
    print(f, d.get("value"), d.get("unit"), (d.get("extra") or {}).get("bf16_throughput_mode", {}).get("value"),
          d.get("loop_only_value"), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"),
          {k: v.get("value", v.get("error")) for k, v in oc.items()}, "status", d.get("status_flags"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
