#!/bin/bash
# End-of-round GPU run.  FIRST the bench lines that (re)write the pinned tile cache for this library (profiles/tune_cache.json, keyed by
# the library's content hash: B = 16 headline incl. the config-3 / config-5 passes of `other_configs`), then (r06) the IN-CONTEXT tile pass
# over the B = 16 step programs (tools/tune_in_context.py: per-op timings of the whole forward decide, kept only where they win by 2 %),
# then B = 32 (config 4's per-GPU shard) and B = 1 at full width; THEN the full GPU suite, which reads that cache (tests/conftest.py): the
# parity tests run on the tiles the benchmark runs on.  Then the lines of record, the profiles and smoke().   tools/final_run.sh r06 [quick]
TAG=${1:-r06}; MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
rm -f profiles/tune_cache.json $OUT/e2e_error.json
python bench.py --retune --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_line_tune.json 2> $OUT/${TAG}_bench_line.err      # two-plane headline + bf16 extra + configs 3 / 5
cp profiles/tune_cache.json $OUT/tune_cache_microbench.json
FRIDO_TUNE_CACHE=$R/profiles/tune_cache.json FRIDO_TUNE_CACHE_READONLY=1 python tools/tune_in_context.py --reps 5 --min-gain 0.02 \
    --out $OUT/${TAG}_tune_in_context.json --write-cache $OUT/tune_cache_ctx.json > $OUT/${TAG}_tune_in_context.log 2>&1
tail -3 $OUT/${TAG}_tune_in_context.log
python - <<PY
import json, shutil
r = json.load(open("$OUT/${TAG}_tune_in_context.json"))
if r["forward_ms_after"] < r["forward_ms_before"] and r["changes"]:
    shutil.copy("$OUT/tune_cache_ctx.json", "profiles/tune_cache.json")
    print("in-context tiles adopted:", len(r["changes"]), "signatures,", r["forward_ms_before"], "->", r["forward_ms_after"], "ms")
else:
    print("in-context pass changed nothing")
PY
python bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline --no-bf16-extra --retune > $OUT/${TAG}_bench_line_batch32.json 2>/dev/null   # config 4's per-GPU shard (the cache gains its signatures)
python bench.py --steps 1 --warmup 0 --batch 1 --ddim-steps 4 --no-cpu-baseline --no-bf16-extra --retune > /dev/null 2>&1   # B = 1 at full width: the shapes of the config-1 / config-2-step goldens and of the rows-vs-B=1 tests
cp profiles/tune_cache.json $OUT/tune_cache.json
(time python -m pytest tests -m gpu -q -s --durations=15 > $OUT/final_gpu_tests.log 2>&1) 2> $OUT/final_gpu_tests.time; tail -6 $OUT/final_gpu_tests.log; cat $OUT/final_gpu_tests.time
cp $OUT/e2e_error.json $OUT/${TAG}_e2e_error.json 2>/dev/null
if [ "$MODE" = "full" ]; then
python bench.py --gpus 1 --steps 5 --warmup 1 > $OUT/${TAG}_bench_line.json 2>/dev/null     # second process: pinned tiles, no tuning, EVERY default leg (CPU baseline, bf16 extra, configs 3 / 5): the driver's spelling
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > $OUT/${TAG}_bench_line_torchrun_n1.json 2>/dev/null   # the RCCL path (uint8 all-gather) at N = 1
FRIDO_GRAPH_STEPS=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > $OUT/${TAG}_bench_line_graph_steps1.json 2>/dev/null
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*line*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d.get("value"), d.get("unit"), (d.get("extra") or {}).get("bf16_throughput_mode", {}).get("value"),
          d.get("loop_only_value"), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"),
          d.get("other_configs_images_per_s"), "status", d.get("status_flags"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
