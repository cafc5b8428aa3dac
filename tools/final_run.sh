#!/bin/bash
# End-of-round GPU run: full GPU test suite, then the bench lines of record (tile cache written to profiles/tune_cache.json).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
(time python -m pytest tests -m gpu -q -s --durations=10 > $OUT/final_gpu_tests.log 2>&1); tail -4 $OUT/final_gpu_tests.log
rm -f profiles/tune_cache.json
python bench.py --steps 3 --warmup 1 --cpu-ddim50 > $OUT/r02_bench_line.json 2> $OUT/r02_bench_line.err
cp profiles/tune_cache.json $OUT/tune_cache.json
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-mode > $OUT/r02_bench_line_repeat.json 2>/dev/null     # second process: pinned tiles, no tuning
python - <<'PY'
import json
for f in ("gpurun_out/r02_bench_line.json", "gpurun_out/r02_bench_line_repeat.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, d["value"], d["unit"], d.get("parity_mode", {}).get("value"), d.get("loop_only_value"), d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
