#!/bin/bash
# Round profile collection on the GPU box (writes summaries under gpurun_out/; copy the ones to keep into profiles/):
#   kernel-trace stats + gap analysis of bench.py, PMC passes (SQ / GRBM / TCC, each in its own run) over
#   tools/profile_forward.py, and an rocm-smi clock / power log taken while bench.py runs.
TAG=${1:-r06_x3}
PREC=${2:-bf16x3}      # arithmetic profiled: bf16x3 (the parity / headline mode) or bf16
ONLY=${3:-all}         # all | trace (kernel trace + gap analysis of the bench command only, on the pinned tiles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
if [ "$ONLY" = all ]; then
export FRIDO_TUNE_CACHE=/tmp/frido_tune_prof.json
python $R/tools/profile_forward.py --precision $PREC --decode --top 60 > $OUT/${TAG}_forward_per_op.txt 2>&1          # also fills the tile cache
# ---- PMC passes (counter collection only: no trace domains next to --pmc) ----
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d /tmp/pmc_sq -- python $R/tools/profile_forward.py --precision $PREC > /tmp/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_grbm -- python $R/tools/profile_forward.py --precision $PREC > /tmp/pmc_grbm.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python $R/tools/profile_forward.py --precision $PREC > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python $R/tools/profile_forward.py --precision $PREC > /tmp/pmc_w.log 2>&1
python $R/tools/pmc_sq.py /tmp/pmc_sq /tmp/pmc_grbm /tmp/pmc_fetch /tmp/pmc_write > $OUT/${TAG}_pmc_mfma.json 2> $OUT/${TAG}_pmc_mfma.err
python $R/tools/pmc_traffic.py /tmp/pmc_fetch > /tmp/tr_f.json 2>/dev/null
mkdir -p /tmp/pmc_both; cp -r /tmp/pmc_fetch /tmp/pmc_both/f; cp -r /tmp/pmc_write /tmp/pmc_both/w
python $R/tools/pmc_traffic.py /tmp/pmc_both > $OUT/${TAG}_pmc_traffic.json 2> $OUT/${TAG}_pmc_traffic.err
tail -3 /tmp/pmc_sq.log > $OUT/${TAG}_pmc_logs.txt; tail -3 /tmp/pmc_grbm.log >> $OUT/${TAG}_pmc_logs.txt
# (r06) L2 behaviour per kernel INSTANCE (template instantiation x grid): hit rate and bytes fetched from MALL / HBM per launch
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_l2 -- python $R/tools/profile_forward.py --precision $PREC > /tmp/pmc_l2.log 2>&1
python $R/tools/pmc_instances.py /tmp/pmc_l2 /tmp/pmc_grbm --top 45 > $OUT/${TAG}_pmc_l2_instances.json 2> $OUT/${TAG}_pmc_l2_instances.err
fi
# ---- kernel trace of the bench command (the headline workload only: the other configs of the default line stay out of the window) ----
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktrace -- python $R/bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > $OUT/${TAG}_bench_under_trace.json 2> /tmp/ktrace.log
cp $(ls /tmp/ktrace/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python $R/tools/gap_analysis.py /tmp/ktrace > $OUT/${TAG}_gap_analysis.json 2> $OUT/${TAG}_gap.err
[ "$ONLY" = trace ] && exit 0
# ---- clocks / power while the bench runs ----
(python $R/bench.py --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline --no-bf16-extra --no-other-configs > $OUT/${TAG}_bench_smi_run.json 2>/dev/null) &
BP=$!
: > $OUT/${TAG}_smi_log.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | tr -d '\n' >> $OUT/${TAG}_smi_log.txt; echo >> $OUT/${TAG}_smi_log.txt
  sleep 1
done
wait $BP
