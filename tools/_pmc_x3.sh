# PMC passes over ONE bf16x3 GEMM launch shape (separate rocprofv3 --pmc runs per counter group; no trace domains):
#   tools/_pmc_x3.sh "conv 16 32 32 384 384" 7      -> matrix-pipe busy share, wave wait states, LDS conflicts, L2 hit rate
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${1:-conv 16 32 32 384 384}; TILE=${2:-7}; NS=${3:-2}
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  d=/tmp/pmcx_$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  rocprofv3 --pmc $C --output-format csv -d $d -- python $R/tools/gemm_bench.py $SHAPE $NS $TILE > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  python - "$d" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "igemm_kernel" in k or "conv3x3_patch" in k:
            agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"{k:28s} per launch {v / n:16.0f}   launches {n}")
PY
done
