# TCC traffic / hit rate of ONE small-M conv launch shape (8x8 plane, 960 -> 960, 256x128k64 split-K 8): is the ring kernel's
# ~10 B/clk/CU operand stream coming out of L2, or out of MALL / HBM?
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  d=/tmp/pmc_$(echo $C | tr ' ' '_')
  SPLITK=8 rocprofv3 --pmc $C --output-format csv -d $d -- python $R/tools/gemm_bench.py conv 16 8 8 960 960 1 17 > /tmp/pmc.log 2>&1
  python - "$d" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "igemm_kernel" in k or "splitk" in k:
            key = ("igemm" if "igemm" in k else "reduce", r["Counter_Name"])
            agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
for k, (v, n) in sorted(agg.items()):
    print(k, "per launch", v / n, "launches", n)
PY
done
