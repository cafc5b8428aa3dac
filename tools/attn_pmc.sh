#!/bin/bash
# LDS bank-conflict share of the attention kernels PER INSTANTIATION (tools/pmc_sq.py folds the template arguments away), over one
# denoiser forward of the sampler:   bash tools/attn_pmc.sh [tag]   -> gpurun_out/<tag>_attn_lds_conflicts.txt
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_attn -- python $R/tools/profile_forward.py --precision bf16x3 > /tmp/pmc_attn.log 2>&1
python - > $OUT/${TAG}_attn_lds_conflicts.txt <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob("/tmp/pmc_attn/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(fn)))
    last = max((int(r["Dispatch_Id"]) for r in rows if "randn_kernel" in r["Kernel_Name"]), default=-1)
    for r in rows:
        if int(r["Dispatch_Id"]) <= last:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "attn" not in k and "flash" not in k and "gn_apply" not in k and "gn_stats_kernel" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k] += r["Counter_Name"] == "SQ_WAVE_CYCLES"
print("# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel instantiation, one denoiser forward x 10 (tools/profile_forward.py)")
for k, v in sorted(agg.items()):
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
    print(f"{k:44s} launches {n[k]:5d}  conflict cycles {c:14.0f}  LDS-active cycles {a:14.0f}  share {c / a if a else 0:.4f}  LDS-active / wave cycles {a / max(v.get('SQ_WAVE_CYCLES', 1), 1):.4f}")
PY
cat $OUT/${TAG}_attn_lds_conflicts.txt
