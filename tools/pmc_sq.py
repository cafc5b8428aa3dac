#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (any number of passes / directories) into per-kernel counter sums and
the derived ratios DESIGN.md quotes.   python tools/pmc_sq.py <dir> [<dir> ...] > profiles/rNN_pmc_mfma.json

Units (MI355X_MICROARCH.md §Per-instruction cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
per wave, summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the chip's 1024 SIMDs;
GRBM_GUI_ACTIVE = shader-clock cycles the dispatch was resident, summed over the 8 XCDs (cross-check: the patch-staged conv's
18.0e9 MFMA-busy cycles = its MFMA count x 16 cycles per v_mfma_f32_16x16x32_bf16; its GUI_ACTIVE / 8 = its average duration x
the ~2.3 GHz clock rocm-smi shows under load).  Hence
    mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)   share of SIMD-cycles the matrix pipe is busy
    wait_any_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES                              waves parked on s_waitcnt / s_barrier
    wait_inst_frac   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                         waves stalled at issue (dependency / pipe)
    active_frac      = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
    lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Only dispatches after the last randn_kernel (x_T draw) are kept, like tools/pmc_traffic.py."""
import collections
import csv
import glob
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(fn)))
        last = max((int(r["Dispatch_Id"]) for r in rows if "randn_kernel" in r["Kernel_Name"]), default=-1)
        for r in rows:
            if int(r["Dispatch_Id"]) <= last:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = k.split("(")[0].split("<")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1


def ratio(v, a, b, scale=1.0):
    return round(v[a] / (scale * v[b]), 4) if v.get(a) is not None and v.get(b) else None


out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_WAVE_CYCLES", 0))):
    n = max(calls[k].values())
    e = dict(launches=n, counters={c: round(x) for c, x in sorted(v.items())})
    # counters of different passes are sums over the same dispatch sequence, so their ratios are well defined
    e["mfma_busy_frac"] = ratio(v, "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", 128.0)
    if v.get("GRBM_GUI_ACTIVE"):
        e["avg_cycles_per_launch"] = round(v["GRBM_GUI_ACTIVE"] / 8.0 / calls[k]["GRBM_GUI_ACTIVE"])
    e["wait_any_frac"] = ratio(v, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    e["wait_inst_frac"] = ratio(v, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
    e["active_frac"] = ratio(v, "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
    e["lds_conflict_frac"] = ratio(v, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
    if v.get("FETCH_SIZE") and v.get("WRITE_SIZE"):
        e["hbm_bytes_per_launch"] = round((2.0 * v["FETCH_SIZE"] / calls[k]["FETCH_SIZE"] + v["WRITE_SIZE"] / calls[k]["WRITE_SIZE"]) * 1024)
    out[k] = e
print(json.dumps(out, indent=1))
