#!/usr/bin/env python3
"""Per-INSTANCE view of rocprofv3 --pmc counter_collection CSVs: counters summed per (kernel template instantiation, grid size) over the
dispatches after the last x_T draw (like tools/pmc_sq.py), every counter divided by the instance's launch count, plus derived cache figures
where their inputs are present:
    l2_hit_rate       = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
    ea_read_MB        = TCC_EA0_RDREQ_sum x 64 B (+ 32-B requests counted at 32) per launch: what the XCDs' L2s fetched from MALL / HBM
    l2_req_MB         = TCC_REQ_sum x 128 B per launch (upper bound: a request moves up to one 128-B line)
    python tools/pmc_instances.py <dir> [<dir> ...] [--top 40] > profiles/rNN_pmc_instances.json"""
import collections
import csv
import glob
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in args:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(fn)))
        last = max((int(r["Dispatch_Id"]) for r in rows if "randn_kernel" in r["Kernel_Name"]), default=-1)
        for r in rows:
            if int(r["Dispatch_Id"]) <= last:
                continue
            full = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            k = f"{full} grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')}"
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    n = max(calls[k].values())
    rec = {"launches": n}
    for c, x in v.items():
        rec[c + "_per_launch"] = round(x / calls[k][c], 1)
    hit, miss = v.get("TCC_HIT_sum"), v.get("TCC_MISS_sum")
    if hit is not None and miss is not None and hit + miss > 0:
        rec["l2_hit_rate"] = round(hit / (hit + miss), 4)
    if v.get("TCC_EA0_RDREQ_sum") is not None:
        r32 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rec["ea_read_MB_per_launch"] = round(((v["TCC_EA0_RDREQ_sum"] - r32) * 64 + r32 * 32) / calls[k]["TCC_EA0_RDREQ_sum"] / 1e6, 2)
    if v.get("TCC_REQ_sum") is not None:
        rec["l2_req_MB_per_launch_upper"] = round(v["TCC_REQ_sum"] * 128 / calls[k]["TCC_REQ_sum"] / 1e6, 2)
    rec["_weight"] = sum(x for c, x in v.items() if c in ("TCC_REQ_sum", "TCC_HIT_sum", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES")) or n
    out[k] = rec
keep = dict(sorted(out.items(), key=lambda kv: -kv[1]["_weight"])[:top])
for r in keep.values():
    r.pop("_weight")
print(json.dumps(keep, indent=1))
