#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel HBM traffic per launch.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
(MI355X_MICROARCH.md §HBM), so reads are doubled.   python tools/pmc_traffic.py <dir> > profiles/rNN_pmc_traffic.json"""
import collections
import csv
import glob
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
# (r06) the same per INSTANCE = (full template name, grid size): which GEMM signature re-reads its operands (an XCD's L2 is 4 MB; a launch
# whose per-XCD working set exceeds it streams the weight panel from MALL / HBM once per tile row)
iagg = collections.defaultdict(lambda: collections.defaultdict(float))
icalls = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(fn)))
    # keep only the LAST sampling pass (everything after the final x_T draw): earlier dispatches are warm-up,
    # plan building and the tile autotuner
    last = max((int(r["Dispatch_Id"]) for r in rows if "randn_kernel" in r["Kernel_Name"]), default=-1)
    for r in rows:
        if int(r["Dispatch_Id"]) <= last:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = "igemm_kernel" if "igemm_kernel" in k else k.split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
        full = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        ik = f"{full} grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')}"
        iagg[ik][r["Counter_Name"]] += float(r["Counter_Value"])
        icalls[ik][r["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    n_f, n_w = calls[k].get("FETCH_SIZE", 0), calls[k].get("WRITE_SIZE", 0)
    if not n_f or not n_w:
        continue
    rd = 2.0 * v["FETCH_SIZE"] * 1024 / n_f
    wr = v["WRITE_SIZE"] * 1024 / n_w
    out[k] = dict(launches=n_f, read_bytes_per_launch=round(rd), write_bytes_per_launch=round(wr),
                  hbm_bytes_per_launch=round(rd + wr), note="reads = 2 x FETCH_SIZE (gfx950 correction)")
inst = {}
for k, v in iagg.items():
    n_f, n_w = icalls[k].get("FETCH_SIZE", 0), icalls[k].get("WRITE_SIZE", 0)
    if n_f and n_w:
        rd, wr = 2.0 * v["FETCH_SIZE"] * 1024 / n_f, v["WRITE_SIZE"] * 1024 / n_w
        inst[k] = dict(launches=n_f, read_MB_per_launch=round(rd / 1e6, 2), write_MB_per_launch=round(wr / 1e6, 2), total_MB=round((rd + wr) * n_f / 1e6, 1))
if inst:
    out["by_instance_top40"] = dict(sorted(inst.items(), key=lambda kv: -kv[1]["total_MB"])[:40])
print(json.dumps(out, indent=1, sort_keys=True))
