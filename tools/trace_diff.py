#!/usr/bin/env python3
"""Which kernels moved between two rocprofv3 kernel traces of the same command (e.g. the shipped library vs a stagger build):
   rocprofv3 --kernel-trace --output-format csv -d DIR_A -- python bench.py --steps 1 --warmup 1 ...     (and DIR_B with the other setting)
   python tools/trace_diff.py DIR_A DIR_B [top]
Groups launches by (kernel name, grid size, workgroup size) -- one GEMM template instantiation serves many shapes, the grid tells them
apart -- and prints the groups with the largest change of TOTAL time, plus the totals.  CPU-only post-processing."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d):
    agg = defaultdict(lambda: [0, 0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                grid = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
                wg = tuple(int(r.get(k, 0) or 0) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
                n_wg = 1
                for g, w in zip(grid, wg):
                    n_wg *= max(1, g // max(1, w))
                a = agg[(r["Kernel_Name"], n_wg, wg[0])]
                a[0] += 1
                a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg


def short(name, n=70):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    keys = set(a) | set(b)
    rows = []
    for k in keys:
        ca, ta = a.get(k, (0, 0))
        cb, tb = b.get(k, (0, 0))
        rows.append((tb - ta, k, ca, ta, cb, tb))
    ta_all, tb_all = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
    print(f"total kernel time: A {ta_all / 1e6:.1f} ms   B {tb_all / 1e6:.1f} ms   B - A {(tb_all - ta_all) / 1e6:+.1f} ms ({100.0 * (tb_all - ta_all) / max(ta_all, 1):+.2f} %)")
    print(f"{'delta ms':>9} {'A ms':>9} {'B ms':>9} {'A us/launch':>11} {'B us/launch':>11} {'launches':>9} {'wgs':>7} {'thr':>4}  kernel")
    for dlt, (name, n_wg, thr), ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0]))[:top]:
        ua = ta / ca / 1e3 if ca else 0.0
        ub = tb / cb / 1e3 if cb else 0.0
        print(f"{dlt / 1e6:9.2f} {ta / 1e6:9.2f} {tb / 1e6:9.2f} {ua:11.1f} {ub:11.1f} {max(ca, cb):9d} {n_wg:7d} {thr:4d}  {short(name)}")


if __name__ == "__main__":
    main()
