#!/usr/bin/env python3
"""Idle-gap analysis of one captured sampling pass from a rocprofv3 kernel trace.
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 1 --warmup 0
   python tools/gap_analysis.py DIR > summary.json
Takes the kernels after the last randn_kernel up to the first host round trip (a gap > 5 ms: bench.py's result checks and its
per-op event timing follow the pass; inside a graph-launched pass no gap comes near that), i.e. one sampling pass, and reports
wall time, the union of kernel intervals (busy), the idle remainder, and the per-kernel-name totals."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    last = max(i for i, r in enumerate(rows) if "randn_kernel" in r[2])
    rows = rows[last:]
    # ... up to the last kernel of this library: what follows are bench.py's own result checks (torch reductions with host
    # round trips between them), not part of the pass
    own = lambda n: not (n.startswith("at::native") or n.startswith("void at::native") or n.startswith("__amd_rocclr"))
    end = max(i for i, r in enumerate(rows) if own(r[2]))
    rows = rows[:end + 1]
    cur = rows[0][1]
    for i, (s0, e0, nm) in enumerate(rows):
        if i > 1000 and s0 - cur > 5_000_000 and not own(nm):      # first host round trip INTO torch code after the pass: the checks
            rows = rows[:i]
            break
        cur = max(cur, e0)
    while not own(rows[-1][2]):                          # trailing torch kernels of the check that caused the round trip
        rows.pop()
    wall = rows[-1][1] - rows[0][0]
    busy, cur_end, gaps = 0, rows[0][0], []
    before, after, prev = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0]), rows[0][2]
    for s, e, nm in rows:
        if s > cur_end:
            gaps.append(s - cur_end)
            if s - cur_end > 2000:            # > 2 us: who is on either side of the bubble?
                kb = prev.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                ka = nm.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                before[kb][0] += 1; before[kb][1] += s - cur_end
                after[ka][0] += 1; after[ka][1] += s - cur_end
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
        prev = nm
    per = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        per[k][0] += 1
        per[k][1] += e - s
    gaps.sort()
    # (r06, r05 verdict next 7) the SAMPLING LOOP alone -- the kernels from the first to the last sampler_step_kernel of the pass, i.e. the
    # replayed step graphs (denoiser forward + update + step counter) without the hoisted pre-passes' first part and without the decode:
    # `forwards` = number of update kernels in the window (+ 1: the first forward lies before the first update), per-kernel-name totals and the
    # GEMM family's total -- bench.py divides its own algorithmic FLOPs per forward by (family_ms / forwards) for `roofline.trace`
    steps = [i for i, r in enumerate(rows) if "sampler_step_kernel" in r[2]]
    loop = {}
    if len(steps) >= 2:
        lrows = rows[steps[0] + 1: steps[-1] + 1]
        lper = defaultdict(lambda: [0, 0])
        for s, e, n in lrows:
            k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            lper[k][0] += 1
            lper[k][1] += e - s
        fam = ("igemm_kernel", "conv3x3_patch_kernel", "conv3x3_gn_kernel")
        nf = len(steps) - 1
        loop = {"forwards": nf, "wall_ms": (lrows[-1][1] - lrows[0][0]) / 1e6, "kernels": len(lrows),
                "kernels_per_forward": round(len(lrows) / nf, 2), "forward_ms_wall": (lrows[-1][1] - lrows[0][0]) / 1e6 / nf,
                "gemm_family_ms": sum(v[1] for k, v in lper.items() if k in fam) / 1e6,
                "gemm_family_launches": sum(v[0] for k, v in lper.items() if k in fam),
                "gemm_family_ms_per_forward": sum(v[1] for k, v in lper.items() if k in fam) / 1e6 / nf,
                "per_kernel_ms": {k: [v[0], round(v[1] / 1e6, 3)] for k, v in sorted(lper.items(), key=lambda kv: -kv[1][1])}}
    out = {"kernels": len(rows), "wall_ms": wall / 1e6, "busy_ms": busy / 1e6, "idle_ms": (wall - busy) / 1e6,
           "idle_frac": (wall - busy) / wall, "sum_dur_ms": sum(e - s for s, e, _ in rows) / 1e6,
           "gap_us_median": gaps[len(gaps) // 2] / 1e3 if gaps else 0, "gap_us_p90": gaps[int(len(gaps) * .9)] / 1e3 if gaps else 0,
           "n_gaps": len(gaps), "gaps_gt_5ms": [round(g / 1e6, 2) for g in gaps if g > 5_000_000],
           "bubbles_gt2us_by_preceding_kernel_ms": {k: [v[0], round(v[1] / 1e6, 3)] for k, v in sorted(before.items(), key=lambda kv: -kv[1][1])[:8]},
           "bubbles_gt2us_by_following_kernel_ms": {k: [v[0], round(v[1] / 1e6, 3)] for k, v in sorted(after.items(), key=lambda kv: -kv[1][1])[:8]},
           "per_kernel_ms": {k: [v[0], round(v[1] / 1e6, 3)] for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])},
           "sampling_loop": loop}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
