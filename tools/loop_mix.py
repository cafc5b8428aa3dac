#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop: python tools/loop_mix.py <file.s> <mangled-name-substring>
Finds the backward branch whose body holds the most MFMAs and counts instruction classes inside it."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo, hi = labels[m.group(1)], i
        n = sum(1 for x in body[lo:hi] if "v_mfma" in x)
        if best is None or n > best[0]:
            best = (n, lo, hi)
n, lo, hi = best
mix = collections.Counter()
for l in body[lo:hi]:
    t = l.strip().split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    cls = ("mfma" if "mfma" in op else "ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_"))
           else "salu" if op.startswith("s_") else "valu")
    mix[cls] += 1
    if cls in ("valu", "salu"):
        mix[op] += 1
print(f"loop of {hi - lo} lines, {n} MFMAs")
for k, v in mix.most_common(40):
    print(f"  {k:28s} {v}")
