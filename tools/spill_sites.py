#!/usr/bin/env python3
"""Where does a kernel spill?  python tools/spill_sites.py <file.s> <mangled-name-substring>
Lists every scratch_load / scratch_store of the kernel with the number of MFMAs that precede it in program order (spills before the
first or after the last MFMA are prologue / epilogue; anything in between sits in the main loop)."""
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
nm = 0
sites = []
for i, l in enumerate(body):
    if "v_mfma" in l:
        nm += 1
    if "scratch_" in l:
        sites.append((i, nm, l.strip().split(";")[0]))
print(f"{len(body)} lines, {nm} MFMAs, {len(sites)} scratch ops")
for i, k, l in sites:
    print(f"  line {i:6d}  after {k:4d} MFMAs  {l}")
