#!/usr/bin/env python3
"""Where does a kernel spill?  tools/spill_sites.py file.s [name-filter]: per kernel, every scratch_ access with its position
relative to the first / last v_mfma of the function (spills outside the MFMA range sit in the prologue / epilogue)."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
filt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
starts.append((len(lines), None))
for (a, name), (b, _) in zip(starts, starts[1:]):
    if filt not in name:
        continue
    body = lines[a:b]
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    sc = [(i, l.strip()) for i, l in enumerate(body) if 'scratch_' in l]
    if not sc:
        continue
    inside = [x for x in sc if mf and mf[0] < x[0] < mf[-1]]
    print(f"{name}: {len(body)} lines, mfma lines {mf[0] if mf else None}..{mf[-1] if mf else None} ({len(mf)}), "
          f"scratch ops {len(sc)}, between first and last mfma {len(inside)}")
    for i, l in sc:
        print('   ', i, l[:70])
