#!/usr/bin/env python3
"""ISA audit of the hand-issued staging loads of conv3x3_gn_kernel (csrc/convgn.hip): between an inline-asm `global_load_dwordx4 vD, ...`
and the next `s_waitcnt vmcnt(...)`, NO instruction may read or write vD -- the compiler believes the value is there as soon as the asm
statement is over, so a live-range split / spill it places in that window would copy STALE registers (the load lands later).
    hipcc ... --cuda-device-only -S convgn.hip -o convgn.s ; python tools/asm_inflight_check.py convgn.s"""
import re
import sys

if len(sys.argv) < 2:
    sys.exit(__doc__)
src = open(sys.argv[1]).read().splitlines()
CARRY = len(sys.argv) > 2 and sys.argv[2] == "carry"
fn = None
inflight = {}          # reg index -> (line no, text)
bad = []
stats = {}
reg_re = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in reg_re.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


queue = []             # outstanding vector-memory operations of the (linearly scanned) wave, oldest first: (dest regs or empty set, line, text)
for ln, line in enumerate(src, 1):
    t = line.strip()
    m = re.match(r"(_Z\w*conv3x3_gn_kernel\w*):", t)
    if m:
        fn = m.group(1)
        queue = []
        stats[fn] = [0, 0]
        continue
    if t.startswith(".Lfunc_end"):
        fn = None
    if fn is not None and re.match(r"\.LBB\d+_\d+:", t):
        # basic-block boundary: a linear scan cannot follow the control flow; the loads of this kernel are consumed far away, so an in-flight
        # set is carried across FALL-THROUGH labels only when CARRY=1 (then joins of other paths show up as false positives)
        if not CARRY:
            queue = []
        continue
    if fn is None or not t or t.startswith((";", ".", "//")):
        continue
    code = t.split(";")[0].strip()
    if not code:
        continue
    op = code.split()[0]
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", code)
        if m:
            n = int(m.group(1))
            queue = queue[len(queue) - n:] if n else []
        continue
    if op == "s_endpgm":
        queue = []
        continue
    inflight = {}
    for regs, l0, c0 in queue:
        for r in regs:
            inflight[r] = (l0, c0)
    # in CARRY mode only READS count (a write on a joining path is the other side of a phi, not a hazard): operands after the first comma;
    # stores read all their operands
    ops_part = code if op.startswith(("global_store", "scratch_store", "buffer_store", "ds_write")) or not CARRY else (code.split(",", 1)[1] if "," in code else "")
    hit = regs_of(ops_part) & set(inflight)
    if hit:
        stats[fn][1] += 1
        bad.append((fn, ln, code, inflight[sorted(hit)[0]]))
    if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "scratch_load", "scratch_store", "flat_load", "flat_store")):
        dest = set()
        m = re.match(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", code)
        if m and "lds" not in code:
            stats[fn][0] += 1
            dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
        queue.append((dest, ln, code))
for fn, (n, b) in stats.items():
    print(f"{fn[:90]}: {n} hand-issued loads, {b} instructions touching an in-flight destination")
for fn, ln, code, (l0, c0) in bad[:20]:
    print(f"  line {ln}: {code}    <- in flight since line {l0}: {c0}")
# exit status (r06: `make -C frido_amd/csrc audit`, part of `all`, and tests/test_host.py run this on every build): non-zero when a hazard
# was found, or when no hand-issued load was seen at all (the audit would then be looking at the wrong thing)
if bad or not stats or any(n == 0 for n, _ in stats.values()):
    sys.exit(1)
