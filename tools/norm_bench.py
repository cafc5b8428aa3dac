#!/usr/bin/env python3
"""Time the GroupNorm / LayerNorm launches of the bf16x3 sampler on their real shapes (B = 16) and print the achieved HBM rate
(bytes = what the op must move once).   python tools/norm_bench.py [--old]      FRIDO_LIB selects the library under test."""
import sys

import torch

sys.path.insert(0, ".")
from frido_amd.builder import ACT_SILU, Builder  # noqa: E402

DEV = torch.device("cuda:0")
B = 16


def timed(b, reps=30):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        b.prog.run(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.prog.run(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def gn(C1, C2, HW, spade, stats_only=False):
    C = C1 + C2
    w = {"n.weight": torch.ones(C, device=DEV), "n.bias": torch.zeros(C, device=DEV)}
    b = Builder(DEV, 2, w)
    f1 = b.f32(B * HW, C1)
    f1.view().normal_()
    f2 = None
    if C2:
        f2 = b.f32(B * HW, C2)
        f2.view().normal_()
    g = be = None
    if spade:
        g, be = b.f32(B * HW, C), b.f32(B * HW, C)
        g.view().normal_()
        be.view().normal_()
    b.groupnorm(f1, f2, B, HW, "n", 1e-5, gamma=g, beta=be, act=ACT_SILU, want_raw=False)
    names = [op[0] for op in b.prog.ops]
    us = timed(b)
    per = {}
    ops = list(b.prog.ops)
    for i, op in enumerate(ops):     # each launch alone
        b.prog.ops = [op]
        b.prog._packed = None
        per[f"{op[0]}#{i}"] = timed(b)
    b.prog.ops = ops
    b.prog._packed = None
    byt = B * HW * C * (4 + 4 + (8 if spade else 0))
    print(f"GN  C={C1}+{C2} HW={HW} spade={int(spade)}: {us:7.1f} us total  {names}  apply-bytes {byt / 1e6:.0f} MB -> {byt / us / 1e6:.2f} TB/s   each: "
          + ", ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in per.items()))


def ln(M, C):
    w = {"n.weight": torch.ones(C, device=DEV), "n.bias": torch.zeros(C, device=DEV)}
    b = Builder(DEV, 2, w)
    f = b.f32(M, C)
    f.view().normal_()
    b.layernorm(f, "n")
    us = timed(b)
    byt = M * C * 8
    print(f"LN  M={M} C={C}: {us:7.1f} us  {byt / 1e6:.0f} MB -> {byt / us / 1e6:.2f} TB/s")


if __name__ == "__main__":
    for C1, C2, HW in ((192, 0, 4096), (192, 192, 4096), (384, 192, 4096), (384, 0, 1024), (384, 384, 1024), (576, 384, 1024)):
        for sp in (False, True):
            gn(C1, C2, HW, sp)
    for M, C in ((16384, 384), (4096, 576), (1024, 960)):
        ln(M, C)
