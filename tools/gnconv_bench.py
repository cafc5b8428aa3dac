#!/usr/bin/env python3
"""Fused GroupNorm + 3x3 conv (csrc/convgn.inc) against the pair it replaces (gn_apply + tuned ring conv), one shape at a time,
per-op HIP events (median of 20):   python tools/gnconv_bench.py B H W C1 C2 Cout spade(0/1) skipC [tile] [splitk]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import _lib  # noqa: E402
from frido_amd.builder import Builder, ACT_SILU  # noqa: E402
from frido_amd.engine import require_gpu  # noqa: E402


def timed(prog, sp, reps=20):
    prog.run(sp)
    acc = None
    runs = []
    for _ in range(reps):
        runs.append(prog.run_timed(sp))
    med = [sorted(r[i] for r in runs)[reps // 2] for i in range(len(prog.ops))]
    return med


def main():
    B, H, W, C1, C2, Cout, spade, Cr = map(int, sys.argv[1:9])
    dev = require_gpu("cuda:0")
    C_ = C1 + C2
    M, HW = B * H * W, H * W
    w = {"n.weight": torch.ones(C_, device=dev), "n.bias": torch.zeros(C_, device=dev),
         "c.weight": torch.randn(Cout, C_, 3, 3, device=dev) * 0.02, "c.bias": torch.zeros(Cout, device=dev)}
    if Cr:
        w.update({"s.weight": torch.randn(Cout, Cr, 1, 1, device=dev) * 0.05, "s.bias": torch.zeros(Cout, device=dev)})
    b = Builder(dev, 2, w)
    f1 = b.f32(M, C1); f1.view().normal_()
    f2 = None
    if C2:
        f2 = b.f32(M, C2); f2.view().normal_()
    g = be = None
    if spade:
        g, be = b.f32(M, C_), b.f32(M, C_)
        g.view().normal_(); be.view().normal_()
    fr = None
    if Cr:
        fr = b.f32(M, Cr); fr.view().normal_()
    sp = torch.cuda.current_stream().cuda_stream
    flops = 2.0 * M * Cout * (9 * C_ + Cr)
    # (a) the two-kernel path
    prog_a = b.new_prog()
    a, raw = b.groupnorm(f1, f2, B, HW, "n", 1e-5, gamma=g, beta=be, act=ACT_SILU)
    if Cr:
        raw_op = b.pack(fr.ptr, 1, M, Cr, 0, Cr)      # (in the model gn_apply emits it as a side output: not charged here)
        b.conv_plus_skip(a, raw_op, B, H, W, "c", "s")
    else:
        b.conv(a, B, H, W, "c")
    names = {v: k[len("FRIDO_OP_"):] for k, v in _lib.OP_KINDS.items()}
    ms = timed(prog_a, sp)
    tot = 0.0
    print(f"== B={B} {H}x{W} C={C1}+{C2} -> {Cout}, spade={spade}, skip={Cr}: {flops / 1e9:.1f} GFLOP")
    for (kind, st), t in zip(prog_a.ops, ms):
        if names[kind] == "PACK":
            continue
        extra = f" tile {st.tile} splitk {st.splitk}" if names[kind] == "GEMM" else ""
        print(f"   two-kernel  {names[kind]:10s} {t * 1e3:8.1f} us{extra}")
        tot += t
    print(f"   two-kernel  total      {tot * 1e3:8.1f} us   conv alone {flops / ms[-1] / 1e9:7.1f} TF/s   pair {flops / tot / 1e9:7.1f} TF/s")
    # (b) fused
    for tile in ([int(sys.argv[9])] if len(sys.argv) > 9 else [20, 21]):
        prog_b = b.new_prog()
        try:
            b.gn_conv(tile, f1, f2, B, H, W, "n", 1e-5, "c", gamma=g, beta=be, act=ACT_SILU, skip=(fr, None, "s") if Cr else None,
                      splitk=int(sys.argv[10]) if len(sys.argv) > 10 else 1)
            ms = timed(prog_b, sp)
        except _lib.FridoHipError as e:
            print(f"   fused tile {tile}: n/a ({str(e)[-60:]})")
            continue
        tot = sum(ms)
        for (kind, st), t in zip(prog_b.ops, ms):
            print(f"   fused t{tile}   {names[kind]:10s} {t * 1e3:8.1f} us" + (f" (incl. splitk_reduce, {st.splitk} slices)" if names[kind] == "GEMM" and st.splitk > 1 else ""))
        print(f"   fused t{tile}   total      {tot * 1e3:8.1f} us   {flops / tot / 1e9:7.1f} TF/s (conv alone {flops / ms[-1] / 1e9:7.1f})")


if __name__ == "__main__":
    main()
