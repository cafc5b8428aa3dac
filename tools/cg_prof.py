#!/usr/bin/env python3
"""Where do the waves of the fused GroupNorm + conv kernel spend their cycles?  Runs ONE fused launch on a library built with -DCG_PROF=1
-DCG_MIDBAR=0 (tools/build_variants.sh prof "-DCG_PROF=1 -DCG_MIDBAR=0") and prints the per-wave cycle sums the kernel leaves behind
(convgn.hip CG_PROF).   FRIDO_LIB=$PWD/tools/ablate/libfrido_prof.so python tools/cg_prof.py B H W C Cout spade(0/1) skipC [tile]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import _lib  # noqa: E402
from frido_amd.builder import Builder, ACT_SILU  # noqa: E402
from frido_amd.engine import require_gpu  # noqa: E402

NAMES = ["wait weights", "in barrier", "DMA issue + staging", "reads + MFMAs", "loop", "prologue", "epilogue"]


def main():
    B, H, W, C1, Cout, spade, Cr = map(int, sys.argv[1:8])
    tile = int(sys.argv[8]) if len(sys.argv) > 8 else (20 if W == 64 else 21)
    dev = require_gpu("cuda:0")
    M = B * H * W
    w = {"n.weight": torch.ones(C1, device=dev), "n.bias": torch.zeros(C1, device=dev),
         "c.weight": torch.randn(Cout, C1, 3, 3, device=dev) * 0.02, "c.bias": torch.zeros(Cout, device=dev)}
    if Cr:
        w.update({"s.weight": torch.randn(Cout, Cr, 1, 1, device=dev) * 0.05, "s.bias": torch.zeros(Cout, device=dev)})
    b = Builder(dev, 2, w)
    f1 = b.f32(M, C1); f1.view().normal_()
    g = be = fr = None
    if spade:
        g, be = b.f32(M, C1), b.f32(M, C1)
        g.view().normal_(); be.view().normal_()
    if Cr:
        fr = b.f32(M, Cr); fr.view().normal_()
    sp = torch.cuda.current_stream().cuda_stream
    prog = b.new_prog()
    b.gn_conv(tile, f1, None, B, H, W, "n", 1e-5, "c", gamma=g, beta=be, act=ACT_SILU, skip=(fr, None, "s") if Cr else None)
    for _ in range(3):
        prog.run(sp)
    ts = [prog.run_timed(sp) for _ in range(10)]
    torch.cuda.synchronize()
    L = _lib.lib()
    bm = 256 if tile == 20 else 128
    nwg = (M // bm) * (Cout // 192)
    n = min(nwg, 2048) * 8 * 8
    buf = (C.c_uint * n)()
    L.frido_cg_prof_read.restype = C.c_int
    rc = L.frido_cg_prof_read(buf, n)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 8, 8).astype(np.float64)
    steps = 9 * (C1 // 32) + Cr // 32
    conv_us = sorted(t[-1] for t in ts)[len(ts) // 2] * 1e3
    print(f"== B={B} {H}x{W} C={C1} -> {Cout} spade={spade} skip={Cr} tile {tile}: {nwg} workgroups, {steps} k-steps, fused conv launch {conv_us:.1f} us (median of 10, per-op events)")
    tot = a[:, :, 4].mean()
    print(f"   mean over all waves (cycles; per k-step in brackets; share of the loop):")
    for i in (0, 1, 2, 3):
        print(f"     {NAMES[i]:22s} {a[:, :, i].mean():10.0f}  [{a[:, :, i].mean() / steps:7.0f}]  {a[:, :, i].mean() / tot * 100:5.1f} %")
    for i in (4, 5, 6):
        print(f"     {NAMES[i]:22s} {a[:, :, i].mean():10.0f}  [{a[:, :, i].mean() / steps:7.0f}]")
    whole = a[:, :, 4] + a[:, :, 5] + a[:, :, 6]
    print(f"   kernel per wave: {whole.mean():.0f} cycles = {conv_us:.1f} us -> {whole.mean() / conv_us / 1e3:.2f} GHz if the launch were all of it")
    print("   early waves (0-3) vs late (4-7), per k-step:")
    for i in (0, 1, 2, 3):
        print(f"     {NAMES[i]:22s} early {a[:, :4, i].mean() / steps:7.0f}   late {a[:, 4:, i].mean() / steps:7.0f}")
    mf = 72 * 16
    print(f"   a wave's own MFMAs per k-step: 72 x 16 = {mf} cycles; both waves of a SIMD: {2 * mf}")
    print("   workgroup 0, per wave:", " | ".join(f"w{w_}: " + "/".join(f"{a[0, w_, i] / steps:.0f}" for i in (0, 1, 2, 3)) for w_ in range(8)))


if __name__ == "__main__":
    main()
