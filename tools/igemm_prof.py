#!/usr/bin/env python3
"""Where do the waves of the ring kernel (igemm_kernel, two-plane virtual-step loop) spend their cycles?  Runs ONE launch per tile on a library
built with -DIG_PROF=1 (tools/build_variants.sh igprof "-DIG_PROF=1") and prints the per-wave cycle sums the kernel leaves behind (igemm.hip IG_PROF).
   FRIDO_LIB=$PWD/tools/ablate/libfrido_igprof.so python tools/igemm_prof.py dense M N K tiles   |   ... conv B H W Cin Cout tiles   |   ... geglu M H K tiles
tiles: the virtual-step tiles of the two-plane mode: 1 (128x128), 3 (64x64), 4 (128x64), 6 (64x128), 7 (256x128, 8 waves), 18 (128x192 on 8 waves)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import _lib  # noqa: E402
from frido_amd.builder import Builder  # noqa: E402
from frido_amd.engine import require_gpu  # noqa: E402

GEO = {1: (128, 128, 4), 3: (64, 64, 4), 4: (128, 64, 4), 6: (64, 128, 4), 7: (256, 128, 8), 18: (128, 192, 8)}      # BM, BN, waves


def main():
    a = sys.argv[1:]
    dev = require_gpu("cuda:0")
    os.environ["FRIDO_TUNE"] = "0"
    from frido_amd import tune
    tune.ENABLED = False
    mode = a[0]
    if mode == "conv":
        B, H, W, Cin, Cout = map(int, a[1:6])
        rest = a[6:]
        b = Builder(dev, 2, {"c.weight": torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05, "c.bias": torch.zeros(Cout, device=dev)})
        x = torch.randn(B, H * W, Cin, device=dev)
        xo = b.pack(x.data_ptr(), 1, B * H * W, Cin, 0, Cin)
        b.conv(xo, B, H, W, "c")
        M, N, K = B * H * W, Cout, 9 * Cin
    elif mode == "geglu":
        M, Hh, K = map(int, a[1:4])
        rest = a[4:]
        b = Builder(dev, 2, {"w.weight": torch.randn(2 * Hh, K, device=dev) * 0.05, "w.bias": torch.zeros(2 * Hh, device=dev)})
        x = torch.randn(M, K, device=dev)
        xo = b.pack(x.data_ptr(), 1, M, K, 0, K)
        b.linear_geglu(xo, "w")
        N = 2 * Hh
    else:
        M, N, K = map(int, a[1:4])
        rest = a[4:]
        b = Builder(dev, 2, {"w.weight": torch.randn(N, K, device=dev) * 0.05, "w.bias": torch.zeros(N, device=dev)})
        x = torch.randn(M, K, device=dev)
        xo = b.pack(x.data_ptr(), 1, M, K, 0, K)
        b.linear(xo, "w")
    tiles = [int(t) for t in rest[0].split(",")] if rest else [7, 18, 1]
    flops = 2.0 * M * N * K
    sp = torch.cuda.current_stream().cuda_stream
    b.prog.run(sp)
    kind, st = b.prog.ops[-1]
    L = _lib.lib()
    L.frido_ig_prof_read.restype = C.c_int
    for tile in tiles:
        bm, bn, nw = GEO[tile]
        st.tile = tile
        reps = 10
        arr = _lib.pack_ops([(kind, st)] * reps)
        ms = (C.c_float * reps)()
        _lib.check(L.frido_run_timed(C.addressof(arr), reps, sp, ms), "run")
        torch.cuda.synchronize()
        t = sorted(ms)[reps // 2]
        nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        n = min(nwg, 4096) * 8 * 8
        buf = (C.c_uint * n)()
        assert L.frido_ig_prof_read(buf, n) == 0
        v = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 8, 8).astype(np.float64)[:, :nw, :]
        nk = v[:, :, 6].mean() + 1          # barriers counted = k-tiles - 1
        tm, tn = bm // (nw // 2) // 16, bn // 2 // 16
        own = 3 * tm * tn * 16
        print(f"== {mode} M={M} N={N} K={K} tile {tile} ({bm}x{bn}, {nw} waves, {nwg} workgroups, {nk:.0f} k-tiles): {t * 1e3:.1f} us, {flops / t / 1e9:.1f} TF/s (median of {reps})")
        loop = v[:, :, 3].mean()
        for i, name in ((0, "wait for the next stage's DMA"), (1, "in the barrier"), (2, "virtual steps (MFMAs, reads, DMA issue)")):
            print(f"     {name:40s} {v[:, :, i].mean():10.0f}  [{v[:, :, i].mean() / nk:7.0f} per k-tile]  {v[:, :, i].mean() / loop * 100:5.1f} % of the loop")
        print(f"     loop {loop:.0f}  prologue {v[:, :, 4].mean():.0f}  epilogue {v[:, :, 5].mean():.0f} cycles per wave;  a wave's own MFMAs: {own} cycles per k-tile"
              f" ({own * nk / loop * 100:.0f} % of its loop), x {nw // 4} waves per SIMD" + ("" if nw == 8 else " (x 2 workgroups per CU where they fit)"))
        if nw == 8:
            print("     waves 0-3 / 4-7 per k-tile: " + " | ".join(f"{name}: {v[:, :4, i].mean() / nk:.0f} / {v[:, 4:, i].mean() / nk:.0f}" for i, name in ((0, "DMA wait"), (1, "barrier"), (2, "steps"))))


if __name__ == "__main__":
    main()
