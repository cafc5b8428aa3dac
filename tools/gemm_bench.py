#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM kernel on one shape, per tile variant.
   python tools/gemm_bench.py conv B H W Cin Cout [nsplit] [tiles]   |   python tools/gemm_bench.py dense M N K [nsplit] [tiles]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import _lib  # noqa: E402
from frido_amd.builder import Builder  # noqa: E402
from frido_amd.engine import require_gpu  # noqa: E402

NAMES = {1: "128x128", 2: "128x192", 3: "64x64", 4: "128x64", 5: "64x192", 6: "64x128"}
NAMES.update({k + 10: v + "k64" for k, v in list(NAMES.items())})
NAMES.update({7: "256x128", 8: "256x256", 9: "patch256x192", 10: "patch128x192", 17: "256x128k64", 18: "128x192w8", 19: "256x192"})
NAMES.update({31: "128x128kg2", 33: "64x64kg2", 34: "128x64kg2", 35: "64x192kg2", 36: "64x128kg2"})      # r06: K split inside the workgroup


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
        ns = int(rest[0]) if rest else 1
        b = Builder(dev, ns, {"c.weight": torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05, "c.bias": torch.zeros(Cout, device=dev)})
        x = torch.randn(B, H * W, Cin, device=dev)
        xo = b.pack(x.data_ptr(), 1, B * H * W, Cin, 0, Cin)
        b.conv(xo, B, H, W, "c")
        flops = 2.0 * B * H * W * Cout * 9 * Cin
    elif mode == "geglu":      # python tools/gemm_bench.py geglu M H K  (fused a * gelu(gate) projection, attention.py:37-44)
        M, H, K = map(int, a[1:4])
        rest = a[4:]
        ns = int(rest[0]) if rest else 1
        b = Builder(dev, ns, {"w.weight": torch.randn(2 * H, K, device=dev) * 0.05, "w.bias": torch.zeros(2 * H, device=dev)})
        x = torch.randn(M, K, device=dev)
        xo = b.pack(x.data_ptr(), 1, M, K, 0, K)
        b.linear_geglu(xo, "w")
        flops = 2.0 * M * 2 * H * K
    else:
        M, N, K = map(int, a[1:4])
        rest = a[4:]
        ns = int(rest[0]) if rest else 1
        b = Builder(dev, ns, {"w.weight": torch.randn(N, K, device=dev) * 0.05, "w.bias": torch.zeros(N, device=dev)})
        x = torch.randn(M, K, device=dev)
        xo = b.pack(x.data_ptr(), 1, M, K, 0, K)
        b.linear(xo, "w")
        flops = 2.0 * M * N * K
    tiles = [int(t) for t in rest[1].split(",")] if len(rest) > 1 else [1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 16]
    sp = torch.cuda.current_stream().cuda_stream
    b.prog.run(sp)
    kind, st = b.prog.ops[-1]
    reps = 20
    noepi = os.environ.get("NOEPI") == "1"
    if noepi:
        st.act = 99
    if os.environ.get("NOEPI", "")[:1] in "23456789" and os.environ.get("NOEPI"):
        st.act = 100 - int(os.environ["NOEPI"])
    sk = int(os.environ.get("SPLITK", "1"))
    if sk > 1:
        st.splitk = sk
        st.sk_mode = int(os.environ.get("SKMODE", "0"))      # 1: in-kernel reduction by the last workgroup of each tile
        if not st.sk_mode:
            st.gn_part = None      # (the two-kernel reduction does not produce the fused GroupNorm partial sums)
        st.ws = tune.workspace_for(st, dev)
    for tile in tiles:
        st.tile = tile
        arr = _lib.pack_ops([(kind, st)] * reps)
        ms = (C.c_float * reps)()
        _lib.check(_lib.lib().frido_run_timed(C.addressof(arr), reps, sp, ms), "run")
        t = sorted(ms)[reps // 2]
        print(f"tile {NAMES[tile]:12s} {t * 1e3:8.1f} us  {flops / t / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    main()
