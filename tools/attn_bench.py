#!/usr/bin/env python3
"""Time the attention launches of the bf16x3 sampler on their real shapes (B = 16): self-attention of the 32x32 plane (flash kernel,
1024 keys, d = 384) and the 77-key cross-attentions (short-key kernel).   python tools/attn_bench.py      FRIDO_LIB selects the library."""
import sys

import torch

sys.path.insert(0, ".")
from frido_amd.builder import Builder  # noqa: E402

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


KEEP = []


def rand_op(b, rows, K):
    x = torch.randn(rows, K, device=DEV)
    KEEP.append(x)
    return b.pack(x.data_ptr(), 1, rows, K, 0, K)


def attn(Nq, Nk, d, self_attn, fused_ln=False):
    """fused_ln: the launch as the sampler issues its cross-attention -- the LayerNorm of the result and an operand copy from the same kernel"""
    b = Builder(DEV, 2, {"ln.weight": torch.ones(d, device=DEV), "ln.bias": torch.zeros(d, device=DEV)})
    q = rand_op(b, B * Nq, d)
    k = q if self_attn else rand_op(b, B * Nk, d)
    Np = (Nk + 31) // 32 * 32
    vt_src = torch.randn(B * d, Np, device=DEV)
    if Np != Nk:
        vt_src[:, Nk:] = 0
    KEEP.append(vt_src)
    vT = b.pack(vt_src.data_ptr(), 1, B * d, Np, 0, Np)
    res = b.f32(B * Nq, d)
    res.view().normal_()
    b.prog.run(torch.cuda.current_stream().cuda_stream)      # fill the operands, then time the attention alone
    torch.cuda.synchronize()
    b.prog.ops.clear()
    b.prog._packed = None
    h = b.attention(q, d, k, d, vT, B, Nq, Nk, d, residual=res, stream=True, **(dict(also_op=True, ln=("ln", 1e-5)) if fused_ln else {}))
    if fused_ln and getattr(h, "ln_copy", None) is None:
        b.layernorm(h, "ln")                    # what the plan does where the kernel does not produce the LayerNorm itself
    kinds = [op[0] for op in b.prog.ops]
    us = timed(b)
    fl = 4.0 * B * Nq * Nk * d
    print(f"attention Nq={Nq} Nk={Nk} d={d}{' +ln +op' if fused_ln else ''}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s algorithmic  ops {kinds}")


if __name__ == "__main__":
    attn(1024, 1024, 384, True)
    attn(1024, 77, 384, False)
    attn(256, 77, 576, False)
    attn(64, 77, 960, False)
    attn(256, 256, 576, True)
    attn(64, 64, 960, True)
    for shp in ((1024, 26, 384), (256, 26, 576), (64, 26, 960)):      # config 2: 26 layout tokens
        attn(*shp, False)
        attn(*shp, False, fused_ln=True)
    attn(64, 64, 960, True, fused_ln=True)
