#!/usr/bin/env python3
"""Checks every DEFERRED split-K reduction of a sampler's step programs (a GN_FUSED op with sk_ws, builder._deferred_splitk) against torch:
runs the program op by op and, right after such an op, recomputes  x1 = alpha * sum_z ws[z] + bias + rowvec + residual  and the
GroupNorm [+SPADE] [+SiLU] of cat(x1, x2) in fp32 from the device buffers the descriptor names.   python tools/verify_deferred.py [config3|config2] [B]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import _lib, configs, synth  # noqa: E402


def dev_copy(ptr, nbytes, sp):
    n = (nbytes + 15) // 16 * 16
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    kind, st = _lib.make_op("FRIDO_OP_COPY", src=ptr, dst=t.data_ptr(), n=n)
    arr = _lib.pack_ops([(kind, st)])
    _lib.check(_lib.lib().frido_run(C.addressof(arr), 1, sp), "copy")
    torch.cuda.synchronize()
    return t[:nbytes]


def f32(ptr, shape, sp):
    n = 1
    for s in shape:
        n *= s
    return dev_copy(ptr, 4 * n, sp).view(torch.float32).view(*shape).clone()


def check_prog(prog, name, sp, step_val, quiet=False):
    names = {v: k[len("FRIDO_OP_"):] for k, v in _lib.OP_KINDS.items()}
    worst = 0.0
    nchk = 0
    for i, (kind, st) in enumerate(prog.ops):
        res_before = None
        if names[kind] == "GN_FUSED" and st.sk_ws and st.sk_residual:      # the residual may be updated IN PLACE (sk_out == sk_residual): read it first
            torch.cuda.synchronize()
            res_before = f32(st.sk_residual, (st.B * st.HW, st.sk_ldr), sp)
        arr = _lib.pack_ops([(kind, st)])
        _lib.check(_lib.lib().frido_run(C.addressof(arr), 1, sp), "run")
        if names[kind] != "GN_FUSED" or not st.sk_ws:
            continue
        torch.cuda.synchronize()
        B, HW, C1, C2 = st.B, st.HW, st.C1, st.C2
        M, Cc = B * HW, C1 + C2
        ws = f32(st.sk_ws, (st.sk_n, M, C1), sp)
        x1 = torch.zeros(M, C1, device="cuda")
        for z in range(st.sk_n):
            x1 = x1 + ws[z]
        add = torch.zeros(C1, device="cuda")
        if st.sk_bias:
            add = add + f32(st.sk_bias, (C1,), sp)
        x1 = x1 * st.sk_alpha + add
        if st.sk_rowvec:
            vs = step_val if st.sk_rowvec_step else 0
            rv = f32(st.sk_rowvec + 4 * vs * st.sk_ldv, (C1,), sp) if st.sk_rows_per_vec >= (1 << 29) else None
            assert rv is not None, "per-row vectors not handled by this checker"
            x1 = x1 + rv
        if st.sk_residual:
            x1 = x1 + res_before[:, :C1]
        x = x1 if not C2 else torch.cat([x1, f32(st.x2, (M, C2), sp)], dim=1)
        if st.sk_out:
            got_x1 = f32(st.sk_out, (M, C1), sp)
            ex = float((got_x1 - x1).abs().max() / x1.abs().max())
        else:
            ex = 0.0
        w, b = f32(st.weight, (Cc,), sp), f32(st.bias, (Cc,), sp)
        y = F.group_norm(x.view(B, HW, Cc).permute(0, 2, 1), st.groups, w, b, st.eps).permute(0, 2, 1).reshape(M, Cc)
        if st.gamma:
            y = y * (1 + f32(st.gamma, (M, Cc), sp)) + f32(st.beta, (M, Cc), sp)
        if st.act == 2:
            y = F.silu(y)
        dt = torch.float16 if _lib.lib().frido_x3_plane_format() == 1 else torch.bfloat16
        hi = dev_copy(st.out_op, 2 * M * Cc, sp).view(dt).view(M, Cc).float()
        lo = dev_copy(st.out_op + 2 * st.out_lo, 2 * M * Cc, sp).view(dt).view(M, Cc).float()
        err = float(((hi + lo) - y).abs().max() / y.abs().max())
        nchk += 1
        worst = max(worst, err, ex)
        flag = "" if max(err, ex) < 1e-4 else "   <<<<<< MISMATCH"
        if quiet and not flag:
            continue
        print(f"  {name} op {i}: GN_FUSED<-sk{st.sk_n} B={B} HW={HW} C={C1}+{C2} spade={bool(st.gamma)} resid={bool(st.sk_residual)} rowvec={bool(st.sk_rowvec)} inplace={bool(st.sk_out) and st.sk_out == st.sk_residual} "
              f"x1_dead={not st.sk_out}: x1 err {ex:.2e}, out err {err:.2e}{flag}")
    print(f"{name}: {nchk} deferred reductions checked, worst {worst:.2e}")
    return (worst, nchk) if quiet else worst


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "config3"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    from frido_amd.models import instantiate_from_config
    from frido_amd.samplers import PLMSSampler, DDIMSampler
    if which == "config3":
        u, v, nctx, shape, cls, kw = configs.UNET_F16F8, configs.VQ_F16F8, 1, (8, 32, 32), PLMSSampler, dict(eta=0.0, unconditional_guidance_scale=1.5)
    else:
        u, v, nctx, shape, cls, kw = configs.UNET_F8F4, configs.VQ_F8F4, 26, (6, 64, 64), DDIMSampler, dict(eta=1.0)
    cfg = configs.frido_cfg(u, v, configs.BERT_FULL)
    cfg["cond_stage_config"], cfg["conditioning_key"], cfg["use_ema"] = "__is_unconditional__", "crossattn", False
    model = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg))
    synth.fill_module(model.model, "model.")
    synth.fill_module(model.first_stage_model, "first_stage_model.")
    model = model.cuda().eval()
    c = torch.from_numpy(synth.seeded_normal("vd:c", (B, nctx, u["context_dim"]))).cuda()
    if which == "config3":
        kw["unconditional_conditioning"] = torch.from_numpy(synth.seeded_normal("vd:uc", (B, nctx, u["context_dim"]))).cuda()
    z, _ = cls(model).sample(S=4, batch_size=B, shape=shape, conditioning=c, num_stage=2, verbose=False, noise="philox", seed=3, log_every_t=10 ** 9, **kw)
    torch.cuda.synchronize()
    sp = torch.cuda.current_stream().cuda_stream
    rt = model.model.diffusion_model.runtime()
    eng = next(iter(rt._sampler_engines.values()))
    worst = 0.0
    for si, stg in enumerate(eng.stages):
        eng.step.zero_()
        worst = max(worst, check_prog(stg.step, f"stage{si}.step", sp, 0))
    print("WORST", worst)


if __name__ == "__main__":
    main()
