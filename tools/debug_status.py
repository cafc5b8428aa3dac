#!/usr/bin/env python3
"""Which op raises the library's sticky status word?  Runs the programs of the benchmark model op by op (B from argv, default 4) and prints
every op after which frido_status_flags() is non-zero.   python tools/debug_status.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from frido_amd import _lib, synth  # noqa: E402
from bench import build_model  # noqa: E402


def walk(prog, name, sp):
    names = {v: k[len("FRIDO_OP_"):] for k, v in _lib.OP_KINDS.items()}
    bad = 0
    for i, (kind, st) in enumerate(prog.ops):
        arr = _lib.pack_ops([(kind, st)])
        _lib.check(_lib.lib().frido_run(C.addressof(arr), 1, sp), "run")
        f = _lib.status_flags(clear=True)
        if f:
            bad += 1
            desc = ""
            if names[kind] == "GEMM":
                desc = f"M={st.M} N={st.N} K={st.K}+{st.K2} conv={st.conv} tile={st.tile} geglu={st.geglu} out_op={bool(st.out_op)} out_f32={bool(st.out_f32)} gn={bool(st.gn_x1)} raw={bool(st.raw_x1)} sk={st.splitk}"
            elif names[kind] in ("ATTN_SMALL", "ATTN_FLASH"):
                desc = f"B={st.B} Nq={st.Nq} Nk={st.Nk} d={st.d} out_op={bool(st.out_op)} ln={bool(st.ln_op)} skip={st.skip_act_store}"
            elif names[kind] in ("GN_FUSED", "GN_APPLY", "GN_STATS"):
                desc = f"B={st.B} HW={st.HW} C={st.C1}+{st.C2}"
            elif names[kind] == "LAYERNORM":
                desc = f"rows={st.rows} C={st.C}"
            elif names[kind] == "PACK":
                desc = f"B={st.B} HW={st.HW} Csrc={st.Csrc} Cuse={st.Cuse} nchw={st.nchw}"
            print(f"  {name} op {i} {names[kind]} flags={f} {desc}")
            if names[kind] == "GEMM" and st.gn_x1:      # fused GroupNorm + conv: recompute the staged operand on the host and count what leaves fp16's range
                import torch.nn.functional as F
                from verify_deferred import f32
                HW = st.Hs * st.Ws
                Bi, C1, C2 = st.M // HW, st.gn_C1, st.gn_C2
                x = f32(st.gn_x1, (st.M, C1), sp)
                if C2:
                    x = torch.cat([x, f32(st.gn_x2, (st.M, C2), sp)], 1)
                Cc = C1 + C2
                y = F.group_norm(x.view(Bi, HW, Cc).permute(0, 2, 1), st.gn_groups, f32(st.gn_weight, (Cc,), sp), f32(st.gn_bias, (Cc,), sp), st.gn_eps)
                y = y.permute(0, 2, 1).reshape(st.M, Cc)
                if st.gn_gamma:
                    g, be = f32(st.gn_gamma, (st.M, Cc), sp), f32(st.gn_beta, (st.M, Cc), sp)
                    print(f"      SPADE maps: max |gamma| {float(g.abs().max()):.3g}, max |beta| {float(be.abs().max()):.3g}")
                    y = y * (1 + g) + be
                if st.gn_act == 2:
                    y = F.silu(y)
                over = (y.abs() > 65504.0)
                per = over.view(Bi, -1).sum(1).tolist()
                print(f"      input max |x| {float(x.abs().max()):.3g}; staged operand max |y| {float(y.abs().max()):.4g}; elements beyond 65504: {int(over.sum())} of {y.numel()} (per sample: {per})")
    print(f"{name}: {len(prog.ops)} ops, {bad} raised the flag")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    model = build_model("bf16x3", dev)
    from frido_amd.pipeline import sample_images
    ctx = torch.from_numpy(synth.seeded_normal("bench:ctx", (B, 26, 640))).to(dev)
    _lib.status_flags(clear=True)
    img = sample_images(model, ctx, S=int(sys.argv[2]) if len(sys.argv) > 2 else 4, eta=1.0, seed=1001, sample0=0, noise="philox", total=B, gather_dtype="uint8", check_status=False)
    torch.cuda.synchronize()
    print("after a DDIM-4 pass: flags", _lib.status_flags(clear=True))
    sp = torch.cuda.current_stream().cuda_stream
    rt = model.model.diffusion_model.runtime()
    eng = next(iter(rt._sampler_engines.values()))
    for si, stg in enumerate(eng.stages):
        eng.step.zero_()
        walk(stg.pre, f"stage{si}.pre", sp)
        walk(stg.step, f"stage{si}.step", sp)
    drt = model.first_stage_model.runtime()
    for key, (zs, plan) in drt.plans.items():
        walk(plan.prog, f"decode{key[:3]}", sp)


if __name__ == "__main__":
    main()
