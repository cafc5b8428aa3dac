#!/usr/bin/env python3
"""Per-op device-time breakdown of one denoiser forward (and optionally the decode) using the native
executor's per-op HIP events.  python tools/profile_forward.py [--batch 16] [--precision bf16] [--decode]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import _lib, configs, synth  # noqa: E402
from bench import build_model  # noqa: E402


TOP = 25


def table(prog, sp, title, top=None):
    top = top or TOP
    prog.run(sp)
    ms = prog.run_timed(sp)
    ms2 = prog.run_timed(sp)
    ms = [min(a, b) for a, b in zip(ms, ms2)]
    names = {v: k[len("FRIDO_OP_"):] for k, v in _lib.OP_KINDS.items()}
    rows = []
    bykind = collections.defaultdict(lambda: [0.0, 0])
    for (kind, st), t in zip(prog.ops, ms):
        n = names[kind]
        desc, fl = "", 0.0
        if n == "GEMM":
            fl = 2.0 * st.M * st.N * (st.K + st.K2) * st.batch
            desc = f"M={st.M} N={st.N} K={st.K}+{st.K2} b={st.batch} {'conv%dx%d s%d u%d d%d' % (st.kh, st.kw, st.stride, st.up_shift, st.dn_shift) if st.conv else 'dense'} t{st.tile}" + (" gn" + ("+spade" if st.gn_gamma else "") if st.gn_x1 else "") + (f" sk{st.splitk}" + ("->gn" if st.sk_mode == 2 else "") if st.splitk > 1 else "")
            n = "GEMM-conv" if st.conv else "GEMM"
        if n in ("GN_APPLY", "GN_STATS", "GN_FUSED"):
            desc = f"B={st.B} HW={st.HW} C={st.C1 + st.C2} S={st.nsplit_px}" + (" spade" if n in ("GN_APPLY", "GN_FUSED") and st.gamma else "") + (f" <-sk{st.sk_n}" if n == "GN_FUSED" and st.sk_ws else "")
        if n == "LAYERNORM":
            desc = f"rows={st.rows} C={st.C}"
        if n in ("ATTN_SMALL", "ATTN_FLASH"):
            desc = f"B={st.B} Nq={st.Nq} Nk={st.Nk} d={st.d}"
            fl = 4.0 * st.B * st.Nq * st.Nk * st.d
        if n == "SOFTMAX":
            desc = f"rows={st.rows} N={st.N}"
        rows.append((t, n, desc, fl))
        bykind[n][0] += t
        bykind[n][1] += 1
    tot = sum(ms)
    print(f"== {title}: {len(ms)} ops, {tot:.3f} ms (sum of per-op event intervals)")
    if os.environ.get("ROUNDS"):
        # GEMM time by the number of workgroup ROUNDS a launch needs (workgroups / resident slots): what a persistent, time-shifted
        # tile scheduler could overlap only exists in launches that run more than one round
        dims = {1: (128, 128, 2), 2: (128, 192, 2), 3: (64, 64, 3), 4: (128, 64, 3), 5: (64, 192, 2), 6: (64, 128, 3), 7: (256, 128, 1), 18: (128, 192, 1),
                0: (64, 64, 3)}
        buckets = collections.defaultdict(float)
        gt = 0.0
        for (kind, st), t in zip(prog.ops, ms):
            if names[kind] != "GEMM":
                continue
            bm, bn, per_cu = dims.get(st.tile % 10 if st.tile not in dims else st.tile, (128, 128, 2))
            wgs = -(-st.M // bm) * -(-st.N // bn) * max(st.splitk, 1) * st.batch
            r = wgs / (256.0 * per_cu)
            buckets["<= 1 round" if r <= 1.0 else ("1 - 2 rounds" if r <= 2.0 else "> 2 rounds")] += t
            gt += t
        print("  GEMM-family time by workgroup rounds (workgroups / (256 CUs x resident workgroups per CU)): "
              + ", ".join(f"{k}: {100 * v / gt:.0f} %" for k, v in sorted(buckets.items())))
    for k, (t, c) in sorted(bykind.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:12s} {c:4d} launches {t:8.3f} ms {100 * t / tot:5.1f}%")
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for t, n, desc, fl in rows:
        a = agg[(n, desc)]
        a[0] += t; a[1] += 1; a[2] += fl
    print("  top shapes:")
    for (n, desc), (t, c, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        tf = fl / (t * 1e-3) / 1e12 if fl else 0
        print(f"   {t:8.3f} ms x{c:3d} {n:10s} {desc:60s} {tf:7.1f} TF/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--decode", action="store_true")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    global TOP
    TOP = a.top
    dev = torch.device("cuda:0")
    model = build_model(a.precision, dev)
    from frido_amd.samplers import DDIMSampler
    ctx = torch.from_numpy(synth.seeded_normal("bench:ctx", (a.batch, 26, 640))).to(dev)
    z, _ = DDIMSampler(model).sample(S=2, batch_size=a.batch, shape=(6, 64, 64), conditioning=ctx, num_stage=2, eta=1.0,
                                     verbose=False, noise="philox")
    rt = model.model.diffusion_model.runtime()
    eng = next(iter(rt._sampler_engines.values()))
    sp = torch.cuda.current_stream().cuda_stream
    eng.step.zero_()   # the step counter indexes the timestep table: rewind it after the sampling run
    table(eng.stages[0].step, sp, "stage-0 forward")
    table(eng.stages[1].step, sp, "stage-1 forward")
    table(eng.stages[1].pre, sp, "stage-1 pre (hoisted)")
    if a.decode:
        model.decode_first_stage(z)
        drt = model.first_stage_model.runtime()
        plan = next(iter(drt.plans.values()))[1]
        table(plan.prog, sp, "VQGAN decode")


if __name__ == "__main__":
    main()
