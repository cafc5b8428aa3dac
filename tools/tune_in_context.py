#!/usr/bin/env python3
"""IN-CONTEXT tile selection for the denoiser's GEMM launches (a round-6 tool; python only, no library change).

Why: frido_amd/tune.py picks a tile per GEMM signature from a back-to-back microbenchmark on scratch buffers.  r05's stagger experiment
showed that such a microbenchmark does not predict what a launch does INSIDE the forward (profiles/r05_stagger_*: 8 us of start delay is
neutral per launch and + 2 % end to end): in the forward a launch finds its activation operand where the previous kernel left it, its
weights cold, its epilogue in its real form (GroupNorm partial sums, residual, time-embedding vector), and it hands over to a different
kernel.  This tool re-decides the tile of every GEMM signature of the step programs from per-op HIP-event timings of the WHOLE forward
(Prog.run_timed): for one signature at a time every op of that signature is switched to a candidate tile (split-K unchanged), the forward
is timed `--reps` times, and the candidate is scored by the summed median time of those ops AND of the op that follows each of them.
A candidate replaces the pinned tile only when it wins by --min-gain.  Results unchanged by construction (a tile is a schedule, the
arithmetic per output element is the same K order); the GPU suite must still be re-run on the new cache before it is committed.

    python tools/tune_in_context.py [--batch 16] [--reps 5] [--min-gain 0.015] [--out gpurun_out/tune_in_context.json] [--write-cache FILE]
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FOUR_WAVE = (1, 2, 3, 4, 5, 6)
EIGHT_WAVE = (7, 18, 19)          # bf16x3 mode (frido_hip.h FridoGemm.tile)
KG2_DIMS = {31: (128, 128), 33: (64, 64), 34: (128, 64), 35: (64, 192), 36: (64, 128)}      # K split over the two wave groups of one workgroup (r06)
TILE_DIMS = {1: (128, 128), 2: (128, 192), 3: (64, 64), 4: (128, 64), 5: (64, 192), 6: (64, 128)}      # the two-per-CU (4-wave) tiles


def workgroups(st):
    """Grid of a 4-wave-tile launch (None for the other tiles): what the library's stagger threshold is compared with."""
    if st.tile not in TILE_DIMS:
        return None
    bm, bn = TILE_DIMS[st.tile]
    return -(-st.M // bm) * -(-st.N // bn) * max(st.batch, 1) * max(st.splitk, 1)


def set_stagger(st, quarter_us):
    st.flags = (st.flags & ~0xFF00) | ((int(quarter_us) & 255) << 8)


def candidates(st):
    """Tiles worth trying for a descriptor: the filter of tune.best_tile, split-K left as it is."""
    if st.nsplit != 2 or st.tile in (0, 9, 10, 20, 21) or (st.tile > 19 and st.tile not in KG2_DIMS) or st.splitk > 1 or st.up2_phase:
        return []           # fused GroupNorm + conv / patch kernels / split-K launches (workspace layout may depend on the tile) stay
    out = [t for t in FOUR_WAVE if not (t in (1, 2, 4) and st.M < 64)]
    if st.M >= 512 and st.N >= 96:
        out += list(EIGHT_WAVE)
    nk = (st.K + st.K2) // 32
    if not st.conv and nk >= 2 and nk % 2 == 0:      # (r06) K split inside the workgroup: sub-round dense launches only (tune.py's filter)
        out += [t for t, (bm, bn) in KG2_DIMS.items()
                if not (t in (31, 34) and st.M < 64) and -(-st.M // bm) * -(-st.N // bn) * max(st.batch, 1) <= 640]
    return [t for t in out if t != st.tile]


def group_ops(progs, sig_of, kind_gemm):
    """signature -> [(prog index, op index)] over the GEMM ops of the given programs."""
    groups = {}
    for pi, prog in enumerate(progs):
        for oi, (kind, st) in enumerate(prog.ops):
            if kind == kind_gemm:
                groups.setdefault(sig_of(st), []).append((pi, oi))
    return groups


def score(times, members, progs):
    """Summed median time of the member ops and of the op after each (its hand-over), in ms.  times[pi] = list of per-op lists."""
    idx = set()
    for pi, oi in members:
        idx.add((pi, oi))
        if oi + 1 < len(progs[pi].ops):
            idx.add((pi, oi + 1))
    return sum(statistics.median(rep[oi] for rep in times[pi]) for pi, oi in idx)


def tune(progs, time_forward, sig_of, kind_gemm, *, min_gain=0.015, min_share=0.002, stagger=(), min_wg=512, log=print):
    """Coordinate descent over signatures, heaviest first.  time_forward() -> times[pi][rep][oi] in ms; raises on a rejected descriptor.
    stagger: start delays (quarter microseconds) to try on the 4-wave-tile launches of at least min_wg workgroups (needs a
    -DFRIDO_STAGGER_RT=1 library with FRIDO_STAGGER_MIN_WG <= min_wg in the environment).
    Returns [(signature, old tile, new tile, old score, new score, delay)] of the changes that were kept."""
    base = time_forward()
    total = sum(statistics.median(rep[oi] for rep in base[pi]) for pi in range(len(progs)) for oi in range(len(progs[pi].ops)))
    groups = group_ops(progs, sig_of, kind_gemm)
    order = sorted(groups, key=lambda g: -score(base, groups[g], progs))
    kept = []
    for sig in order:
        members = groups[sig]
        st0 = progs[members[0][0]].ops[members[0][1]][1]
        cands = candidates(st0)
        cur = score(base, members, progs)
        if not cands or cur < min_share * total:
            continue
        old = st0.tile
        best_t, best_s = old, cur
        for t in cands:
            for pi, oi in members:
                progs[pi].ops[oi][1].tile = t
                progs[pi]._packed = None
            try:
                s = score(time_forward(), members, progs)
            except Exception as e:      # noqa: BLE001  (the library rejects a tile that does not apply to the descriptor)
                log(f"   tile {t}: rejected ({type(e).__name__})")
                continue
            if s < best_s:
                best_t, best_s = t, s
        keep = best_t != old and best_s <= cur * (1.0 - min_gain)
        for pi, oi in members:
            progs[pi].ops[oi][1].tile = best_t if keep else old
            progs[pi]._packed = None
        log(f"M={st0.M} N={st0.N} K={st0.K}+{st0.K2} b={st0.batch} {'conv' if st0.conv else 'dense'} x{len(members)}: tile {old} "
            f"{cur * 1e3:.1f} us -> best {best_t} {best_s * 1e3:.1f} us  {'KEPT' if keep else 'unchanged'}")
        best_q = 0
        if stagger and st0.nsplit == 2 and st0.tile in TILE_DIMS and (workgroups(st0) or 0) >= min_wg:
            # start delay of the second resident slot (FridoGemm.flags bits 8..15, -DFRIDO_STAGGER_RT=1 builds): the same in-context score
            ref_s = best_s if keep else cur
            q_s = ref_s
            for q in stagger:
                for pi, oi in members:
                    set_stagger(progs[pi].ops[oi][1], q)
                    progs[pi]._packed = None
                s = score(time_forward(), members, progs)
                if s < q_s:
                    best_q, q_s = q, s
            if q_s > ref_s * (1.0 - min_gain):
                best_q = 0
            for pi, oi in members:
                set_stagger(progs[pi].ops[oi][1], best_q)
                progs[pi]._packed = None
            log(f"   stagger: best {best_q / 4:.2f} us  {ref_s * 1e3:.1f} -> {q_s * 1e3:.1f} us  {'KEPT' if best_q else 'none'}")
            if best_q:
                best_s = q_s
        if keep or best_q:
            kept.append((sig, old, best_t if keep else old, cur, best_s, best_q))
            base = time_forward()       # later signatures are judged in the new context
    return kept, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--min-gain", type=float, default=0.015)
    ap.add_argument("--out", default="gpurun_out/tune_in_context.json")
    ap.add_argument("--stagger", default="", help="comma-separated start delays in microseconds to try per signature, e.g. 1,2,4,6,8,12,16 "
                                                  "(FRIDO_LIB = a -DFRIDO_STAGGER_RT=1 build, FRIDO_STAGGER_MIN_WG=<--min-wg> in the environment)")
    ap.add_argument("--min-wg", type=int, default=512)
    ap.add_argument("--write-cache", default="", help="write a tile cache = the loaded one with the kept changes (same format as profiles/tune_cache.json)")
    a = ap.parse_args()
    import torch
    from frido_amd import _lib, synth, tune as T
    from bench import build_model
    from frido_amd.samplers import DDIMSampler
    dev = torch.device("cuda:0")
    model = build_model("bf16x3", dev)
    ctx = torch.from_numpy(synth.seeded_normal("bench:ctx", (a.batch, 26, 640))).to(dev)
    DDIMSampler(model).sample(S=2, batch_size=a.batch, shape=(6, 64, 64), conditioning=ctx, num_stage=2, eta=1.0, verbose=False, noise="philox")
    eng = next(iter(model.model.diffusion_model.runtime()._sampler_engines.values()))
    progs = [stg.step for stg in eng.stages]
    sp = torch.cuda.current_stream().cuda_stream

    def time_forward():
        out = []
        for p in progs:
            eng.step.zero_()
            p.run(sp)                   # (validates the descriptors; warms the caches the way the previous step would have)
            out.append([p.run_timed(sp) for _ in range(a.reps)])
        return out

    stagger = tuple(int(round(float(x) * 4)) for x in a.stagger.split(",") if x.strip())
    kept, total = tune(progs, time_forward, T.signature, _lib.OP_KINDS["FRIDO_OP_GEMM"], min_gain=a.min_gain, stagger=stagger, min_wg=a.min_wg)
    after = time_forward()
    total_after = sum(statistics.median(rep[oi] for rep in after[pi]) for pi in range(len(progs)) for oi in range(len(progs[pi].ops)))
    rec = {"batch": a.batch, "forward_ms_before": round(total, 4), "forward_ms_after": round(total_after, 4),
           "changes": [{"sig": list(s), "tile_old": o, "tile_new": n, "score_us_old": round(c * 1e3, 2), "score_us_new": round(b * 1e3, 2),
                        "stagger_us": q / 4} for s, o, n, c, b, q in kept]}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(f"forward (both stages, summed per-op medians): {total:.3f} -> {total_after:.3f} ms, {len(kept)} signatures changed; wrote {a.out}")
    if a.write_cache:
        entries = dict(T._cache)
        for s, _, n, _, _, q in kept:
            entries[s] = (n, entries.get(s, (0, 1))[1]) + ((q,) if q else ())
        json.dump({"lib": T._lib_tag(), "entries": [[list(k), list(v)] for k, v in entries.items() if not isinstance(k[-1], str)]},
                  open(a.write_cache, "w"))
        print(f"wrote {a.write_cache} ({len(entries)} entries)")


if __name__ == "__main__":
    main()
