#!/usr/bin/env python3
"""WHERE does the benchmark workload leave fp16's range?  Pass 1 runs the sampling loop with a per-step callback that reads (and clears)
the library's sticky status word; pass 2 re-runs the same seed, stops right before the first flagged step and walks that step's program
op by op (tools/debug_status.py), recomputing the flagged operand on the host.   python tools/find_saturation.py [B] [S]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from frido_amd import _lib, synth  # noqa: E402
from bench import build_model  # noqa: E402
from debug_status import walk  # noqa: E402


class Stop(Exception):
    pass


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    model = build_model("bf16x3", dev)
    from frido_amd.samplers import DDIMSampler
    unet = model.model.diffusion_model
    ctx = torch.from_numpy(synth.seeded_normal("bench:ctx", (B, 26, 640))).to(dev)
    kw = dict(S=S, batch_size=B, shape=(unet.in_channels, unet.image_size, unet.image_size), conditioning=ctx, num_stage=unet.num_stage,
              eta=1.0, verbose=False, noise="philox", seed=1001, sample0=0, log_every_t=10 ** 9)
    calls, flagged = [0], []

    def cb1(i):
        torch.cuda.synchronize()
        import ctypes as C
        per = []
        for k in range(8):
            w = C.c_uint32(0)
            if _lib.lib().frido_status_word_of(k, C.byref(w)) != 0:
                break
            per.append(int(w.value))
        f = _lib.status_flags(clear=True)
        if f:
            flagged.append((calls[0], calls[0] // S, i, f, per))
        calls[0] += 1

    import hashlib
    if os.environ.get("FIND_SAT_REPEAT"):
        for rep in range(int(os.environ["FIND_SAT_REPEAT"])):      # run-to-run identity of the result itself (no callback: pure graph replay)
            _lib.status_flags(clear=True)
            zz, _ = DDIMSampler(model).sample(**kw)
            torch.cuda.synchronize()
            print(f"repeat {rep}: z sha {hashlib.sha256(zz.cpu().numpy().tobytes()).hexdigest()[:16]} flags {_lib.status_flags(clear=True)}", flush=True)
    _lib.status_flags(clear=True)
    z, _ = DDIMSampler(model).sample(callback=cb1, **kw)
    torch.cuda.synchronize()
    print(f"with per-step callback: z sha {hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:16]}")
    print(f"pass 1: {calls[0]} steps, {len(flagged)} raised the status word; first: {flagged[:8]}; latent max |z| {float(z.abs().max()):.3g}")
    by_stage = {}
    for c, s, i, f, per in flagged:
        by_stage.setdefault(s, []).append(i)
    for s, lst in by_stage.items():
        print(f"  stage {s}: flagged steps {lst[:20]}{' ...' if len(lst) > 20 else ''} ({len(lst)} of {S})")
    if not flagged or flagged[0][0] == 0:
        return
    first = flagged[0][0]
    calls[0] = 0

    def cb2(i):
        calls[0] += 1
        if calls[0] == first:
            torch.cuda.synchronize()
            raise Stop()

    try:
        DDIMSampler(model).sample(callback=cb2, **kw)
    except Stop:
        pass
    _lib.status_flags(clear=True)
    sp = torch.cuda.current_stream().cuda_stream
    eng = next(iter(unet.runtime()._sampler_engines.values()))
    s = flagged[0][1]
    print(f"pass 2: stopped before call {first} (stage {s}, step {flagged[0][2]}); device step counter {int(eng.step.item())}; walking stage{s}.step")
    walk(eng.stages[s].step, f"stage{s}.step", sp)


if __name__ == "__main__":
    main()
