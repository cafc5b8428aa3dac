#!/usr/bin/env python3
"""Experiment: does splitting the per-GPU batch into independent sub-batches that run CONCURRENTLY on their own streams
(own graphs, own scratch) beat one batch-16 chain?  Every image is independent, and a kernel's prologue / epilogue
(~20 us per launch) could overlap the other chain's MFMA work when two workgroups share a CU.
    python tools/dual_stream_exp.py [--batch 16] [--parts 2] [--ddim-steps 200]"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import synth  # noqa: E402
from bench import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--ddim-steps", type=int, default=200)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = build_model(a.precision, dev)
    from frido_amd.samplers import DDIMSampler
    ctx = torch.from_numpy(synth.seeded_normal("bench:ctx", (a.batch, 26, 640))).to(dev)
    kw = dict(S=a.ddim_steps, shape=(6, 64, 64), num_stage=2, eta=1.0, verbose=False, noise="philox", seed=7, log_every_t=10 ** 9)

    def whole():
        z, _ = DDIMSampler(model).sample(batch_size=a.batch, conditioning=ctx, sample0=0, **kw)
        return z

    sub = a.batch // a.parts
    outs = [None] * a.parts

    def part(i):
        torch.cuda.set_device(dev)
        z, _ = DDIMSampler(model).sample(batch_size=sub, conditioning=ctx[i * sub:(i + 1) * sub].contiguous(), sample0=i * sub,
                                         replica=i, **kw)
        outs[i] = z

    def split_threads():
        th = [threading.Thread(target=part, args=(i,)) for i in range(a.parts)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return torch.cat(outs, dim=0)

    def split_serial():
        for i in range(a.parts):
            part(i)
        return torch.cat(outs, dim=0)

    for name, fn in (("whole", whole), ("split-serial", split_serial), ("split-threads", split_threads)):
        fn()                                   # build + tune + capture
        torch.cuda.synchronize()
    ref = None
    for rnd in range(2):
        for name, fn in (("whole", whole), ("split-serial", split_serial), ("split-threads", split_threads)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if ref is None:
                ref = z
            same = float((z - ref).abs().max())
            rel = float((z - ref).norm() / ref.norm())
            per = [(float((z[i] - ref[i]).norm() / ref[i].norm())) for i in range(z.shape[0])]
            print(f"{name:14s} {dt * 1e3:8.1f} ms for {a.batch} latents ({a.batch / dt:.3f} /s, loop only)   max|z - whole| = {same:.3e}  rel {rel:.2e}  "
                  f"worst sample {max(per):.2e}  |z|max {float(ref.abs().max()):.2f}", flush=True)


if __name__ == "__main__":
    main()
