#!/usr/bin/env python3
"""BASELINE config 5 on one MI355X: layout2img 512x512 with a 3-scale feature pyramid (frido_amd/configs.py UNET_512 / VQ_512:
f8f4's denoiser on a 9 x 128 x 128 latent in three stages, self-attention over 4096 tokens, decoder attention over 16384 keys),
per-GPU batch 8 (the config's 64 images over 8 GPUs), DDIM-200 x 3 stages + decode.  Synthetic weights / context.
    python tools/bench_config5.py [--batch 8] [--ddim-steps 200] [--steps 1]
Prints one JSON line.  Not the headline metric: the full three-stage sampling loop at this size, measured once for the record
(parity of the forward and the decode at these dimensions: test_config5_three_scale_512_forward_and_decode)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frido_amd import configs, synth  # noqa: E402
from frido_amd.engine import require_gpu  # noqa: E402


def _args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16x3", choices=["bf16", "bf16x3"])      # bf16x3 = the arithmetic of the parity tests
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ddim-steps", type=int, default=200)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    return ap.parse_args(argv)


def run(**kw):
    """The measurement as a function (bench.py calls it for its `extra` block): keyword overrides of the CLI defaults -> result dict."""
    args = _args([])
    for k, v in kw.items():
        assert hasattr(args, k), k
        setattr(args, k, v)
    dev = require_gpu(f"cuda:{torch.cuda.current_device()}")
    from frido_amd.models import instantiate_from_config
    from frido_amd.pipeline import sample_images
    cfg = configs.frido_cfg(configs.UNET_512, configs.VQ_512, configs.BERT_FULL)
    cfg["cond_stage_config"] = "__is_unconditional__"
    cfg["conditioning_key"] = "crossattn"
    cfg["use_ema"] = False
    cfg["unet_config"]["params"]["precision"] = args.precision
    cfg["first_stage_config"]["params"]["precision"] = args.precision
    model = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg))
    synth.fill_module(model.model, "model.")
    synth.fill_module(model.first_stage_model, "first_stage_model.")
    model = model.to(dev).eval()
    B = args.batch
    ctx = torch.from_numpy(synth.seeded_normal("c5:ctx", (B, 26, configs.UNET_512["context_dim"]))).to(dev)

    def one(k):
        return sample_images(model, ctx, S=args.ddim_steps, eta=1.0, seed=500 + k, sample0=0, noise="philox", total=B)

    for k in range(args.warmup):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        img = one(args.warmup + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert tuple(img.shape) == (B, 3, 512, 512) and bool(torch.isfinite(img).all())
    unet = model.model.diffusion_model
    return {"metric": f"images/sec @ DDIM-{args.ddim_steps}, layout2img 512x512, 3-scale pyramid (BASELINE config 5)",
                      "value": round(B * args.steps / dt, 4), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "dtype": args.precision,
                      "data": "synthetic (random-init weights, N(0,1) context, Philox x_T / noise)",
                      "config": {"workload": f"9 x 128 x 128 latent, {unet.num_stage} stages x DDIM-{args.ddim_steps} eta=1.0, per-GPU batch {B}, "
                                             "512 x 512 MS-VQGAN decode (attention over 16384 keys)",
                                 "denoiser_forwards_per_step": unet.num_stage * args.ddim_steps}}


def main():
    print(json.dumps(run(**vars(_args()))))


if __name__ == "__main__":
    main()
