#!/usr/bin/env python3
"""Throughput of BASELINE config 3 on one MI355X: COCO-2014 text2img 256x256 (configs/frido/t2i/frido_f16f8_coco_clip.yaml:21-77
restated in frido_amd/configs.py: 8 x 32 x 32 latent, two stages, ONE 768-d context token), batch 32, PLMS-100 with
classifier-free guidance 1.5 (scripts/sample_diffusion.py's t2i call), MS-VQGAN f16f8 decode.  Synthetic weights and context
(the CLIP text encoder's weights are unreachable: conditioning tensors are fed directly, as in the parity tests).
    python tools/bench_config3.py [--precision bf16|bf16x3] [--batch 32] [--steps 2]
Prints one JSON line.  Not the headline metric (bench.py is): a second measured workload for the record."""
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--plms-steps", type=int, default=100)
    ap.add_argument("--steps", type=int, default=2)
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
    from frido_amd.samplers import PLMSSampler
    cfg = configs.frido_cfg(configs.UNET_F16F8, configs.VQ_F16F8, configs.BERT_FULL)
    cfg["cond_stage_config"] = "__is_unconditional__"
    cfg["conditioning_key"] = "crossattn"
    cfg["use_ema"] = False
    cfg["unet_config"]["params"]["precision"] = args.precision
    cfg["first_stage_config"]["params"]["precision"] = args.precision
    model = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg))
    synth.fill_module(model.model, "model.")
    synth.fill_module(model.first_stage_model, "first_stage_model.")
    model = model.to(dev).eval()
    unet = model.model.diffusion_model
    B = args.batch
    nctx, dctx = 1, configs.UNET_F16F8["context_dim"]
    c = torch.from_numpy(synth.seeded_normal("c3:ctx", (B, nctx, dctx))).to(dev)
    uc = torch.from_numpy(synth.seeded_normal("c3:uc", (B, nctx, dctx))).to(dev)
    shape = (unet.in_channels, unet.image_size, unet.image_size)

    def one(k):
        z, _ = PLMSSampler(model).sample(S=args.plms_steps, batch_size=B, shape=shape, conditioning=c, num_stage=unet.num_stage,
                                         eta=0.0, verbose=False, unconditional_guidance_scale=1.5, unconditional_conditioning=uc,
                                         noise="philox", seed=100 + k, log_every_t=10 ** 9)
        return model.decode_first_stage(z)

    for k in range(args.warmup):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        img = one(args.warmup + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert img.shape[0] == B and bool(torch.isfinite(img).all())
    return {"metric": f"images/sec @ PLMS-{args.plms_steps} + CFG 1.5, COCO text2img 256x256 (BASELINE config 3)",
                      "value": round(B * args.steps / dt, 4), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "dtype": args.precision,
                      "data": "synthetic (random-init weights, N(0,1) context tokens, Philox x_T)",
                      "config": {"workload": f"t2i f16f8, batch {B}, PLMS-{args.plms_steps} x {unet.num_stage} stages, CFG 1.5 "
                                             f"(cond + uncond batched: {2 * B} rows per forward), image {tuple(img.shape[1:])}"}}


def main():
    print(json.dumps(run(**vars(_args()))))


if __name__ == "__main__":
    main()
