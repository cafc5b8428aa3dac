#!/usr/bin/env python3
"""shard [lo, hi) of the two-rank test's workload in THIS process (no second process): prints a checksum of latents and images.
python tools/debug_tworank.py lo hi [repeat]"""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FRIDO_TUNE_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tune_cache.json"))
os.environ.setdefault("FRIDO_TUNE_CACHE_READONLY", "1")
os.environ.setdefault("FRIDO_TUNE_ON_MISS", "static")
from bench import build_model
from frido_amd import synth
from frido_amd.samplers import DDIMSampler
lo, hi = int(sys.argv[1]), int(sys.argv[2])
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
model = build_model("bf16x3", dev)
ctx = torch.from_numpy(synth.seeded_normal("two:ctx", (4, 26, 640))[lo:hi]).to(dev)
unet = model.model.diffusion_model
for r in range(rep):
    z, _ = DDIMSampler(model).sample(S=4, batch_size=hi - lo, shape=(unet.in_channels, unet.image_size, unet.image_size), conditioning=ctx,
                                     num_stage=unet.num_stage, eta=1.0, verbose=False, noise="philox", seed=77, sample0=lo, log_every_t=10 ** 9)
    img = model.decode_first_stage(z, to_uint8="np")
    torch.cuda.synchronize()
    print(f"shard [{lo},{hi}) rep {r} pid {os.getpid()}: z sha {hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:12]} img sha {hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
