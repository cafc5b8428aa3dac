#!/usr/bin/env python3
"""Error-budget table of the bf16x3 mode: keep ONE bf16 plane of a single operand class inside the parity
arithmetic and measure the end-to-end error against the reference's own CPU run (config 1: layout2i f8f4 full width, B = 1,
DDIM-50, eta = 1, torch noise stream; tests/golden/sampler_full.npz).   python tools/x3_single_plane_table.py
Each variant runs in its own process (the switches are read at import).  -> profiles/r03_x3_single_plane_table.txt"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("bf16x3 everywhere (shipped)", {}, "bf16x3", "bf16x3"),
            ("MS-VQGAN decoder in bf16 (one plane)", {}, "bf16x3", "bf16"),
            ("SPADE gamma / beta maps stored as bf16", {"FRIDO_X3_SPADE_BF16": "1"}, "bf16x3", "bf16x3"),
            ("cached cross-attention K / V^T: hi plane only", {"FRIDO_X3_CROSSKV_HI": "1"}, "bf16x3", "bf16x3"),
            ("bf16 everywhere (throughput mode)", {}, "bf16", "bf16")]

CHILD = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden")); sys.path.insert(0, REPO)
import test_model_gpu as T
from golden_cfg import UNET_FULL, VQ_FULL, BERT_SMALL, frido_cfg
from frido_amd.models import instantiate_from_config
from frido_amd.synth import fill_module
from frido.models.diffusion.ddim import DDIMSampler
g = T.golden("sampler_full")
cfg = frido_cfg(dict(UNET_FULL, precision=UP), dict(VQ_FULL, precision=VP), BERT_SMALL)
cfg["cond_stage_config"], cfg["conditioning_key"] = "__is_unconditional__", "crossattn"
m = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg))
fill_module(m.model, "model."); fill_module(m.first_stage_model, "first_stage_model.")
m.scale_factor.copy_(torch.tensor([0.9, 1.1])); m = m.cuda().eval()
rec = T._Rec(); torch.manual_seed(23)
z, _ = DDIMSampler(m).sample(S=50, batch_size=1, shape=(6, 64, 64), conditioning=torch.from_numpy(g["c"]).cuda(), num_stage=2, eta=1.0,
                             verbose=False, log_every_t=int(g["ddim50_args"][3]), noise=rec)
T._record = lambda *a, **k: None
rep = T._e2e_report("x", m, g, "ddim50", z, [3, 3])
print("RESULT " + json.dumps({k: float(v) for k, v in rep.items()}))
'''


def main():
    rows = []
    for name, env, up, vp in VARIANTS:
        code = f"REPO = {REPO!r}\nUP = {up!r}\nVP = {vp!r}\n" + CHILD
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), cwd=REPO)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(name, "FAILED", out.stderr[-800:])
            continue
        rows.append((name, json.loads(line[0][7:])))
    print(f"{'variant':52s} {'latent rel':>10s} {'VQ flips':>9s} {'pix p50':>9s} {'pix p99':>9s} {'pix max':>9s} {'<= 1e-3':>8s}")
    for name, r in rows:
        ok = r["pix_max"] <= 1e-3
        print(f"{name:52s} {r['latent_rel']:10.2e} {r['vq_flip_rate'] * 8192:9.0f} {r['pix_p50']:9.2e} {r['pix_p99']:9.2e} {r['pix_max']:9.2e} {'yes' if ok else 'NO':>8s}")


if __name__ == "__main__":
    main()
