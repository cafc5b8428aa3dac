#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING THE REFERENCE on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes are data
(inputs + expected outputs) and are committed.  Weights are never stored: both sides regenerate
them with frido_amd.synth.fill_tensor keyed by state_dict name.

    python tests/golden/make_golden.py [names...]

Reference entry points exercised (file:line in /root/reference):
  frido/modules/diffusionmodules/util.py:21-74,151-171   schedules, timestep_embedding
  frido/modules/diffusionmodules/pyunet.py:867-950       PyUNetModel.forward
  taming/models/msvqgan.py:326-399                       VQModelInterface.encode / decode
  taming/modules/vqvae/quantize.py:267-308               VectorQuantizer2.forward
  frido/models/diffusion/ddim.py:56-273, plms.py:57-303  DDIM / PLMS sampling loops
  frido/models/diffusion/frido.py:823-891                decode_first_stage
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from frido_amd.synth import fill_tensor, seeded_normal  # noqa: E402
sys.path.remove(REPO)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("_ref_harness", os.path.join(REPO, "oracle", "_ref_harness.py"))
H = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(H)

sys.path.insert(0, HERE)
from golden_cfg import (UNET_SMALL, UNET_SMALL_D2, UNET_FULL, VQ_SMALL, VQ_FULL, UNET_SMALL3, VQ_SMALL3, BERT_SMALL,  # noqa: E402
                        frido_cfg, BERT_FULL, UNET_F16F8, VQ_F16F8, UNET_512, VQ_512)


PROFILE = None       # --profile heavy: the trained-checkpoint-like filler (frido_amd/synth.py _fill_heavy); fixtures get the suffix _heavy


def fill_module(mod, prefix=""):
    """Fill every parameter/buffer-free weight of `mod` from the deterministic filler."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            p.copy_(torch.from_numpy(fill_tensor(prefix + name, p.shape, PROFILE)))
    return mod


def save(name, **arrs):
    if PROFILE:
        name = f"{name}_{PROFILE}"
        arrs["filler_profile"] = np.array(PROFILE)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------------------------------------
def gen_schedules():
    util = H.import_ref("frido.modules.diffusionmodules.util")
    betas = util.make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0155)
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
    out = {"betas": betas.astype(np.float32), "alphas_cumprod": alphas_cumprod.astype(np.float32),
           "betas64": betas, "alphas_cumprod64": alphas_cumprod}
    ac32 = torch.tensor(alphas_cumprod, dtype=torch.float32)  # what DDPM.register_schedule stores (frido.py:144-155)
    for S in (4, 50, 100, 200, 250):
        ts = util.make_ddim_timesteps("uniform", S, 1000, verbose=False)
        out[f"ts_{S}"] = ts
        for eta in (0.0, 1.0):
            sig, a, ap = util.make_ddim_sampling_parameters(ac32.cpu(), ts, eta, verbose=False)
            tag = f"{S}_{int(eta)}"
            out[f"sigmas_{tag}"] = np.asarray(sig)
            out[f"alphas_{tag}"] = np.asarray(a)
            out[f"alphas_prev_{tag}"] = np.asarray(ap)
            out[f"sqrt1m_{tag}"] = np.sqrt(1.0 - np.asarray(a))
    t = torch.tensor([1, 6, 251, 501, 996], dtype=torch.long)
    out["temb_t"] = t.numpy()
    out["temb_192"] = util.timestep_embedding(t, 192).numpy()
    out["temb_32"] = util.timestep_embedding(t, 32).numpy()
    save("schedules", **out)


# ---------------------------------------------------------------------------------------------
def build_unet(cfg):
    m = H.import_ref("frido.modules.diffusionmodules.pyunet")
    net = m.PyUNetModel(**cfg)
    fill_module(net, "model.diffusion_model.")
    return net.eval()


def gen_unet(tag, cfg, B, nctx, hw, capture=True):
    net = build_unet(cfg)
    cin = cfg["in_channels"]
    x = T(seeded_normal(f"{tag}:x", (B, cin, hw, hw)))
    ctx = T(seeded_normal(f"{tag}:ctx", (B, nctx, cfg["context_dim"])))
    out = {"x": x.numpy(), "ctx": ctx.numpy()}
    nstage = cfg.get("num_stage", 1)
    splits = cfg["split_embed_dim_list"]
    for s in range(nstage):
        t = torch.tensor([996 - 37 * i for i in range(B)], dtype=torch.long)
        xin = x[:, :sum(splits[:s + 1])].contiguous()
        caps = {}
        hooks = []
        if capture:
            def mk(name):
                def hook(mod, inp, o):
                    caps[name] = o.detach().numpy().copy()
                return hook
            hooks.append(net.input_blocks[0].register_forward_hook(mk("ib0")))
            hooks.append(net.middle_block.register_forward_hook(mk("mid")))
            hooks.append(net.pre_input_blocks[s].register_forward_hook(mk("pre")))
            hooks.append(net.output_blocks[-1].register_forward_hook(mk("ob_last")))
        smax = [0.0]
        for blk in list(net.input_blocks) + [net.middle_block] + list(net.output_blocks):
            hooks.append(blk.register_forward_hook(lambda m, i, o, smax=smax: smax.__setitem__(0, max(smax[0], float(o.detach().abs().max())))))
        with torch.no_grad():
            e = net(xin, t, context=ctx, stage=s)
        for h in hooks:
            h.remove()
        out[f"stream_absmax_{s}"] = np.float64(smax[0])
        print(f"  stage {s}: residual stream max |x| = {smax[0]:.4g}, eps max {float(e.abs().max()):.4g}")
        out[f"t_{s}"] = t.numpy()
        out[f"eps_{s}"] = e.numpy()
        for k, v in caps.items():
            out[f"{k}_{s}"] = v
    nparam = sum(p.numel() for p in net.parameters())
    out["nparam"] = np.int64(nparam)
    out["keys"] = np.array(sorted(k for k, _ in net.named_parameters()))
    save(tag, **out)


# ---------------------------------------------------------------------------------------------
def build_vq(cfg):
    m = H.import_ref("taming.models.msvqgan")
    p = dict(cfg)
    p["lossconfig"] = {"target": "torch.nn.Identity"}
    net = m.VQModelInterface(**p)
    fill_module(net, "first_stage_model.")
    return net.eval()


def gen_vq(tag, cfg, B, subsample=1):
    net = build_vq(cfg)
    res = cfg["ddconfig"]["resolution"]
    f = 2 ** (len(cfg["ddconfig"]["ch_mult"]) - 1)
    hl = res // f
    zc = cfg["ddconfig"]["z_channels"]
    h = T(seeded_normal(f"{tag}:h", (B, zc, hl, hl)) * 1.5)
    with torch.no_grad():
        dec, code = net.decode(h, return_code=True)
        # quantizer alone, incl. an exact-tie case: duplicate two codebook rows
        vq = net.ms_quantize[0]
        zq, _, info = vq(h[:, :cfg["embed_dim"][0]])
        w0 = vq.embedding.weight.data.clone()
        vq.embedding.weight.data[7] = vq.embedding.weight.data[3]
        zt = vq.embedding.weight.data[3].view(1, -1, 1, 1).repeat(1, 1, 2, 2).clone()
        zq_tie, _, info_tie = vq(zt)
        vq.embedding.weight.data.copy_(w0)
    out = {"h": h.numpy(), "dec": dec.numpy()[:, :, ::subsample, ::subsample],
           "dec_sum": np.float64(dec.double().sum().item()),
           "dec_abs_sum": np.float64(dec.double().abs().sum().item()),
           "code": np.asarray(code, dtype=np.int64), "zq0": zq.numpy(),
           "idx0": info[2].numpy(), "idx_tie": info_tie[2].numpy(),
           "subsample": np.int64(subsample)}
    if subsample == 1:  # small model: also pin encode
        x = T(np.tanh(seeded_normal(f"{tag}:img", (B, 3, res, res))))
        with torch.no_grad():
            enc = net.encode(x)
        out["img"] = x.numpy()
        out["enc"] = enc.numpy()
    out["keys"] = np.array(sorted(k for k, _ in net.named_parameters()))
    save(tag, **out)


# ---------------------------------------------------------------------------------------------
class NoiseTape:
    """Records every torch.randn draw (ddim.py:128,260 via util.py:264-267)."""

    def __init__(self):
        self.draws = []
        self._orig = torch.randn

    def __enter__(self):
        def rec(*a, **k):
            r = self._orig(*a, **k)
            self.draws.append(r.detach().numpy().copy())
            return r
        torch.randn = rec
        return self

    def __exit__(self, *a):
        torch.randn = self._orig


_STREAM_MAX = [0.0]


def _watch_stream(model):
    """running max |x| over the outputs of every U-Net block (what the raw-stream operand producers of the HIP path see)"""
    net = model.model.diffusion_model
    for blk in list(net.input_blocks) + [net.middle_block] + list(net.output_blocks):
        blk.register_forward_hook(lambda m, i, o: _STREAM_MAX.__setitem__(0, max(_STREAM_MAX[0], float(o.detach().abs().max()))))


def build_frido(ucfg, vcfg, bcfg):
    fr = H.import_ref("frido.models.diffusion.frido")
    H.patch_samplers()
    cfg = frido_cfg(ucfg, vcfg, bcfg)
    cfg["first_stage_config"]["params"]["lossconfig"] = {"target": "torch.nn.Identity"}
    cfg["cond_stage_config"]["params"]["device"] = "cpu"
    model = fr.FridoDiffusion(**cfg)
    fill_module(model.model, "model.")
    fill_module(model.first_stage_model, "first_stage_model.")
    fill_module(model.cond_stage_model, "cond_stage_model.")
    model.scale_factor.copy_(torch.tensor([0.9, 1.1, 1.05][:len(vcfg["embed_dim"])]))
    _watch_stream(model)
    return model.eval()


def gen_sampler(tag, ucfg, vcfg, bcfg, B, nctx):
    DDIM, PLMS = H.patch_samplers()
    model = build_frido(ucfg, vcfg, bcfg)
    hw = ucfg["image_size"]
    C = ucfg["in_channels"]
    nstage = ucfg["num_stage"]
    tokens = torch.from_numpy(np.random.default_rng(5).integers(0, bcfg["vocab_size"], (B, nctx)))
    with torch.no_grad():
        c = model.get_learned_conditioning(tokens)
        uc = torch.zeros_like(c)  # layout2i unconditional conditioning (sample_diffusion.py:241-256)
    out = {"tokens": tokens.numpy(), "c": c.numpy()}

    def run(name, sampler_cls, S, eta, scale, log_every_t=2):
        torch.manual_seed(23)
        smp = sampler_cls(model)
        with NoiseTape() as tape, torch.no_grad():
            samples, inter = smp.sample(S=S, batch_size=B, shape=(C, hw, hw), conditioning=c,
                                        num_stage=nstage, eta=eta, verbose=False, log_every_t=log_every_t,
                                        unconditional_guidance_scale=scale,
                                        unconditional_conditioning=uc if scale != 1.0 else None)
            img = model.decode_first_stage(samples)
        out[f"{name}_samples"] = samples.numpy()
        out[f"{name}_img"] = img.numpy()
        out[f"{name}_noise"] = np.concatenate([d.reshape(-1) for d in tape.draws])
        out[f"{name}_noise_shapes"] = np.array([list(d.shape) for d in tape.draws], dtype=np.int64)
        out[f"{name}_nx"] = np.int64(len(inter["x_inter"]))
        out[f"{name}_x_inter_last"] = inter["x_inter"][-1].numpy()
        out[f"{name}_pred_x0_1"] = inter["pred_x0"][1].numpy()
        out[f"{name}_args"] = np.array([S, eta, scale, log_every_t], dtype=np.float64)

    run("ddim_eta1", DDIM, 4, 1.0, 1.0)
    run("ddim_eta0_cfg", DDIM, 5, 0.0, 1.5)
    run("plms", PLMS, 6, 0.0, 1.0)
    run("plms_cfg", PLMS, 5, 0.0, 1.5, log_every_t=3)
    # single p_sample_ddim step with all intermediates (ddim.py:188-273)
    torch.manual_seed(7)
    smp = DDIM(model)
    smp.make_schedule(ddim_num_steps=50, ddim_eta=1.0, verbose=False)
    smp.num_stage = nstage
    x = T(seeded_normal(f"{tag}:px", (B, C, hw, hw)))
    ts = torch.full((B,), int(smp.ddim_timesteps[30]), dtype=torch.long)
    with NoiseTape() as tape, torch.no_grad():
        xp, px0 = smp.p_sample_ddim(x, c, ts, nstage - 1, index=30)
    out["step_x"] = x.numpy()
    out["step_xprev"] = xp.numpy()
    out["step_predx0"] = px0.numpy()
    out["step_noise"] = tape.draws[0]
    out["stream_absmax"] = np.float64(_STREAM_MAX[0])
    print(f"  residual stream max |x| over all runs = {_STREAM_MAX[0]:.4g}")
    save(tag, **out)


# ---------------------------------------------------------------------------------------------
# Full-size end-to-end fixtures (BASELINE.json configs 1, 3, 5).  The per-step noise is NOT stored (it would be MBs of
# incompressible floats): the runs draw from torch's CPU generator after torch.manual_seed(23), which the HIP samplers'
# noise="torch" mode reproduces draw for draw; a checksum of the stream is stored so that a generator mismatch is
# reported as such.
class GoldenCorrector:
    """A score corrector in the reference's protocol (ddim.py:228-230): deterministic, depends on e_t, x and t."""

    def modify_score(self, model, e_t, x, t, c, gain=1.0):
        return gain * e_t + 0.01 * torch.tanh(x) * (t.float().view(-1, 1, 1, 1) / 1000.0)


def gen_sampler_opts():
    """Sampler options of ddim.py the shipped scripts do not use (r04): noise_dropout (ddim.py:260-262) and score_corrector
    (ddim.py:228-230), on the small 2-stage config.  The runs draw from torch's CPU generator after manual_seed(23): randn AND
    dropout masks -- the HIP sampler's noise="torch" mode makes the same draws in the same order."""
    DDIM, PLMS = H.patch_samplers()
    model = build_frido(UNET_SMALL, VQ_SMALL, BERT_SMALL)
    B, nctx = 2, 5
    tokens = torch.from_numpy(np.random.default_rng(5).integers(0, BERT_SMALL["vocab_size"], (B, nctx)))
    with torch.no_grad():
        c = model.get_learned_conditioning(tokens)
        uc = torch.zeros_like(c)
    out = {"c": c.numpy()}
    for name, kw in (("dropout", dict(eta=1.0, noise_dropout=0.25)),
                     ("corrector", dict(eta=1.0, score_corrector=GoldenCorrector(), corrector_kwargs=dict(gain=0.9))),
                     ("corrector_cfg_dropout", dict(eta=0.5, noise_dropout=0.4, score_corrector=GoldenCorrector(), corrector_kwargs=dict(gain=1.1),
                                                    unconditional_guidance_scale=1.5, unconditional_conditioning=uc))):
        torch.manual_seed(23)
        with torch.no_grad():
            samples, inter = DDIM(model).sample(S=5, batch_size=B, shape=(6, 16, 16), conditioning=c, num_stage=2, verbose=False,
                                                log_every_t=2, **kw)
        out[f"{name}_samples"] = samples.numpy()
        out[f"{name}_pred_x0_1"] = inter["pred_x0"][1].numpy()
        out[f"{name}_nx"] = np.int64(len(inter["x_inter"]))
    # PLMS with dropout: eta = 0, the dropout only consumes generator state (plms.py:247-303)
    torch.manual_seed(23)
    with torch.no_grad():
        samples, _ = PLMS(model).sample(S=5, batch_size=B, shape=(6, 16, 16), conditioning=c, num_stage=2, verbose=False, log_every_t=2,
                                        noise_dropout=0.3)
        tail = torch.randn(4)             # the generator's state after the run: every draw (randn + dropout masks) was consumed
    out["plms_dropout_samples"] = samples.numpy()
    out["plms_dropout_rng_tail"] = tail.numpy()
    # PLMS with a score corrector (applied inside every model evaluation, plms.py:236-238), with and without CFG
    for name, kw in (("plms_corrector", dict(score_corrector=GoldenCorrector(), corrector_kwargs=dict(gain=0.9))),
                     ("plms_corrector_cfg", dict(score_corrector=GoldenCorrector(), corrector_kwargs=dict(gain=1.1),
                                                 unconditional_guidance_scale=1.5, unconditional_conditioning=uc))):
        torch.manual_seed(23)
        with torch.no_grad():
            samples, inter = PLMS(model).sample(S=6, batch_size=B, shape=(6, 16, 16), conditioning=c, num_stage=2, verbose=False,
                                                log_every_t=2, **kw)
        out[f"{name}_samples"] = samples.numpy()
        out[f"{name}_pred_x0_1"] = inter["pred_x0"][1].numpy()
        out[f"{name}_nx"] = np.int64(len(inter["x_inter"]))
    save("sampler_opts", **out)


def _decode_with_codes(model, samples):
    """decode_first_stage (frido.py:823-891) + the per-scale codes (the reference drops return_code for VQ first stages)."""
    z = samples.clone()
    start = 0
    for i, e in enumerate(model.first_stage_model.embed_dim):
        z[:, start:start + e] *= 1. / model.scale_factor[i]
        start += e
    dec, code = model.first_stage_model.decode(z, return_code=True)
    return dec, np.asarray(code, dtype=np.int64)


def _pack_img(out, name, img, ss):
    out[f"{name}_img"] = img.numpy()[:, :, ::ss, ::ss]
    out[f"{name}_img_sum"] = np.float64(img.double().sum().item())
    out[f"{name}_img_abs_sum"] = np.float64(img.double().abs().sum().item())
    out[f"{name}_img_ss"] = np.int64(ss)


def _run_sampler(out, model, name, sampler_cls, S, eta, scale, c, uc, shape, nstage, log_every_t, ss, x_T=None):
    torch.manual_seed(23)
    smp = sampler_cls(model)
    with NoiseTape() as tape, torch.no_grad():
        samples, inter = smp.sample(S=S, batch_size=c.shape[0], shape=shape, conditioning=c, num_stage=nstage, eta=eta,
                                    verbose=False, log_every_t=log_every_t, unconditional_guidance_scale=scale,
                                    unconditional_conditioning=uc if scale != 1.0 else None, x_T=x_T)
        img = model.decode_first_stage(samples)
        img2, code = _decode_with_codes(model, samples)
    assert torch.equal(img, img2)
    flat = np.concatenate([d.reshape(-1) for d in tape.draws]) if tape.draws else np.zeros(0, np.float32)
    out[f"{name}_samples"] = samples.numpy()
    out[f"{name}_code"] = code.astype(np.int32)
    _pack_img(out, name, img, ss)
    out[f"{name}_noise_n"] = np.int64(flat.size)
    out[f"{name}_noise_sum"] = np.float64(flat.astype(np.float64).sum())
    out[f"{name}_noise_head"] = flat[:16].copy()
    out[f"{name}_nx"] = np.int64(len(inter["x_inter"]))
    out[f"{name}_x_inter_last"] = inter["x_inter"][-1].numpy()
    out[f"{name}_pred_x0_1"] = inter["pred_x0"][1].numpy() if len(inter["pred_x0"]) > 1 else np.zeros(0, np.float32)
    out[f"{name}_args"] = np.array([S, eta, scale, log_every_t], dtype=np.float64)
    return samples


def gen_sampler_full():
    """BASELINE config 1: layout2i f8f4 at full width, B = 1, DDIM-50 eta = 1 (ddim.py:116-186, frido.py:823-891), plus
    DDIM-4; context = the real 32-layer cond stage on random layout tokens."""
    DDIM, PLMS = H.patch_samplers()
    bcfg = dict(BERT_FULL, vocab_size=1024 + 256)
    model = build_frido(UNET_FULL, VQ_FULL, bcfg)
    B, nctx = 1, 26
    tokens = torch.from_numpy(np.random.default_rng(11).integers(0, 1024, (B, nctx)))
    with torch.no_grad():
        c = model.get_learned_conditioning(tokens)
    out = {"tokens": tokens.numpy(), "c": c.numpy(), "scale_factor": model.scale_factor.numpy()}
    _run_sampler(out, model, "ddim4", DDIM, 4, 1.0, 1.0, c, None, (6, 64, 64), 2, 2, 4)
    if not PROFILE:      # (the heavy-profile fixture keeps the 8 forwards of DDIM-4: the dynamic range is the point, not the step count)
        _run_sampler(out, model, "ddim50", DDIM, 50, 1.0, 1.0, c, None, (6, 64, 64), 2, 10, 4)
    out["stream_absmax"] = np.float64(_STREAM_MAX[0])
    save("sampler_full", **out)


class _Ident(torch.nn.Module):
    def forward(self, x):
        return x


def _frido_no_cond(ucfg, vcfg, nscale):
    fr = H.import_ref("frido.models.diffusion.frido")
    H.patch_samplers()
    cfg = frido_cfg(ucfg, vcfg, dict())
    cfg["first_stage_config"]["params"]["lossconfig"] = {"target": "torch.nn.Identity"}
    cfg["cond_stage_config"] = {"target": "torch.nn.Identity"}      # conditioning tensors are fed directly
    cfg["cond_stage_trainable"] = False
    model = fr.FridoDiffusion(**cfg)
    fill_module(model.model, "model.")
    fill_module(model.first_stage_model, "first_stage_model.")
    model.scale_factor.copy_(torch.tensor([0.9, 1.1, 1.05][:nscale]))
    return model.eval()


def gen_sampler_t2i():
    """BASELINE config 3 at its true dimensions (configs/frido/t2i/frido_f16f8_coco_clip.yaml:21-77): latent 8 x 32 x 32,
    split [4, 4], context = ONE L2-normalised 768-d token (encoders/modules.py:210-219), PLMS (plms.py:116-194) with
    classifier-free guidance 1.5 (tools/frido/eval_t2i_clip.sh), f16f8 first stage with 2 x 8192 codes."""
    DDIM, PLMS = H.patch_samplers()
    model = _frido_no_cond(UNET_F16F8, VQ_F16F8, 2)
    B = 2
    c = T(seeded_normal("t2i_full:c", (B, 1, 768)))
    c = c / torch.linalg.norm(c, dim=2, keepdim=True)
    uc = T(seeded_normal("t2i_full:uc", (1, 1, 768)))
    uc = (uc / torch.linalg.norm(uc, dim=2, keepdim=True)).repeat(B, 1, 1)
    out = {"c": c.numpy(), "uc": uc.numpy(), "scale_factor": model.scale_factor.numpy()}
    _run_sampler(out, model, "plms_cfg", PLMS, 6, 0.0, 1.5, c, uc, (8, 32, 32), 2, 2, 4)
    _run_sampler(out, model, "ddim_cfg", DDIM, 4, 0.0, 1.5, c, uc, (8, 32, 32), 2, 2, 4)
    # one denoiser forward per stage (pyunet.py:867-950) for a tight single-forward bound
    net = model.model.diffusion_model
    x = T(seeded_normal("t2i_full:x", (B, 8, 32, 32)))
    out["x"] = x.numpy()
    for s in range(2):
        t = torch.tensor([996 - 37 * i for i in range(B)], dtype=torch.long)
        with torch.no_grad():
            e = net(x[:, :4 * (s + 1)].contiguous(), t, context=c, stage=s)
        out[f"t_{s}"], out[f"eps_{s}"] = t.numpy(), e.numpy()
    save("sampler_t2i", **out)


def gen_unet_512():
    """BASELINE config 5 denoiser: 3 stages on a 9 x 128 x 128 latent, 92 context tokens; one forward per stage."""
    net = build_unet(UNET_512)
    B, nctx, hw = 1, 92, 128
    x = T(seeded_normal("u512:x", (B, 9, hw, hw)))
    ctx = T(seeded_normal("u512:ctx", (B, nctx, 640)))
    out = {"x": x.numpy(), "ctx": ctx.numpy()}
    for s in range(3):
        t = torch.tensor([801 - 250 * s], dtype=torch.long)
        with torch.no_grad():
            e = net(x[:, :3 * (s + 1)].contiguous(), t, context=ctx, stage=s)
        out[f"t_{s}"], out[f"eps_{s}"] = t.numpy(), e.numpy()
        print("stage", s, "done", float(e.abs().max()))
    save("unet_512", **out)


def gen_vq_512():
    """BASELINE config 5 first stage: decode of a 9 x 128 x 128 latent to 512 x 512 (AttnBlock over 16384 keys,
    taming/modules/diffusionmodules/model.py:168-192)."""
    net = build_vq(VQ_512)
    h = T(seeded_normal("vq512:h", (1, 9, 128, 128)) * 1.5)
    with torch.no_grad():
        dec, code = net.decode(h, return_code=True)
    out = {"h": h.numpy(), "code": np.asarray(code, dtype=np.int32)}
    _pack_img(out, "dec", dec, 8)
    save("vq_512", **out)


def gen_vq_full_enc():
    """SURVEY a16 at full width: VQModelInterface.encode (msvqgan.py:326-374: MSEncoder, coarse-to-fine with VQ + ConvTranspose +
    shared decoder) of one 256 x 256 image with the layout2i f8f4 first stage."""
    net = build_vq(VQ_FULL)
    x = T(np.tanh(seeded_normal("vq_full:img", (1, 3, 256, 256))))
    with torch.no_grad():
        enc = net.encode(x)
    save("vq_full_enc", enc=enc.numpy())          # the image is regenerated from its seed by the tests


def gen_sampler_xt():
    """The x_T quirk (ddim.py:150-152, plms.py:150-152): a supplied x_T is taken as the FINISHED stage-0 result -- stage 0
    and its pooling hand-off are skipped."""
    DDIM, PLMS = H.patch_samplers()
    model = build_frido(UNET_SMALL, VQ_SMALL, BERT_SMALL)
    g = np.load(os.path.join(HERE, "sampler_small.npz"))
    c = T(g["c"])
    out = {}
    xT = T(seeded_normal("xt:x", (2, 6, 16, 16)))
    out["x_T"] = xT.numpy()
    for name, cls, S, eta in (("ddim", DDIM, 4, 1.0), ("plms", PLMS, 4, 0.0)):
        torch.manual_seed(23)
        with NoiseTape() as tape, torch.no_grad():
            samples, inter = cls(model).sample(S=S, batch_size=2, shape=(6, 16, 16), conditioning=c, num_stage=2, eta=eta,
                                               verbose=False, log_every_t=2, x_T=xT)
        out[f"{name}_samples"] = samples.numpy()
        out[f"{name}_noise"] = np.concatenate([d.reshape(-1) for d in tape.draws])
        out[f"{name}_nx"] = np.int64(len(inter["x_inter"]))
    # single-stage model: x_T comes back unchanged
    save("sampler_xt", **out)


def gen_sampler_ddim200():
    """BASELINE config 2's step count on config 1's plumbing: layout2i f8f4 at full width, B = 1, DDIM-200 eta = 1 (400 denoiser
    forwards of the reference, ddim.py:116-186) + decode.  Same tokens / context as sampler_full."""
    DDIM, PLMS = H.patch_samplers()
    bcfg = dict(BERT_FULL, vocab_size=1024 + 256)
    model = build_frido(UNET_FULL, VQ_FULL, bcfg)
    tokens = torch.from_numpy(np.random.default_rng(11).integers(0, 1024, (1, 26)))
    with torch.no_grad():
        c = model.get_learned_conditioning(tokens)
    out = {"tokens": tokens.numpy(), "c": c.numpy(), "scale_factor": model.scale_factor.numpy()}
    _run_sampler(out, model, "ddim200", DDIM, 200, 1.0, 1.0, c, None, (6, 64, 64), 2, 50, 4)
    save("sampler_ddim200", **out)


def gen_sampler_512():
    """BASELINE config 5, MULTI-STEP at its true size: the 3-stage DDIM loop (two hand-offs incl. the 4x4 block mean of stage 0,
    ddim.py:146-149,177-185) on the 9 x 128 x 128 latent with 92 context tokens, S = 2, eta = 1, + the 512 x 512 decode."""
    DDIM, PLMS = H.patch_samplers()
    model = _frido_no_cond(UNET_512, VQ_512, 3)
    c = T(seeded_normal("s512:ctx", (1, 92, 640)))
    out = {"c": c.numpy(), "scale_factor": model.scale_factor.numpy()}
    _run_sampler(out, model, "ddim2", DDIM, 2, 1.0, 1.0, c, None, (9, 128, 128), 3, 1, 8)
    save("sampler_512", **out)


def gen_shipped_cfgs():
    """The `model:` tree of every configs/frido/**/*.yaml the reference ships, as JSON (data: the host test feeds each one to
    instantiate_from_config)."""
    import glob
    import json
    import yaml
    root = os.path.join(H.REF if hasattr(H, "REF") else "/root/reference", "configs")
    out = {os.path.relpath(f, root): yaml.safe_load(open(f))["model"] for f in sorted(glob.glob(os.path.join(root, "frido", "*", "*.yaml")))}
    with open(os.path.join(HERE, "shipped_model_cfgs.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("shipped_model_cfgs.json:", len(out), "configs")


GENS = {
    "schedules": gen_schedules,
    "unet_small": lambda: gen_unet("unet_small", UNET_SMALL, B=2, nctx=5, hw=16),
    "unet_small_d2": lambda: gen_unet("unet_small_d2", UNET_SMALL_D2, B=2, nctx=5, hw=16),
    "unet_small3": lambda: gen_unet("unet_small3", UNET_SMALL3, B=1, nctx=7, hw=16),
    "unet_full": lambda: gen_unet("unet_full", UNET_FULL, B=1, nctx=26, hw=64, capture=False),
    "vq_small": lambda: gen_vq("vq_small", VQ_SMALL, B=2),
    "vq_full": lambda: gen_vq("vq_full", VQ_FULL, B=1, subsample=8),
    "sampler_small": lambda: gen_sampler("sampler_small", UNET_SMALL, VQ_SMALL, BERT_SMALL, B=2, nctx=5),
    "sampler_opts": gen_sampler_opts,
    "sampler_small3": lambda: gen_sampler("sampler_small3", UNET_SMALL3, VQ_SMALL3, BERT_SMALL, B=1, nctx=7),
    "sampler_xt": gen_sampler_xt,
    "vq_full_enc": gen_vq_full_enc,
    "sampler_full": gen_sampler_full,
    "sampler_t2i": gen_sampler_t2i,
    "unet_512": gen_unet_512,
    "vq_512": gen_vq_512,
    "sampler_ddim200": gen_sampler_ddim200,
    "sampler_512": gen_sampler_512,
    "shipped_cfgs": gen_shipped_cfgs,
}

if __name__ == "__main__":
    if "--profile" in sys.argv:
        i = sys.argv.index("--profile")
        PROFILE = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    names = sys.argv[1:] or list(GENS)
    torch.set_num_threads(8)
    for n in names:
        print("==", n)
        GENS[n]()
