"""The oracle (CPU restatement of the reference) against the golden fixtures captured from the reference's own
modules (tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then pin the HIP path to it."""
import numpy as np
import pytest
import torch

from helpers import golden, synth_sd, unet_holder, vq_holder
from golden_cfg import UNET_SMALL, UNET_SMALL_D2, UNET_SMALL3, UNET_FULL, VQ_SMALL, VQ_SMALL3, VQ_FULL
from oracle import samplers as S
from oracle.unet import unet_forward, timestep_embedding
from oracle.vqgan import vq_decode, vq_encode, quantize


def test_schedules_and_timestep_embedding():
    g = golden("schedules")
    betas = S.make_betas()
    assert np.array_equal(betas, g["betas64"])
    ac = S.alphas_cumprod_f32(betas)
    assert np.array_equal(ac.numpy(), g["alphas_cumprod"])
    for Sx in (4, 50, 100, 200, 250):
        ts = S.ddim_timesteps(Sx)
        assert np.array_equal(ts, g[f"ts_{Sx}"])
        for eta in (0.0, 1.0):
            sig, a, ap = S.ddim_params(ac, ts, eta)
            tag = f"{Sx}_{int(eta)}"
            assert np.array_equal(sig, g[f"sigmas_{tag}"]) and np.array_equal(a, g[f"alphas_{tag}"])
            assert np.array_equal(ap, g[f"alphas_prev_{tag}"])
    t = torch.from_numpy(g["temb_t"])
    assert np.array_equal(timestep_embedding(t, 192).numpy(), g["temb_192"])
    assert np.array_equal(timestep_embedding(t, 32).numpy(), g["temb_32"])


@pytest.mark.parametrize("name,cfg", [("unet_small", UNET_SMALL), ("unet_small3", UNET_SMALL3), ("unet_small_d2", UNET_SMALL_D2)])
def test_unet_oracle_bit_exact(name, cfg):
    g = golden(name)
    root = unet_holder(cfg)
    assert sorted(root.state_dict().keys()) == list(g["keys"])          # state_dict layout == reference
    assert sum(p.numel() for p in root.parameters()) == int(g["nparam"])
    sd = synth_sd(root, "model.diffusion_model.")
    x, ctx = torch.from_numpy(g["x"]), torch.from_numpy(g["ctx"])
    splits = cfg["split_embed_dim_list"]
    for s in range(cfg["num_stage"]):
        taps = {}
        e = unet_forward(sd, cfg, x[:, :sum(splits[:s + 1])], torch.from_numpy(g[f"t_{s}"]), ctx, s, taps=taps)
        assert torch.equal(e, torch.from_numpy(g[f"eps_{s}"]))
        for k, v in taps.items():
            assert torch.equal(v, torch.from_numpy(g[f"{k}_{s}"])), k


@pytest.mark.slow
def test_unet_full_width_oracle():
    g = golden("unet_full")
    root = unet_holder(UNET_FULL)
    assert sum(p.numel() for p in root.parameters()) == int(g["nparam"])
    sd = synth_sd(root, "model.diffusion_model.")
    x, ctx = torch.from_numpy(g["x"]), torch.from_numpy(g["ctx"])
    e = unet_forward(sd, UNET_FULL, x[:, :3], torch.from_numpy(g["t_0"]), ctx, 0)
    assert float((e - torch.from_numpy(g["eps_0"])).abs().max()) < 1e-5


def test_vq_oracle_decode_encode_and_ties():
    g = golden("vq_small")
    root = vq_holder(VQ_SMALL)
    assert sorted(root.state_dict().keys()) == list(g["keys"])
    sd = synth_sd(root, "first_stage_model.")
    dec, codes = vq_decode(sd, VQ_SMALL, torch.from_numpy(g["h"]), return_code=True)
    assert all(np.array_equal(c.numpy(), g["code"][i]) for i, c in enumerate(codes))
    assert float((dec - torch.from_numpy(g["dec"])).abs().max()) < 1e-5
    enc = vq_encode(sd, VQ_SMALL, torch.from_numpy(g["img"]))
    assert float((enc - torch.from_numpy(g["enc"])).abs().max()) < 1e-5
    cb = sd["first_stage_model.ms_quantize.0.embedding.weight"].clone()
    zq, idx = quantize(cb, torch.from_numpy(g["h"])[:, :3])
    assert np.array_equal(idx.numpy(), g["idx0"]) and float((zq - torch.from_numpy(g["zq0"])).abs().max()) < 1e-6
    cb[7] = cb[3]                                              # duplicated code: lowest index wins (quantize.py:281)
    _, idx_t = quantize(cb, cb[3].view(1, -1, 1, 1).repeat(1, 1, 2, 2))
    assert np.array_equal(idx_t.numpy(), g["idx_tie"]) and set(idx_t.tolist()) == {3}


@pytest.mark.parametrize("name,ucfg,vcfg", [("sampler_small", UNET_SMALL, VQ_SMALL), ("sampler_small3", UNET_SMALL3, VQ_SMALL3)])
def test_sampler_oracle_bit_exact(name, ucfg, vcfg):
    g = golden(name)
    usd = synth_sd(unet_holder(ucfg), "model.diffusion_model.")
    vsd = synth_sd(vq_holder(vcfg), "first_stage_model.")
    ac = S.alphas_cumprod_f32(S.make_betas())
    c = torch.from_numpy(g["c"])
    uc = torch.zeros_like(c)
    am = lambda x, t, cond, s: unet_forward(usd, ucfg, x, t, cond, s)
    splits, embed, ns = ucfg["split_embed_dim_list"], vcfg["embed_dim"], ucfg["num_stage"]
    shape = (c.shape[0], ucfg["in_channels"], 16, 16)
    for run, fn, kw in [("ddim_eta1", S.ddim_sample, dict(eta=1.0)), ("ddim_eta0_cfg", S.ddim_sample, dict(eta=0.0)),
                        ("plms", S.plms_sample, {}), ("plms_cfg", S.plms_sample, {})]:
        Sx, eta, scale, lev = g[f"{run}_args"]
        out, inter = fn(am, ac, int(Sx), shape, c, splits, embed, ns, scale=float(scale), uc=uc,
                        noise=S.NoiseSource(g[f"{run}_noise"]), log_every_t=int(lev), **kw)
        assert torch.equal(out, torch.from_numpy(g[f"{run}_samples"])), run
        assert len(inter["x_inter"]) == int(g[f"{run}_nx"])
        assert torch.equal(inter["x_inter"][-1], torch.from_numpy(g[f"{run}_x_inter_last"]))
        img = S.decode_first_stage(lambda z: vq_decode(vsd, vcfg, z), out, [0.9, 1.1, 1.05][:len(embed)], embed)
        assert float((img - torch.from_numpy(g[f"{run}_img"])).abs().max()) < 1e-5
    torch.manual_seed(23)   # same seed as the reference run -> same noise stream -> same samples
    out, _ = S.ddim_sample(am, ac, 4, shape, c, splits, embed, ns, eta=1.0, log_every_t=2)
    assert torch.equal(out, torch.from_numpy(g["ddim_eta1_samples"]))
    # single p_sample_ddim step (ddim.py:188-273) incl. stage masks
    sig, al, alp = S.ddim_params(ac, S.ddim_timesteps(50), 1.0)
    x = torch.from_numpy(g["step_x"])
    e = am(x, torch.full((x.shape[0],), int(S.ddim_timesteps(50)[30])), c, ns - 1)
    start = sum(embed[:ns - 1])
    e = torch.cat((torch.zeros(e.size(0), start, 16, 16), e), dim=1)
    xp, px0 = S._x_prev(x, e, al[30], alp[30], sig[30], np.sqrt(1.0 - al)[30], start, torch.from_numpy(g["step_noise"]))
    assert torch.equal(xp, torch.from_numpy(g["step_xprev"])) and torch.equal(px0, torch.from_numpy(g["step_predx0"]))


def test_bert_oracle_bit_exact():
    """Cond stage (BERTEmbedder): oracle vs the `c` tensors the reference produced from the same token ids."""
    from golden_cfg import BERT_SMALL
    from frido_amd.models import BERTEmbedder
    from oracle.bert import bert_embed
    m = BERTEmbedder(**BERT_SMALL)
    sd = synth_sd(m, "cond_stage_model.")
    for name in ("sampler_small", "sampler_small3"):
        g = golden(name)
        c = bert_embed(sd, torch.from_numpy(g["tokens"]), BERT_SMALL["n_layer"])
        assert torch.equal(c, torch.from_numpy(g["c"]))


def test_sampler_oracle_x_T_quirk():
    """ddim.py:150-152 / plms.py:150-152: x_T is adopted as the finished stage-0 result."""
    g = golden("sampler_xt")
    c = torch.from_numpy(golden("sampler_small")["c"])
    usd = synth_sd(unet_holder(UNET_SMALL), "model.diffusion_model.")
    ac = S.alphas_cumprod_f32(S.make_betas())
    am = lambda x, t, cond, s: unet_forward(usd, UNET_SMALL, x, t, cond, s)
    xT = torch.from_numpy(g["x_T"])
    out, inter = S.ddim_sample(am, ac, 4, (2, 6, 16, 16), c, [3, 3], [3, 3], 2, eta=1.0, noise=S.NoiseSource(g["ddim_noise"]),
                               log_every_t=2, x_T=xT)
    assert torch.equal(out, torch.from_numpy(g["ddim_samples"])) and len(inter["x_inter"]) == int(g["ddim_nx"])
    assert torch.equal(out[:, :3], xT[:, :3])            # stage-0 channels pass through un-denoised and un-pooled
    out, inter = S.plms_sample(am, ac, 4, (2, 6, 16, 16), c, [3, 3], [3, 3], 2, noise=S.NoiseSource(g["plms_noise"]),
                               log_every_t=2, x_T=xT)
    assert torch.equal(out, torch.from_numpy(g["plms_samples"])) and len(inter["x_inter"]) == int(g["plms_nx"])
    one, _ = S.ddim_sample(am, ac, 4, (2, 3, 16, 16), c, [3], [3], 1, eta=1.0, x_T=xT[:, :3])
    assert torch.equal(one, xT[:, :3])                   # single stage: x_T comes back unchanged


def _check_img(img, g, name, tol):
    ss = int(g[f"{name}_img_ss"])
    ref = torch.from_numpy(g[f"{name}_img"])
    assert float((img[:, :, ::ss, ::ss] - ref).abs().max()) < tol
    assert abs(float(img.double().sum()) - float(g[f"{name}_img_sum"])) < 1e-5 * float(g[f"{name}_img_abs_sum"])


@pytest.mark.slow
def test_t2i_true_dims_oracle():
    """BASELINE config 3 at the reference's real f16f8 dimensions: PLMS + CFG 1.5, one 768-d context token."""
    from golden_cfg import UNET_F16F8, VQ_F16F8
    g = golden("sampler_t2i")
    usd = synth_sd(unet_holder(UNET_F16F8), "model.diffusion_model.")
    vsd = synth_sd(vq_holder(VQ_F16F8), "first_stage_model.")
    ac = S.alphas_cumprod_f32(S.make_betas())
    c, uc = torch.from_numpy(g["c"]), torch.from_numpy(g["uc"])
    am = lambda x, t, cond, s: unet_forward(usd, UNET_F16F8, x, t, cond, s)
    x = torch.from_numpy(g["x"])
    for s in range(2):
        e = am(x[:, :4 * (s + 1)], torch.from_numpy(g[f"t_{s}"]), c, s)
        assert float((e - torch.from_numpy(g[f"eps_{s}"])).abs().max()) < 1e-5
    Sx, eta, scale, lev = g["plms_cfg_args"]
    torch.manual_seed(23)
    ns = S.NoiseSource()
    out, inter = S.plms_sample(am, ac, int(Sx), (2, 8, 32, 32), c, [4, 4], [4, 4], 2, scale=float(scale), uc=uc, noise=ns,
                               log_every_t=int(lev))
    assert float((out - torch.from_numpy(g["plms_cfg_samples"])).abs().max()) < 2e-5
    assert len(inter["x_inter"]) == int(g["plms_cfg_nx"])
    img, codes = vq_decode(vsd, VQ_F16F8, S.decode_first_stage(lambda z: z, out, g["scale_factor"].tolist(), [4, 4]), return_code=True)
    assert np.mean(np.stack([cd.numpy() for cd in codes]).reshape(-1) != g["plms_cfg_code"].reshape(-1)) < 1e-3
    # decode on the reference's own latent: the VQ decisions then agree and the pixels are comparable everywhere
    img = S.decode_first_stage(lambda z: vq_decode(vsd, VQ_F16F8, z), torch.from_numpy(g["plms_cfg_samples"]), g["scale_factor"].tolist(), [4, 4])
    _check_img(img, g, "plms_cfg", 2e-5)


@pytest.mark.slow
def test_512_three_scale_oracle():
    """BASELINE config 5: denoiser on the 9 x 128 x 128 latent (stage 2: two SPADE-conditioned pyramids), and the 512 x 512
    decode with 16384-key attention blocks."""
    from golden_cfg import UNET_512, VQ_512
    g = golden("unet_512")
    usd = synth_sd(unet_holder(UNET_512), "model.diffusion_model.")
    x, ctx = torch.from_numpy(g["x"]), torch.from_numpy(g["ctx"])
    e = unet_forward(usd, UNET_512, x, torch.from_numpy(g["t_2"]), ctx, 2)
    assert float((e - torch.from_numpy(g["eps_2"])).abs().max()) < 2e-5
    del usd
    gv = golden("vq_512")
    vsd = synth_sd(vq_holder(VQ_512), "first_stage_model.")
    dec, codes = vq_decode(vsd, VQ_512, torch.from_numpy(gv["h"]), return_code=True)
    assert np.array_equal(np.stack([cd.numpy() for cd in codes]).reshape(gv["code"].shape), gv["code"])
    _check_img(dec, gv, "dec", 2e-5)


@pytest.mark.slow
def test_512_three_stage_multistep_oracle():
    """BASELINE config 5, multi-step at its true size: the oracle's 3-stage DDIM-2 loop on the 9 x 128 x 128 latent (both hand-offs,
    incl. the 4 x 4 block mean of stage 0) vs the reference's run (tests/golden/make_golden.py sampler_512)."""
    from golden_cfg import UNET_512
    g = golden("sampler_512")
    usd = synth_sd(unet_holder(UNET_512), "model.diffusion_model.")
    ac = S.alphas_cumprod_f32(S.make_betas())
    am = lambda x, t, cond, s: unet_forward(usd, UNET_512, x, t, cond, s)
    torch.manual_seed(23)
    out, inter = S.ddim_sample(am, ac, 2, (1, 9, 128, 128), torch.from_numpy(g["c"]), [3, 3, 3], [3, 3, 3], 3, eta=1.0,
                               log_every_t=int(g["ddim2_args"][3]))
    assert len(inter["x_inter"]) == int(g["ddim2_nx"])
    assert float((out - torch.from_numpy(g["ddim2_samples"])).abs().max()) < 2e-5 * float(np.abs(g["ddim2_samples"]).max())


@pytest.mark.slow
def test_full_width_ddim4_oracle():
    """BASELINE config 1 plumbing at DDIM-4: the real 32-layer cond stage, both stages at full width, decode."""
    from golden_cfg import BERT_FULL
    from frido_amd.models import BERTEmbedder
    from oracle.bert import bert_embed
    g = golden("sampler_full")
    bcfg = dict(BERT_FULL, vocab_size=1024 + 256)
    c = bert_embed(synth_sd(BERTEmbedder(**bcfg), "cond_stage_model."), torch.from_numpy(g["tokens"]), bcfg["n_layer"])
    assert float((c - torch.from_numpy(g["c"])).abs().max()) < 1e-5
    usd = synth_sd(unet_holder(UNET_FULL), "model.diffusion_model.")
    vsd = synth_sd(vq_holder(VQ_FULL), "first_stage_model.")
    ac = S.alphas_cumprod_f32(S.make_betas())
    am = lambda x, t, cond, s: unet_forward(usd, UNET_FULL, x, t, cond, s)
    torch.manual_seed(23)
    out, inter = S.ddim_sample(am, ac, 4, (1, 6, 64, 64), torch.from_numpy(g["c"]), [3, 3], [3, 3], 2, eta=1.0, log_every_t=2)
    assert float((out - torch.from_numpy(g["ddim4_samples"])).abs().max()) < 2e-5
    img = S.decode_first_stage(lambda z: vq_decode(vsd, VQ_FULL, z), torch.from_numpy(g["ddim4_samples"]), g["scale_factor"].tolist(), [3, 3])
    _check_img(img, g, "ddim4", 2e-5)


@pytest.mark.slow
def test_vq_full_width_encode_oracle():
    """SURVEY a16 at full width: the oracle's encode vs the reference's (coarse channels exact to rounding; the fine scale
    passes through a VQ of the coarse one, so a single code flip would show up as an outlier -- there is none on CPU)."""
    from frido_amd.synth import seeded_normal
    g = golden("vq_full_enc")
    sd = synth_sd(vq_holder(VQ_FULL), "first_stage_model.")
    x = torch.from_numpy(np.tanh(seeded_normal("vq_full:img", (1, 3, 256, 256))))
    enc = vq_encode(sd, VQ_FULL, x)
    assert enc.shape == g["enc"].shape and float((enc - torch.from_numpy(g["enc"])).abs().max()) < 2e-5


def _golden_corrector(gain):
    """tests/golden/make_golden.py GoldenCorrector.modify_score in the oracle's callable form."""
    return lambda e_t, x, t, c: gain * e_t + 0.01 * torch.tanh(x) * (t.float().view(-1, 1, 1, 1) / 1000.0)


def test_sampler_options_oracle_bit_exact():
    """ddim.py:228-230 (score_corrector) and :260-262 (noise_dropout) against the reference's own runs (sampler_opts fixture): the
    oracle draws randn AND the dropout masks from torch's CPU generator in the reference's order, so the same seed gives the
    same samples, bit for bit."""
    g = golden("sampler_opts")
    c = torch.from_numpy(g["c"])
    usd = synth_sd(unet_holder(UNET_SMALL), "model.diffusion_model.")
    ac = S.alphas_cumprod_f32(S.make_betas())
    am = lambda x, t, cond, s: unet_forward(usd, UNET_SMALL, x, t, cond, s)
    runs = (("dropout", dict(eta=1.0, noise_dropout=0.25)),
            ("corrector", dict(eta=1.0, score_corrector=_golden_corrector(0.9))),
            ("corrector_cfg_dropout", dict(eta=0.5, noise_dropout=0.4, score_corrector=_golden_corrector(1.1), scale=1.5, uc=torch.zeros_like(c))))
    for name, kw in runs:
        torch.manual_seed(23)
        out, inter = S.ddim_sample(am, ac, 5, (2, 6, 16, 16), c, [3, 3], [3, 3], 2, log_every_t=2, **kw)
        assert torch.equal(out, torch.from_numpy(g[f"{name}_samples"])), name
        assert torch.equal(inter["pred_x0"][1], torch.from_numpy(g[f"{name}_pred_x0_1"])) and len(inter["x_inter"]) == int(g[f"{name}_nx"])
    for name, kw in (("plms_corrector", dict(score_corrector=_golden_corrector(0.9))),
                     ("plms_corrector_cfg", dict(score_corrector=_golden_corrector(1.1), scale=1.5, uc=torch.zeros_like(c)))):
        torch.manual_seed(23)
        out, inter = S.plms_sample(am, ac, 6, (2, 6, 16, 16), c, [3, 3], [3, 3], 2, log_every_t=2, **kw)
        assert torch.equal(out, torch.from_numpy(g[f"{name}_samples"])), name
        assert torch.equal(inter["pred_x0"][1], torch.from_numpy(g[f"{name}_pred_x0_1"])) and len(inter["x_inter"]) == int(g[f"{name}_nx"])


# ---- round 5: the same pins under a trained-checkpoint-like dynamic range (frido_amd/synth.py profile "heavy": heavy-tailed weights,
#      GroupNorm / LayerNorm scales in [0.2, 3], 0.5-sigma biases, a residual stream in the thousands) --------------------------------
def test_oracle_heavy_profile_unet_vq_sampler():
    g = golden("unet_small_heavy")
    assert str(g["filler_profile"]) == "heavy" and 1.0e3 < float(g["stream_absmax_1"]) < 6.0e4
    sd = synth_sd(unet_holder(UNET_SMALL), "model.diffusion_model.", "heavy")
    x, ctx = torch.from_numpy(g["x"]), torch.from_numpy(g["ctx"])
    for s in range(2):
        e = unet_forward(sd, UNET_SMALL, x[:, :3 * (s + 1)], torch.from_numpy(g[f"t_{s}"]), ctx, s)
        assert torch.equal(e, torch.from_numpy(g[f"eps_{s}"])), s
    gv = golden("vq_small_heavy")
    vsd = synth_sd(vq_holder(VQ_SMALL), "first_stage_model.", "heavy")
    dec, codes = vq_decode(vsd, VQ_SMALL, torch.from_numpy(gv["h"]), return_code=True)
    assert all(np.array_equal(c.numpy(), gv["code"][i]) for i, c in enumerate(codes))
    assert float((dec - torch.from_numpy(gv["dec"])).abs().max()) < 1e-5 * max(1.0, float(np.abs(gv["dec"]).max()))
    gs = golden("sampler_small_heavy")
    ac = S.alphas_cumprod_f32(S.make_betas())
    c = torch.from_numpy(gs["c"])
    am = lambda xx, t, cond, s: unet_forward(sd, UNET_SMALL, xx, t, cond, s)
    for run, fn, kw in [("ddim_eta1", S.ddim_sample, dict(eta=1.0)), ("plms_cfg", S.plms_sample, {})]:
        Sx, eta, scale, lev = gs[f"{run}_args"]
        out, _ = fn(am, ac, int(Sx), (2, 6, 16, 16), c, [3, 3], [3, 3], 2, scale=float(scale), uc=torch.zeros_like(c),
                    noise=S.NoiseSource(gs[f"{run}_noise"]), log_every_t=int(lev), **kw)
        assert torch.equal(out, torch.from_numpy(gs[f"{run}_samples"])), run


def test_oracle_block_walk_is_its_own_and_agrees_with_the_product_walk():
    """(r06, r05 verdict hygiene) The oracle reads its block list off the CHECKPOINT KEYS (oracle/walk.py: children in index order, kind from
    the parameter names a child owns -- how the reference's nn.Sequential containers execute), and imports nothing under frido_amd/.  The
    product's plan builders walk frido_amd/arch.py (a restatement of the constructor loops).  Two independent readings: they must list the
    same blocks, in the same order, under the same prefixes, for every configuration a fixture uses."""
    import ast
    import os
    from golden_cfg import UNET_F16F8, UNET_512, VQ_F16F8, VQ_512
    from frido_amd.arch import decoder_arch, encoder_arch, unet_arch
    from oracle import walk
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in sorted(os.listdir(os.path.join(root, "oracle"))):
        if name.endswith(".py"):
            tree = ast.parse(open(os.path.join(root, "oracle", name)).read())
            mods = [n.module or "" for n in ast.walk(tree) if isinstance(n, ast.ImportFrom)] + [a.name for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names]
            assert not [m for m in mods if m.split(".")[0] in ("frido_amd", "frido", "taming", "ldm")], (name, mods)
    flat = lambda groups: [[(b.kind, b.prefix) for b in g] for g in groups]
    for cfg in (UNET_SMALL, UNET_SMALL_D2, UNET_SMALL3, UNET_FULL, UNET_F16F8, UNET_512):
        sd = {"model.diffusion_model." + k: None for k in unet_holder(cfg).state_dict()}
        ib, mid, ob, depth = walk.unet_blocks(sd, "model.diffusion_model.")
        a = unet_arch(cfg)
        assert flat(ib) == flat(a.input_blocks) and flat([mid]) == flat([a.middle]) and flat(ob) == flat(a.output_blocks)
        assert depth == a.transformer_depth
    for cfg in (VQ_SMALL, VQ_SMALL3, VQ_FULL, VQ_F16F8, VQ_512):
        sd = {"first_stage_model." + k: None for k in vq_holder(cfg).state_dict()}
        assert flat([walk.decoder_blocks(sd, "first_stage_model.")]) == flat([decoder_arch(cfg["ddconfig"]).body])
        down, heads = walk.encoder_blocks(sd, "first_stage_model.")
        e = encoder_arch(cfg["edconfig"])
        assert flat(down) == flat(e.down) and flat(heads) == flat(e.heads)
