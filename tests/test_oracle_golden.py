"""The oracle (CPU restatement of the reference) against the golden fixtures captured from the reference's own
modules (tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then pin the HIP path to it."""
import numpy as np
import pytest
import torch

from helpers import golden, synth_sd, unet_holder, vq_holder
from golden_cfg import UNET_SMALL, UNET_SMALL3, UNET_FULL, VQ_SMALL, VQ_SMALL3, VQ_FULL
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


@pytest.mark.parametrize("name,cfg", [("unet_small", UNET_SMALL), ("unet_small3", UNET_SMALL3)])
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
