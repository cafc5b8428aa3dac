"""CPU-side checks of the host logic: schedule tables vs the reference goldens, the C-ABI library (loads, exports
every declared symbol, struct sizes agree with the header), the reference's class/config surface, and the absence
of any CPU fallback."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from helpers import golden
from golden_cfg import UNET_SMALL, VQ_SMALL, BERT_SMALL, UNET_FULL, frido_cfg
from frido_amd import _lib, schedules


def test_schedule_tables_match_reference_bitwise():
    g = golden("schedules")
    betas = schedules.make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0155)
    assert np.array_equal(betas, g["betas64"])
    tabs = schedules.ddpm_tables(betas)
    assert np.array_equal(tabs["betas"], g["betas"]) and np.array_equal(tabs["alphas_cumprod"], g["alphas_cumprod"])
    for S in (4, 50, 100, 200, 250):
        ts = schedules.make_ddim_timesteps("uniform", S, 1000)
        assert np.array_equal(ts, g[f"ts_{S}"])
        for eta in (0.0, 1.0):
            sig, a, ap = schedules.make_ddim_sampling_parameters(tabs["alphas_cumprod"], ts, eta)
            tag = f"{S}_{int(eta)}"
            assert np.array_equal(a, g[f"alphas_{tag}"]) and np.array_equal(ap, g[f"alphas_prev_{tag}"])
            assert np.array_equal(sig, g[f"sigmas_{tag}"])
            tab, t_loop = schedules.sampler_coef_table(tabs["alphas_cumprod"], S, eta)
            n = ts.shape[0]
            assert tab.shape == (n, schedules.COEF_ROW) and np.array_equal(t_loop, ts[::-1])
            assert np.array_equal(tab[:, 0], a[::-1].astype(np.float32))
            assert np.array_equal(tab[:, 1], ap[::-1].astype(np.float32))
            assert np.array_equal(tab[:, 3], g[f"sqrt1m_{tag}"][::-1].astype(np.float32))
    tab, _ = schedules.sampler_coef_table(tabs["alphas_cumprod"], 50, 0.0, plms=True)
    assert tab[0, 4:9].tolist() == [1, 1, 0, 0, 2] and tab[1, 4:9].tolist() == [3, -1, 0, 0, 2]
    assert tab[2, 4:9].tolist() == [23, -16, 5, 0, 12] and tab[7, 4:9].tolist() == [55, -59, 37, -9, 24]


def test_abi_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"libfrido_hip.so lacks {name} (declared in include/frido_hip.h)"
    assert L.frido_abi_version() == _lib.ABI_VERSION >= 2       # bumped with every incompatible descriptor / workspace / plane-format change
    assert L.frido_sizeof_op() == C.sizeof(_lib.FridoOp)
    for kname, sname in _lib.KIND_STRUCT.items():
        assert L.frido_sizeof_desc(_lib.OP_KINDS[kname]) == C.sizeof(_lib.STRUCTS[sname]), sname


def test_shipped_libraries_are_not_profiling_or_experiment_builds():
    """r06: -DCG_PROF / -DIG_PROF builds (tools/cg_prof.py, tools/igemm_prof.py) export a reader for their per-wave cycle sums and carry an
    s_memtime + lgkmcnt(0) in every k-step; the loop-form experiments (CG_MIDBAR, CG_PINGPONG, FRIDO_MIDBAR) are off in what ships.  Both
    libraries in the tree must be the plain build: no profiling entry point, and the macros' defaults in the sources say 0."""
    import os
    import re
    for key in ("f16", "bf16"):
        L = C.CDLL(_lib.LIB_PATHS[key])
        for name in ("frido_cg_prof_read", "frido_ig_prof_read"):
            assert not hasattr(L, name), f"{_lib.LIB_PATHS[key]} is a profiling build ({name})"
    src = os.path.join(os.path.dirname(_lib.__file__), "csrc")
    text = open(os.path.join(src, "convgn.hip")).read() + open(os.path.join(src, "igemm.hip")).read() + open(os.path.join(src, "igemm_shared.h")).read()
    for macro in ("CG_PROF", "IG_PROF", "CG_MIDBAR", "CG_PINGPONG", "FRIDO_MIDBAR", "CG_ABLATE", "FRIDO_ABLATE"):
        m = re.search(r"#ifndef %s\n#define %s (\d+)" % (macro, macro), text)
        assert m and m.group(1) == "0", macro
    assert re.search(r"#ifndef FRIDO_SILU_DIV\n#define FRIDO_SILU_DIV 0", open(os.path.join(src, "common.h")).read())


def test_bad_descriptors_are_rejected_without_touching_a_device():
    L = _lib.lib()
    kind, st = _lib.make_op("FRIDO_OP_GEMM", M=16, N=16, K=20, batch=1, nsplit=1)     # K not a multiple of 32
    assert L.frido_gemm(C.addressof(st), None) == -1
    assert b"multiple of 32" in L.frido_last_error()
    kind, st = _lib.make_op("FRIDO_OP_SOFTMAX", rows=4, N=5000, Npad=5024)
    assert L.frido_softmax(C.addressof(st), None) == -1
    # the LayerNorm output of the short-key attention kernel needs workgroups that own whole rows (>= 256 of them)
    kind, st = _lib.make_op("FRIDO_OP_ATTN_SMALL", Q=64, K=64, VT=64, out_act=64, ln_op=64, ln_w=64, ln_b=64, B=2, Nq=64, Nk=26, d=384, dv=384,
                            ldq=384, ldk=384, ldvt=32, ld_act=384, ldr=384, ld_ln=384, nsplit=2)
    assert L.frido_attn_small(C.addressof(st), None) == -1
    assert b"ln_op" in L.frido_last_error()


def test_split_k_workspace_sizing_is_a_host_function():
    """frido_gemm_workspace_bytes (include/frido_hip.h): 0 without split-K; 64 KiB of arrival tickets + splitk * M * N floats for the
    two-kernel reduction; whole tiles (M, N padded to the family's largest tile edges) for the in-kernel reduction."""
    import ctypes as C
    from frido_amd import _lib
    L = _lib.lib()
    st = _lib.STRUCTS["FridoGemm"]()
    st.M, st.N, st.K, st.splitk = 1000, 960, 8640, 1
    assert L.frido_gemm_workspace_bytes(C.addressof(st)) == 0
    st.splitk = 8
    assert L.frido_gemm_workspace_bytes(C.addressof(st)) == 65536 + 8 * 1000 * 960 * 4
    st.sk_mode = 1
    assert L.frido_gemm_workspace_bytes(C.addressof(st)) == 65536 + 8 * 1024 * 1152 * 4


def test_operand_plane_format_and_host_packer_agree():
    """Two-plane operands: the host packer (engine.pack_matrix) splits in the format the library multiplies in
    (frido_x3_plane_format: fp16 pairs = 22 mantissa bits, bf16 pairs = 16); one-plane operands are bf16."""
    from frido_amd import _lib
    from frido_amd.engine import pack_matrix, plane_dtype
    fmt = _lib.lib().frido_x3_plane_format()
    assert fmt in (0, 1) and plane_dtype(1) == torch.bfloat16 and plane_dtype(2) == (torch.float16 if fmt else torch.bfloat16)
    w = torch.randn(37, 50, generator=torch.Generator().manual_seed(5)) * 3
    two, one = pack_matrix(w, 2), pack_matrix(w, 1)
    assert two.K == 64 and float(two.to_f32()[:, 50:].abs().max()) == 0.0            # K zero-padded to 32
    rel = lambda o: float(((o.to_f32()[:, :50] - w).abs() / w.abs().clamp_min(0.25)).max())
    assert rel(two) < (2.0 ** -20 if fmt else 2.0 ** -15) and 2.0 ** -12 < rel(one) < 2.0 ** -7


def test_reference_targets_resolve_and_state_dict_layout():
    from frido_amd.models import instantiate_from_config
    import frido.models.diffusion.frido as F
    import ldm.models.diffusion.msldm as L
    import ldm.modules.diffusionmodules.openaimodel as O
    import frido.modules.diffusionmodules.pyunet as P
    assert L.MSLatentDiffusion is F.FridoDiffusion and O.UNetModel is P.PyUNetModel
    cfg = frido_cfg(UNET_SMALL, VQ_SMALL, BERT_SMALL)
    cfg["cond_stage_config"] = "__is_unconditional__"
    cfg["conditioning_key"] = "crossattn"
    m = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg))
    keys = set(m.state_dict().keys())
    g = golden("unet_small")
    assert {"model.diffusion_model." + k for k in g["keys"]} <= keys
    assert {"first_stage_model." + k for k in golden("vq_small")["keys"]} <= keys
    assert {"scale_factor", "betas", "alphas_cumprod", "model_ema.decay", "model_ema.num_updates"} <= keys
    ema = [k for k in keys if k.startswith("model_ema.diffusion_model")]
    assert len(ema) == len(g["keys"]) and all("." not in k[len("model_ema."):] for k in ema)     # ema.py:16-20 mangling
    assert m.scale_factor.shape == (2,) and m.num_timesteps == 1000
    assert m.model.diffusion_model.num_stage == 2 and m.embed_dim_list == [3, 3] and m.use_split_head
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})


def test_no_cpu_fallback_anywhere():
    from frido_amd._lib import FridoHipError
    from frido_amd.models import PyUNetModel, VQModelInterface
    from frido_amd.samplers import DDIMSampler
    u = PyUNetModel(**UNET_SMALL)
    with pytest.raises(FridoHipError):
        u(torch.zeros(1, 3, 16, 16), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 5, 64), stage=0)
    v = VQModelInterface(**VQ_SMALL, lossconfig=dict(target="taming.modules.losses.DummyLoss"))
    with pytest.raises(FridoHipError):
        v.decode(torch.zeros(1, 6, 16, 16))

    class M:
        num_timesteps = 1000
    with pytest.raises(FridoHipError):
        DDIMSampler(M()).sample(S=4, batch_size=1, shape=(6, 16, 16), conditioning=torch.zeros(1, 5, 64), verbose=False)


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "frido_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_full_width_holder_has_reference_parameter_count():
    from helpers import unet_holder
    assert sum(p.numel() for p in unet_holder(UNET_FULL).parameters()) == int(golden("unet_full")["nparam"]) == 511669446   # the reference module (SURVEY App. A rounds to 511.67 M)


def test_checkpoint_ingestion_roundtrip(tmp_path):
    """A reference-style checkpoint ({'state_dict': ...} with model.*, model_ema.*, first_stage_model.*, cond_stage_model.*,
    scale_factor) loads through init_from_ckpt / load_state_dict(strict=False) (scripts/sample_diffusion.py:452-469)."""
    from frido_amd.models import instantiate_from_config
    from frido_amd.synth import fill_module
    cfg = frido_cfg(UNET_SMALL, VQ_SMALL, BERT_SMALL)
    src = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg))
    fill_module(src.model, "model.")
    fill_module(src.first_stage_model, "first_stage_model.")
    fill_module(src.cond_stage_model, "cond_stage_model.")
    src.model_ema.copy_to  # noqa: B018  (exists)
    for k, v in src.model_ema.named_buffers():
        if v.dtype.is_floating_point and v.dim() > 0:
            v.fill_(0.5)
    src.scale_factor.copy_(torch.tensor([0.7, 1.3]))
    path = tmp_path / "model.ckpt"
    sd = dict(src.state_dict())
    sd["some.unknown.key"] = torch.zeros(1)               # strict=False tolerates extras
    torch.save({"state_dict": sd, "global_step": 1}, path)
    cfg2 = frido_cfg(UNET_SMALL, VQ_SMALL, BERT_SMALL)
    cfg2["ckpt_path"] = str(path)
    dst = instantiate_from_config(dict(target="frido.models.diffusion.frido.FridoDiffusion", params=cfg2))
    for k, v in src.state_dict().items():
        d = dst.state_dict()[k]
        assert torch.equal(d, v), (k, int(torch.isnan(v.float()).sum()), int(torch.isnan(d.float()).sum()), int((d != v).sum()))
    w0 = dst.model.diffusion_model.time_embed[0].weight.clone()
    with dst.ema_scope():                                  # EMA weights swapped in (frido.py:181-194) ...
        assert float(dst.model.diffusion_model.time_embed[0].weight.mean()) == 0.5
    assert torch.equal(dst.model.diffusion_model.time_embed[0].weight, w0)    # ... and restored
    assert torch.equal(dst.scale_factor, torch.tensor([0.7, 1.3]))


def test_sampling_script_import_surface():
    """Every name scripts/sample_diffusion.py:16-21 imports from the model packages resolves here (the CLI's DataModule /
    PNG writer are host I/O and stay the reference's own)."""
    from frido.util import log_txt_as_img, exists, default, ismap, isimage, mean_flat, count_params  # noqa: F401
    from frido.util import instantiate_from_config_main as instantiate_from_config  # noqa: F401
    from frido.models.diffusion.ddim import DDIMSampler  # noqa: F401
    from frido.models.diffusion.plms import PLMSSampler  # noqa: F401
    from taming.data.utils import custom_collate
    assert exists(0) and not exists(None) and default(None, lambda: 3) == 3 and default(2, 5) == 2
    assert ismap(torch.zeros(1, 4, 2, 2)) and not ismap(torch.zeros(1, 3, 2, 2)) and isimage(torch.zeros(1, 3, 2, 2))
    assert not isimage(np.zeros((1, 3, 2, 2))) and mean_flat(torch.ones(2, 3, 4)).tolist() == [1.0, 1.0]
    assert count_params(torch.nn.Linear(3, 2)) == 8
    img = log_txt_as_img((64, 32), ["a caption", ["x", "y"]], size=8)
    assert img.shape == (2, 3, 32, 64) and float(img.max()) <= 1.0 and float(img.min()) >= -1.0
    # collate: tensors stack, strings stay lists, dicts recurse, ragged per-sample Annotation lists pass through
    import collections
    Annotation = collections.namedtuple("Annotation", "bbox category_no")
    batch = [{"image": torch.zeros(4, 4, 3), "objects_bbox": torch.arange(26), "file_name": "a.png",
              "annotations": [Annotation((0, 0, 1, 1), 3)]},
             {"image": torch.ones(4, 4, 3), "objects_bbox": torch.arange(26) + 1, "file_name": "b.png",
              "annotations": [Annotation((0, 0, 1, 1), 5), Annotation((0, 0, .5, .5), 7)]}]
    out = custom_collate(batch)
    assert out["image"].shape == (2, 4, 4, 3) and out["objects_bbox"].shape == (2, 26) and out["file_name"] == ["a.png", "b.png"]
    assert len(out["annotations"]) == 2 and len(out["annotations"][1]) == 2
    assert custom_collate([1, 2]).tolist() == [1, 2] and custom_collate([np.ones(3), np.zeros(3)]).shape == (2, 3)


def test_caller_surface_of_frido_diffusion():
    """get_img_ids (frido.py:818), q_sample (frido.py:302-320), the CLIP cond-stage target of the t2i YAML."""
    from frido_amd.models import instantiate_from_config
    cfg = frido_cfg(UNET_SMALL, VQ_SMALL, BERT_SMALL)
    cfg["cond_stage_config"] = dict(target="frido.modules.encoders.modules.FrozenCLIPTextEmbedder")
    cfg["cond_stage_trainable"], cfg["cond_stage_key"] = False, "caption"
    m = instantiate_from_config(dict(target="ldm.models.diffusion.msldm.MSLatentDiffusion", params=cfg))
    assert m.cond_stage_model.use_tknz_fn and m.get_img_ids({"file_name": ["x"]}) == ["x"]
    with pytest.raises(NotImplementedError, match="not reachable"):      # strings need CLIP's BPE vocabulary; token ids run on the GPU
        m.get_learned_conditioning(["a photo"])
    keys = set(m.cond_stage_model.state_dict())       # OpenAI CLIP's own key names: a reference checkpoint's cond_stage_model.model.* load
    assert {"model.token_embedding.weight", "model.positional_embedding", "model.text_projection", "model.ln_final.bias",
            "model.transformer.resblocks.11.attn.in_proj_weight", "model.transformer.resblocks.0.mlp.c_fc.bias"} <= keys
    assert m.cond_stage_model.state_dict()["model.transformer.resblocks.3.attn.in_proj_weight"].shape == (3 * 768, 768)
    x0 = torch.randn(2, 6, 4, 4)
    t = torch.tensor([10, 900])
    nz = torch.randn_like(x0)
    a, s1 = m.sqrt_alphas_cumprod[t].view(2, 1, 1, 1), m.sqrt_one_minus_alphas_cumprod[t].view(2, 1, 1, 1)
    assert torch.equal(m.q_sample(x0, t, noise=nz), a * x0 + s1 * nz)
    q = m.q_sample(x0, t, ch_start=3, ch_end=5, noise=nz, mix_tau=0.25)
    assert torch.equal(q[:, 5:], nz[:, 5:]) and torch.equal(q[:, 3:5], (a * x0 + s1 * nz)[:, 3:5])
    assert torch.allclose(q[:, :3], 0.75 * x0[:, :3] + 0.25 * nz[:, :3])


def test_sampler_argument_validation_on_cpu():
    from frido_amd.samplers import DDIMSampler, PLMSSampler

    class M:
        num_timesteps = 1000
        alphas_cumprod = torch.from_numpy(schedules.ddpm_tables(schedules.make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0155))["alphas_cumprod"])
    with pytest.raises(ValueError, match="conditionings"):
        DDIMSampler(M()).sample(S=4, batch_size=2, shape=(6, 16, 16), conditioning=torch.zeros(1, 5, 64), verbose=False)
    with pytest.raises(ValueError):
        PLMSSampler(M()).make_schedule(ddim_num_steps=4, ddim_eta=0.5)


def test_every_shipped_yaml_model_tree_instantiates():
    """All 11 configs/frido/**/*.yaml `model:` trees of the reference (restated as data in tests/golden/shipped_model_cfgs.json by
    tests/golden/make_golden.py shipped_cfgs) go through instantiate_from_config unchanged: the stale ldm.* targets, the
    use_tokenizer: True cond stages (which now defer their tokenizer error to encode()) and the CLIP t2i config included.
    ckpt_path entries point at files of the reference's download script, so they are dropped like scripts/sample_diffusion.py -r
    would override them."""
    import json
    from frido_amd.models import instantiate_from_config, FridoDiffusion, PyUNetModel, VQModelInterface
    cfgs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "shipped_model_cfgs.json")))
    assert len(cfgs) == 11

    def strip(node):
        if isinstance(node, dict):
            return {k: strip(v) for k, v in node.items() if k not in ("ckpt_path",)}
        return node

    for name, tree in sorted(cfgs.items()):
        m = instantiate_from_config(strip(tree))
        assert isinstance(m, FridoDiffusion), name
        u = m.model.diffusion_model
        assert isinstance(u, PyUNetModel) and isinstance(m.first_stage_model, VQModelInterface), name
        assert u.num_stage == len(m.split_embed_dim_list if hasattr(m, "split_embed_dim_list") else [0]) or u.num_stage >= 1
        assert hasattr(m.cond_stage_model, "encode"), name


def test_lit_ema_shadows_only_trainable_parameters():
    """ema.py:16-20: a parameter with requires_grad == False has no shadow buffer (and no `model_ema.*` checkpoint key);
    copy_to leaves it alone."""
    import torch.nn as nn
    from frido_amd.models import LitEma
    m = nn.Sequential(nn.Linear(3, 3), nn.Linear(3, 2))
    for p in m[1].parameters():
        p.requires_grad = False
    ema = LitEma(m)
    assert set(ema.m_name2s_name) == {"0.weight", "0.bias"}
    assert {n for n, _ in ema.named_buffers()} == {"decay", "num_updates", "0weight", "0bias"}
    frozen = m[1].weight.clone()
    getattr(ema, "0weight").fill_(7.0)
    ema.copy_to(m)
    assert float(m[0].weight.mean()) == 7.0 and torch.equal(m[1].weight, frozen)
    with pytest.raises(ValueError):
        LitEma(m, decay=1.5)


def test_diffusion_wrapper_routes_every_conditioning_key_like_the_reference():
    """frido.py:1635-1654: None -> model(x, t); concat -> model(cat(x, c_concat)); crossattn -> context = cat(c_crossattn, 1);
    hybrid -> both; adm -> y = c_crossattn[0].  (r05: the keys other than 'crossattn' used to raise in the wrapper.)"""
    import torch.nn as nn
    from frido_amd.models import DiffusionWrapper, PyUNetModel

    class Rec(nn.Module):
        def forward(self, x, t, context=None, y=None, stage=None):
            self.got = dict(x=x, t=t, context=context, y=y, stage=stage)
            return x

    x, t = torch.randn(2, 6, 4, 4), torch.tensor([5, 9])
    cc, c2 = [torch.randn(2, 3, 4, 4)], [torch.randn(2, 5, 8), torch.randn(2, 2, 8)]
    got = {}
    for key in (None, "concat", "crossattn", "hybrid", "adm"):
        w = DiffusionWrapper.__new__(DiffusionWrapper)
        nn.Module.__init__(w)
        w.diffusion_model, w.conditioning_key = Rec(), key
        w(x, t, c_concat=cc, c_crossattn=c2, stage=1)
        got[key] = w.diffusion_model.got
        assert got[key]["stage"] == 1 and torch.equal(got[key]["t"], t)
    assert torch.equal(got[None]["x"], x) and got[None]["context"] is None and got[None]["y"] is None
    assert torch.equal(got["concat"]["x"], torch.cat([x] + cc, 1)) and got["concat"]["context"] is None
    assert torch.equal(got["crossattn"]["x"], x) and torch.equal(got["crossattn"]["context"], torch.cat(c2, 1))
    assert torch.equal(got["hybrid"]["x"], torch.cat([x] + cc, 1)) and torch.equal(got["hybrid"]["context"], torch.cat(c2, 1))
    assert torch.equal(got["adm"]["x"], x) and got["adm"]["y"] is c2[0] and got["adm"]["context"] is None
    # the denoiser itself says what it was not built for (clear NotImplementedError, not an AttributeError three frames down)
    from golden_cfg import UNET_SMALL
    u = PyUNetModel(**UNET_SMALL)
    with pytest.raises(NotImplementedError, match="class-conditional"):
        u(x, t, context=c2[0], y=torch.tensor([1, 2]), stage=0)
    with pytest.raises(NotImplementedError, match="without a context"):
        u(x, t, stage=0)


def test_plane_format_selection_and_pool_hold_host_logic():
    """r05 host logic without a GPU: the precision keyword -> (nsplit, plane format) mapping, the thread-local library routing of
    _lib.use_planes (nesting restores, unknown formats are rejected, both builds export every declared symbol), and Pool.hold -- the fix
    of the deferred-reduction aliasing bug: a released buffer that an op about to be emitted still reads is taken out of circulation."""
    import threading
    from frido_amd import _lib, config
    from frido_amd.engine import Pool
    assert (config.nsplit("bf16x3"), config.planes("bf16x3")) == (2, "f16")
    assert (config.nsplit("bf16x3_bf16"), config.planes("bf16x3_bf16")) == (2, "bf16")
    assert (config.nsplit("bf16"), config.planes("bf16")) == (1, "f16")
    with pytest.raises(ValueError):
        config.nsplit("fp64")
    assert _lib.active_planes() == "f16"
    with _lib.use_planes("bf16"):
        assert _lib.active_planes() == "bf16"
        with _lib.use_planes("f16"):
            assert _lib.active_planes() == "f16"
        assert _lib.active_planes() == "bf16"
        seen = []
        th = threading.Thread(target=lambda: seen.append(_lib.active_planes()))      # another thread keeps the default
        th.start(); th.join()
        assert seen == ["f16"]
    assert _lib.active_planes() == "f16"
    with pytest.raises(ValueError):
        with _lib.use_planes("fp8"):
            pass
    for planes, fmt in (("f16", 1), ("bf16", 0)):          # both builds of the same sources: same ABI, different element format
        L = _lib.lib(planes)
        assert L.frido_x3_plane_format() == fmt and L.frido_abi_version() == _lib.ABI_VERSION
        assert not [s for s in _lib.declared_symbols() if not hasattr(L, s)]
    pool = Pool(torch.device("cpu"))
    a, b = pool.alloc(1000), pool.alloc(1000)
    pa = a.data_ptr()
    pool.release(a)
    assert pool.hold(b.data_ptr()) is None                  # b is in use, not free: nothing to hold
    held = pool.hold(pa)
    assert held is a and pool.alloc(1000) is not a          # while held, the same size class hands out a fresh buffer
    pool.release(held)
    assert pool.alloc(1000) is a


def test_auto_plane_selection_host_logic(monkeypatch):
    """r06 (frido_amd/autoplanes.py) without a GPU: a module on the DEFAULT precision keyword whose run saturates an fp16 plane is moved to
    the bf16-pair build and the call repeated with the host noise REWOUND (torch's generator state, a recorded tape); the discarded
    attempt's SATURATED bit does not reach the sticky word, other bits do; pinned keywords and FRIDO_AUTO_PLANES=0 run once."""
    import types
    import warnings
    from frido_amd import _lib, autoplanes, config

    class M:
        def __init__(self, precision=None):
            self.precision, self.invalidated = precision, 0
        planes = property(lambda self: config.planes(self.precision))

        def invalidate(self):
            self.invalidated += 1

    script, polled = [], []

    def fake_poll(planes=None, clear=True, keep=True):
        polled.append(planes)
        w = script.pop(0)
        if clear and keep:
            _lib._sticky[0] |= w
        return w
    monkeypatch.setattr(_lib, "status_poll", fake_poll)
    monkeypatch.setattr(_lib, "lib", lambda planes=None: types.SimpleNamespace(frido_status_poll=True))
    monkeypatch.setattr(_lib, "_sticky", [0])
    assert config.auto_planes(None) and config.auto_planes("bf16x3") and not config.auto_planes("bf16x3_f16") and not config.auto_planes("bf16x3_bf16")
    assert (config.nsplit("bf16x3_f16"), config.planes("bf16x3_f16")) == (2, "f16")
    # (1) clean run: one call; an earlier bit and a NONFINITE bit of the run both stay visible
    m, calls = M(), []
    script[:] = [_lib.STATUS_NONFINITE, 0]
    assert autoplanes.run(m, lambda n: calls.append(n) or "out", "t", noise="philox") == "out"
    assert calls == ["philox"] and m.precision is None and _lib._sticky[0] == _lib.STATUS_NONFINITE and not script
    # (2) saturated run: repeated on the bf16 pairs with the SAME host noise (generator and tape), one warning, SATURATED not kept
    _lib._sticky[0] = 0
    m, draws = M("bf16x3"), []
    tape = iter([torch.full((2,), 1.0), torch.full((2,), 2.0), torch.full((2,), 99.0)])
    script[:] = [0, _lib.STATUS_SATURATED | _lib.STATUS_NONFINITE, 0]

    def call(noise):
        draws.append((torch.randn(3), noise((2,)).clone(), noise((2,)).clone(), m.planes))
        return len(draws)
    torch.manual_seed(5)
    with pytest.warns(_lib.FridoNumericsWarning, match="bf16-pair planes"):
        assert autoplanes.run(m, call, "t", noise=lambda shape: next(tape)) == 2
    assert m.precision == "bf16x3_bf16" and m.planes == "bf16" and m.invalidated == 1 and polled[-3:] == ["f16", "f16", "bf16"]
    assert draws[0][3] == "f16" and draws[1][3] == "bf16"
    assert all(torch.equal(draws[0][i], draws[1][i]) for i in range(3)) and float(draws[1][2][0]) == 2.0      # rewound, not re-drawn
    assert _lib._sticky[0] == _lib.STATUS_NONFINITE
    # (3) pinned formats / the switch off: exactly one call, no polls
    for mod, auto in ((M("bf16x3_f16"), True), (M("bf16x3_bf16"), True), (M("bf16"), True), (M(), False)):
        monkeypatch.setattr(config, "AUTO_PLANES", auto)
        n0, script[:] = len(polled), []
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            assert autoplanes.run(mod, lambda n: "x", "t") == "x"
        assert len(polled) == n0 and mod.invalidated == 0
