"""Parameter holders: nn.Module trees whose state_dict keys and shapes are exactly the reference's
(SURVEY.md Appendix A/B), so the reference's checkpoints (`model.diffusion_model.*`,
`first_stage_model.*`, `model_ema.*`) load with load_state_dict.  They hold weights only — there
is NO torch forward here: the compute lives in the HIP engine (frido_amd/engine.py), which packs
these parameters into its own device layouts.  Parameters are created uninitialised
(torch.empty) — real values come from a checkpoint or from frido_amd.synth.
"""
import torch
import torch.nn as nn

from .arch import unet_arch, decoder_arch, encoder_arch


def _p(*shape):
    # requires_grad like every nn layer of the reference: LitEma shadows exactly the parameters that require grad (ema.py:16-20),
    # so the denoiser's holders must look trainable for the `model_ema.*` keys to exist; the first stage is frozen by
    # FridoDiffusion.instantiate_first_stage (frido.py:617-623).  Nothing here is ever differentiated: the HIP path runs under no_grad.
    return nn.Parameter(torch.empty(*shape))


class Conv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = _p(cout, cin, k, k)
        self.bias = _p(cout)


class ConvT(nn.Module):   # nn.ConvTranspose2d layout: (cin, cout, k, k)
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = _p(cin, cout, k, k)
        self.bias = _p(cout)


class Lin(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = _p(cout, cin)
        if bias:
            self.bias = _p(cout)


class Affine(nn.Module):   # GroupNorm / LayerNorm scale+shift
    def __init__(self, c):
        super().__init__()
        self.weight = _p(c)
        self.bias = _p(c)


class Emb(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = _p(n, d)


class Nop(nn.Module):
    pass


def seq(*mods):
    return nn.Sequential(*mods)


class Spade(nn.Module):   # spade_norm.py:26-42
    def __init__(self, c, cond_c, hidden=128):
        super().__init__()
        self.param_free_norm = Affine(c)
        self.mlp_shared = seq(Conv(cond_c, hidden, 3), Nop())
        self.mlp_gamma = Conv(hidden, c, 3)
        self.mlp_beta = Conv(hidden, c, 3)


def _norm(c, cond_c, spade):
    return Spade(c, cond_c) if spade else Affine(c)


class ResBlockP(nn.Module):   # pyunet.py:208-248
    def __init__(self, cin, cout, emb, cond_c, spade):
        super().__init__()
        self.in_layers = seq(_norm(cin, cond_c, spade), Nop(), Conv(cin, cout, 3))
        self.emb_layers = seq(Nop(), Lin(emb, cout))
        self.out_layers = seq(_norm(cout, cond_c, spade), Nop(), Nop(), Conv(cout, cout, 3))
        self.skip_connection = Nop() if cin == cout else Conv(cin, cout, 1)


class CrossAttnP(nn.Module):  # attention.py:153-168
    def __init__(self, dim, ctx):
        super().__init__()
        self.to_q = Lin(dim, dim, bias=False)
        self.to_k = Lin(ctx, dim, bias=False)
        self.to_v = Lin(ctx, dim, bias=False)
        self.to_out = seq(Lin(dim, dim), Nop())


class GegluP(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = Lin(dim, inner * 2)


class FFP(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = seq(GegluP(dim, dim * 4), Nop(), Lin(dim * 4, dim))


class TBlockP(nn.Module):     # attention.py:197-205
    def __init__(self, dim, ctx):
        super().__init__()
        self.attn1 = CrossAttnP(dim, dim)
        self.ff = FFP(dim)
        self.attn2 = CrossAttnP(dim, ctx)
        self.norm1 = Affine(dim)
        self.norm2 = Affine(dim)
        self.norm3 = Affine(dim)


class SpatialTransformerP(nn.Module):   # attention.py:250-280
    def __init__(self, c, cond_c, ctx, spade, depth=1):
        super().__init__()
        self.norm = _norm(c, cond_c, spade)
        self.proj_in = Conv(c, c, 1)
        self.transformer_blocks = nn.ModuleList([TBlockP(c, ctx) for _ in range(depth)])
        self.proj_out = Conv(c, c, 1)


class DownP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.op = Conv(c, c, 3)


class UpP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv(c, c, 3)


def _make(b, a):
    if b.kind == "res":
        return ResBlockP(b.cin, b.cout, a.time_embed_dim, a.model_channels, a.use_spade)
    if b.kind == "st":
        return SpatialTransformerP(b.cin, a.model_channels, a.context_dim, a.use_spade, a.transformer_depth)
    if b.kind == "down":
        return DownP(b.cin)
    if b.kind == "up":
        return UpP(b.cin)
    raise ValueError(b.kind)


def build_unet_params(root: nn.Module, cfg):
    """Populate `root` with the parameter tree of pyunet.py:560-803."""
    a = unet_arch(cfg)
    mc, te = a.model_channels, a.time_embed_dim
    root.time_embed = seq(Lin(mc, te), Nop(), Lin(te, te))
    if a.num_stage > 1:
        root.stage_emb = Emb(a.num_stage, te)
    if a.use_split_head:
        n = len(a.splits)
        if a.use_spade:
            root.pre_input_cond_blocks = nn.ModuleList(
                [seq(Conv(sum(a.splits[:i + 1]), mc, 3)) for i in range(n - 1)])
            root.pre_input_blocks = nn.ModuleList([seq(Conv(a.splits[i], mc, 3)) for i in range(n)])
        else:
            root.pre_input_blocks = nn.ModuleList([seq(Conv(sum(a.splits[:i + 1]), mc, 3)) for i in range(n)])
        blocks = []
    else:
        blocks = [seq(Conv(a.in_channels, mc, 3))]
    for blk in a.input_blocks:
        blocks.append(seq(*[_make(b, a) for b in blk]))
    root.input_blocks = nn.ModuleList(blocks)
    root.middle_block = seq(*[_make(b, a) for b in a.middle])
    root.output_blocks = nn.ModuleList([seq(*[_make(b, a) for b in blk]) for blk in a.output_blocks])
    if a.use_split_head:
        root.out = nn.ModuleList([seq(Affine(mc), Nop(), Conv(mc, a.splits[i], 3)) for i in range(len(a.splits))])
    else:
        root.out = seq(Affine(mc), Nop(), Conv(mc, cfg["out_channels"], 3))
    return a


# ---- MS-VQGAN ------------------------------------------------------------------------------------
class VResP(nn.Module):       # taming model.py:78-112 (temb_channels == 0)
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = Affine(cin)
        self.conv1 = Conv(cin, cout, 3)
        self.norm2 = Affine(cout)
        self.conv2 = Conv(cout, cout, 3)
        if cin != cout:
            self.nin_shortcut = Conv(cin, cout, 1)


class VAttnP(nn.Module):      # taming model.py:140-165
    def __init__(self, c):
        super().__init__()
        self.norm = Affine(c)
        self.q = Conv(c, c, 1)
        self.k = Conv(c, c, 1)
        self.v = Conv(c, c, 1)
        self.proj_out = Conv(c, c, 1)


class VSampleP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv(c, c, 3)


def _set(root, dotted, mod):
    """Create intermediate Nop()/ModuleList containers so that root.<dotted> == mod."""
    parts = dotted.split(".")
    cur = root
    for i, p in enumerate(parts[:-1]):
        nxt = parts[i + 1]
        if p.isdigit():
            idx = int(p)
            while len(cur) <= idx:
                cur.append(nn.ModuleList() if nxt.isdigit() else Nop())
            cur = cur[idx]
        else:
            if not hasattr(cur, p):
                setattr(cur, p, nn.ModuleList() if nxt.isdigit() else Nop())
            cur = getattr(cur, p)
    last = parts[-1]
    if last.isdigit():
        idx = int(last)
        while len(cur) <= idx:
            cur.append(Nop())
        cur[idx] = mod
    else:
        setattr(cur, last, mod)


def _vmake(b):
    return {"res": lambda: VResP(b.cin, b.cout), "attn": lambda: VAttnP(b.cin),
            "up": lambda: VSampleP(b.cin), "down": lambda: VSampleP(b.cin)}[b.kind]()


def build_decoder_params(root, dd, prefix="decoder"):
    a = decoder_arch(dd, prefix)
    _set(root, f"{prefix}.conv_in", Conv(a.z_channels, a.block_in, 3))
    for b in a.body:
        _set(root, b.prefix, _vmake(b))
    _set(root, f"{prefix}.norm_out", Affine(a.last_ch))
    _set(root, f"{prefix}.conv_out", Conv(a.last_ch, a.out_ch, 3))
    return a


def build_encoder_params(root, ed, prefix="encoder"):
    a = encoder_arch(ed, prefix)
    _set(root, f"{prefix}.conv_in", Conv(a.in_channels, a.ch, 3))
    for blocks in a.down:
        for b in blocks:
            _set(root, b.prefix, _vmake(b))
    for i in range(a.multiscale):
        for b in a.heads[i]:
            _set(root, b.prefix, _vmake(b))
        _set(root, f"{prefix}.norm_out_ms.{i}", Affine(a.head_ch[i]))
        _set(root, f"{prefix}.conv_out_ms.{i}", Conv(a.head_ch[i], a.z_channels[i], 3))
    return a


def shared_decoder_cfg(embed_dim, ii):
    """The hard-coded shared decoder of msvqgan.py:86-87 (ch=128, one level, attention only in mid)."""
    return dict(double_z=False, z_channels=sum(embed_dim[:ii + 2]), resolution=256, in_channels=None,
                out_ch=embed_dim[0], ch=128, ch_mult=[1], num_res_blocks=2,
                attn_resolutions=[2, 4, 8, 16, 32, 64], dropout=0.0)


def build_msvqgan_params(root, edconfig, ddconfig, n_embed, embed_dim):
    """msvqgan.py:38-87."""
    build_encoder_params(root, edconfig)
    build_decoder_params(root, ddconfig)
    n = len(n_embed)
    root.ms_quantize = nn.ModuleList()
    root.ms_quant_conv = nn.ModuleList()
    for i in range(n):
        q = Nop()
        q.embedding = Emb(n_embed[i], embed_dim[i])
        root.ms_quantize.append(q)
        zc = edconfig["z_channels"][i] * (2 if edconfig.get("double_z") else 1)
        root.ms_quant_conv.append(Conv(zc, embed_dim[i], 1))
    root.post_quant_conv = Conv(sum(embed_dim), ddconfig["z_channels"], 1)
    root.upsample = nn.ModuleList()
    root.shared_decoder = nn.ModuleList()
    root.shared_post_quant_conv = nn.ModuleList()
    for i in range(n - 1):
        root.upsample.append(ConvT(embed_dim[0], embed_dim[0], 4))
        root.shared_post_quant_conv.append(Conv(embed_dim[0], edconfig["z_channels"][0], 1))
        holder = Nop()
        build_decoder_params(holder, shared_decoder_cfg(embed_dim, i), prefix="d")
        root.shared_decoder.append(holder.d)


# ---- cond stage: BERTEmbedder = x-transformers TransformerWrapper(Encoder) (encoders/modules.py:85-97) ------------
def build_bert_params(root, n_embed, n_layer, vocab_size, max_seq_len, heads=8, dim_head=64):
    """Parameter tree of frido/modules/x_transformer.py:548-597 (TransformerWrapper) + 370-479 (Encoder)."""
    tr = Nop()
    tr.token_emb = Emb(vocab_size, n_embed)
    pe = Nop()
    pe.emb = Emb(max_seq_len, n_embed)
    tr.pos_emb = pe
    inner = heads * dim_head
    layers = []
    for _ in range(n_layer):
        attn = Nop()
        attn.to_q, attn.to_k, attn.to_v = Lin(n_embed, inner, False), Lin(n_embed, inner, False), Lin(n_embed, inner, False)
        attn.to_out = Lin(inner, n_embed)
        layers.append(nn.ModuleList([Affine(n_embed), attn, Nop()]))
        ff = Nop()
        ff.net = seq(seq(Lin(n_embed, 4 * n_embed), Nop()), Nop(), Lin(4 * n_embed, n_embed))
        layers.append(nn.ModuleList([Affine(n_embed), ff, Nop()]))
    al = Nop()
    al.layers = nn.ModuleList(layers)
    tr.attn_layers = al
    tr.norm = Affine(n_embed)
    tr.to_logits = Lin(n_embed, vocab_size)
    root.transformer = tr


# ---- cond stage: FrozenCLIPTextEmbedder.model = OpenAI CLIP (text tower only; clip/model.py CLIP.__init__) ------------------
CLIP_TEXT_ARCH = {      # version -> (embed_dim, context_length, vocab_size, transformer_width, transformer_heads, transformer_layers)
    "ViT-L/14": (768, 77, 49408, 768, 12, 12),
    "ViT-B/32": (512, 77, 49408, 512, 8, 12),
    "ViT-B/16": (512, 77, 49408, 512, 8, 12),
}


def build_clip_text_params(root, embed_dim, context_length, vocab_size, width, heads, layers):
    """Parameter tree of the text side of clip.model.CLIP under `root.model` (state_dict keys `model.token_embedding.weight`,
    `model.positional_embedding`, `model.transformer.resblocks.N.{ln_1,attn.in_proj_*,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj}`,
    `model.ln_final`, `model.text_projection`, `model.logit_scale`): a reference checkpoint's `cond_stage_model.model.*` keys
    load into it; its `model.visual.*` keys are simply unexpected keys of a strict=False load."""
    m = Nop()
    m.token_embedding = Emb(vocab_size, width)
    m.positional_embedding = _p(context_length, width)
    blocks = []
    for _ in range(layers):
        blk = Nop()
        blk.ln_1, blk.ln_2 = Affine(width), Affine(width)
        attn = Nop()
        attn.in_proj_weight, attn.in_proj_bias = _p(3 * width, width), _p(3 * width)
        attn.out_proj = Lin(width, width)
        blk.attn = attn
        mlp = Nop()
        mlp.c_fc, mlp.c_proj = Lin(width, 4 * width), Lin(4 * width, width)
        blk.mlp = mlp
        blocks.append(blk)
    tr = Nop()
    tr.resblocks = seq(*blocks)
    m.transformer = tr
    m.ln_final = Affine(width)
    m.text_projection = _p(width, embed_dim)
    m.logit_scale = nn.Parameter(torch.empty(()), requires_grad=False)
    root.model = m
