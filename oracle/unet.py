"""ORACLE (test infrastructure, never shipped on the product path): CPU restatement, in plain
PyTorch fp32, of the reference denoiser forward.  Pinned against tests/golden/unet_*.npz, which
are captured from the reference's own PyUNetModel (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Functional style: `sd` is a flat {state_dict key -> torch.Tensor} with the reference's key names
under `prefix` ('model.diffusion_model.' in a full checkpoint).
"""
import math

import torch
import torch.nn.functional as F

from .walk import unet_blocks


def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171 — [cos | sin] of t * exp(-ln(max_period) * k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class _P:
    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd


def _conv(p, name, x, stride=1, padding=1):
    return F.conv2d(x, p(name + ".weight"), p(name + ".bias"), stride=stride, padding=padding)


def _norm(p, name, x, cond, eps):
    """GroupNorm32 (util.py:214-216, eps 1e-5) / Normalize (attention.py:76-77, eps 1e-6), optionally
    wrapped in SPADE (spade_norm.py:44-60): norm(x) * (1 + gamma) + beta with gamma/beta convs of the
    nearest-resized conditioning feature map."""
    if p.has(name + ".param_free_norm.weight"):
        n = F.group_norm(x.float(), 32, p(name + ".param_free_norm.weight"), p(name + ".param_free_norm.bias"), eps)
        if cond is None:
            return n
        c = F.interpolate(cond, size=x.shape[2:], mode="nearest")
        a = F.relu(_conv(p, name + ".mlp_shared.0", c))
        return n * (1 + _conv(p, name + ".mlp_gamma", a)) + _conv(p, name + ".mlp_beta", a)
    return F.group_norm(x.float(), 32, p(name + ".weight"), p(name + ".bias"), eps)


def res_block(p, b, x, emb, cond):
    """pyunet.py:262-300 (no up/down, no scale-shift)."""
    pre = b.prefix
    h = _norm(p, pre + ".in_layers.0", x, cond, 1e-5)
    h = _conv(p, pre + ".in_layers.2", F.silu(h))
    e = F.linear(F.silu(emb), p(pre + ".emb_layers.1.weight"), p(pre + ".emb_layers.1.bias"))
    h = h + e[:, :, None, None]
    h = _norm(p, pre + ".out_layers.0", h, cond, 1e-5)
    h = _conv(p, pre + ".out_layers.3", F.silu(h))
    if p.has(pre + ".skip_connection.weight"):
        x = _conv(p, pre + ".skip_connection", x, padding=0)
    return x + h


def cross_attention(p, pre, x, context):
    """attention.py:170-193 with heads == 1 (pyunet.py:638-640): scale = dim ** -0.5."""
    q = F.linear(x, p(pre + ".to_q.weight"))
    ctx = x if context is None else context
    k = F.linear(ctx, p(pre + ".to_k.weight"))
    v = F.linear(ctx, p(pre + ".to_v.weight"))
    sim = torch.einsum("bid,bjd->bij", q, k) * (q.shape[-1] ** -0.5)
    out = torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), v)
    return F.linear(out, p(pre + ".to_out.0.weight"), p(pre + ".to_out.0.bias"))


def spatial_transformer(p, b, x, context, cond, depth=1):
    """attention.py:289-326 + BasicTransformerBlock._forward 222-227 + GEGLU 37-44."""
    pre = b.prefix
    B, C, H, W = x.shape
    h = _norm(p, pre + ".norm", x, cond, 1e-6)
    h = _conv(p, pre + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    for d in range(depth):      # attention.py:274-277,321-322: `depth` BasicTransformerBlocks in sequence
        t = f"{pre}.transformer_blocks.{d}"
        ln = lambda n, z: F.layer_norm(z, (C,), p(f"{t}.{n}.weight"), p(f"{t}.{n}.bias"), 1e-5)
        h = cross_attention(p, t + ".attn1", ln("norm1", h), None) + h
        h = cross_attention(p, t + ".attn2", ln("norm2", h), context) + h
        g = F.linear(ln("norm3", h), p(t + ".ff.net.0.proj.weight"), p(t + ".ff.net.0.proj.bias"))
        a, gate = g.chunk(2, dim=-1)
        h = F.linear(a * F.gelu(gate), p(t + ".ff.net.2.weight"), p(t + ".ff.net.2.bias")) + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _conv(p, pre + ".proj_out", h, padding=0) + x


def _run(p, blocks, h, emb, context, cond, depth=1):
    for b in blocks:
        if b.kind == "res":
            h = res_block(p, b, h, emb, cond)
        elif b.kind == "st":
            h = spatial_transformer(p, b, h, context, cond, depth)
        elif b.kind == "down":   # pyunet.py:152-156: conv3x3 stride 2 padding 1
            h = _conv(p, b.prefix + ".op", h, stride=2, padding=1)
        elif b.kind == "up":     # pyunet.py:119-121: nearest x2 then conv3x3
            h = _conv(p, b.prefix + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"))
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, t, context, stage, prefix="model.diffusion_model.", taps=None):
    """pyunet.py:867-950.  x: (B, sum(splits[:stage+1]), H, W); returns eps (B, splits[stage], H, W)."""
    # (r06) the block walk comes from the checkpoint's own keys (oracle/walk.py), the scalar options straight from the config -- nothing
    # under frido_amd/ is consulted
    input_blocks, middle, output_blocks, depth = unet_blocks(sd, prefix)
    num_stage, splits = cfg.get("num_stage", 1), list(cfg.get("split_embed_dim_list", []))
    use_split_head, use_spade = cfg.get("use_split_head", False), cfg.get("use_SPADE_norm", False)
    p = _P(sd, prefix)
    emb = timestep_embedding(t, cfg["model_channels"])
    emb = F.linear(emb, p("time_embed.0.weight"), p("time_embed.0.bias"))
    emb = F.linear(F.silu(emb), p("time_embed.2.weight"), p("time_embed.2.bias"))
    if num_stage > 1:
        emb = emb + p("stage_emb.weight")[stage][None]
    cond = None
    if use_split_head:
        c0 = sum(splits[:stage]) if use_spade else 0
        c1 = sum(splits[:stage + 1])
        h = _conv(p, f"pre_input_blocks.{stage}.0", x[:, c0:c1])
        if c0:
            cond = _conv(p, f"pre_input_cond_blocks.{stage - 1}.0", x[:, :c0])
    else:
        h = _conv(p, "input_blocks.0.0", x)
    if taps is not None:
        taps["pre"] = h
    hs = [h]
    for i, blk in enumerate(input_blocks):
        h = _run(p, blk, h, emb, context, cond, depth)
        hs.append(h)
        if taps is not None and i == 0:
            taps["ib0"] = h
    h = _run(p, middle, h, emb, context, cond, depth)
    if taps is not None:
        taps["mid"] = h
    for blk in output_blocks:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run(p, blk, h, emb, context, cond, depth)
    if taps is not None:
        taps["ob_last"] = h
    o = f"out.{stage}" if use_split_head else "out"
    h = F.group_norm(h, 32, p(o + ".0.weight"), p(o + ".0.bias"), 1e-5)
    return _conv(p, o + ".2", F.silu(h))
