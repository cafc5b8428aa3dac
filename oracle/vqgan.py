"""ORACLE (test infrastructure only): CPU restatement in plain PyTorch fp32 of the MS-VQGAN
decode / encode path.  Pinned against tests/golden/vq_*.npz captured from the reference's own
taming.models.msvqgan.VQModelInterface.  Functional over a flat state_dict (`first_stage_model.`
prefix in a full checkpoint).
"""
import torch
import torch.nn.functional as F

from .walk import decoder_blocks, encoder_blocks


class _P:
    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.sd


def _conv(p, name, x, stride=1, padding=1):
    return F.conv2d(x, p(name + ".weight"), p(name + ".bias"), stride=stride, padding=padding)


def _gn(p, name, x):
    return F.group_norm(x, 32, p(name + ".weight"), p(name + ".bias"), 1e-6)   # model.py:34-35


def resnet_block(p, pre, x):
    """taming/modules/diffusionmodules/model.py:117-137 with temb=None."""
    h = _conv(p, pre + ".conv1", F.silu(_gn(p, pre + ".norm1", x)))
    h = _conv(p, pre + ".conv2", F.silu(_gn(p, pre + ".norm2", h)))
    if p.has(pre + ".nin_shortcut.weight"):
        x = _conv(p, pre + ".nin_shortcut", x, padding=0)
    return x + h


def attn_block(p, pre, x):
    """model.py:168-192: biased 1x1 q/k/v, softmax over keys of q^T k * C^-0.5, 1x1 proj, residual."""
    B, C, H, W = x.shape
    h = _gn(p, pre + ".norm", x)
    q = _conv(p, pre + ".q", h, padding=0).reshape(B, C, H * W).permute(0, 2, 1)
    k = _conv(p, pre + ".k", h, padding=0).reshape(B, C, H * W)
    v = _conv(p, pre + ".v", h, padding=0).reshape(B, C, H * W)
    w = torch.bmm(q, k) * (int(C) ** (-0.5))
    w = F.softmax(w, dim=2)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(B, C, H, W)
    return x + _conv(p, pre + ".proj_out", o, padding=0)


def _run(p, blocks, h):
    for b in blocks:
        if b.kind == "res":
            h = resnet_block(p, b.prefix, h)
        elif b.kind == "attn":
            h = attn_block(p, b.prefix, h)
        elif b.kind == "up":      # model.py:49-53
            h = _conv(p, b.prefix + ".conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
        elif b.kind == "down":    # model.py:68-72: pad right/bottom by 1, conv3x3 s2 p0
            h = _conv(p, b.prefix + ".conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    return h


def quantize(codebook, z):
    """taming/modules/vqvae/quantize.py:267-308 (VectorQuantizer2.forward, inference part):
    argmin_j |z|^2 + |e_j|^2 - 2 z.e_j, gather, z + (z_q - z).  z: (B, C, H, W).
    Returns (z_q, flat indices (B*H*W,))."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, codebook.shape[1])
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) \
        - 2 * torch.einsum("bd,dn->bn", zf, codebook.t())
    idx = torch.argmin(d, dim=1)
    zq = codebook[idx].view(zp.shape)
    zq = zp + (zq - zp)
    return zq.permute(0, 3, 1, 2).contiguous(), idx


def decoder_forward(p, dd, z, prefix="decoder"):
    """model.py:618-649."""
    h = _conv(p, prefix + ".conv_in", z)
    h = _run(p, decoder_blocks(p.sd, p.prefix, prefix), h)        # (r06: the block list is read off the checkpoint's keys, oracle/walk.py; `dd` is no longer consulted)
    h = F.silu(_gn(p, prefix + ".norm_out", h))
    return _conv(p, prefix + ".conv_out", h)


@torch.no_grad()
def vq_decode(sd, cfg, h_in, prefix="first_stage_model.", return_code=False):
    """taming/models/msvqgan.py:376-399: per-scale quantise, REVERSED concat, post_quant_conv, Decoder."""
    p = _P(sd, prefix)
    embed = cfg["embed_dim"]
    qs, codes = [], []
    start = 0
    for i, e in enumerate(embed):
        zq, idx = quantize(p(f"ms_quantize.{i}.embedding.weight"), h_in[:, start:start + e])
        qs.append(zq)
        codes.append(idx.reshape(len(h_in), -1))
        start += e
    quant = torch.cat(qs[::-1], dim=1)
    quant = _conv(p, "post_quant_conv", quant, padding=0)
    dec = decoder_forward(p, cfg["ddconfig"], quant)
    return (dec, codes) if return_code else dec


def encoder_forward(p, ed, x, prefix="encoder"):
    """model.py:512-546 (MSEncoder.forward): returns the `multiscale` head outputs, fine first."""
    down, heads = encoder_blocks(p.sd, p.prefix, prefix)      # (r06: read off the checkpoint's keys, oracle/walk.py)
    multiscale = len(heads)
    h = _conv(p, prefix + ".conv_in", x)
    level_out = []
    for blocks in down:
        for b in blocks:
            if b.kind == "down":
                level_out.append(h)
            h = _run(p, [b], h)
        if blocks[-1].kind != "down":
            level_out.append(h)
    outs = []
    for i in range(multiscale):
        h = level_out[-(multiscale - i)]
        h = _run(p, heads[i], h)
        h = F.silu(_gn(p, f"{prefix}.norm_out_ms.{i}", h))
        outs.append(_conv(p, f"{prefix}.conv_out_ms.{i}", h))
    return outs


@torch.no_grad()
def vq_encode(sd, cfg, x, prefix="first_stage_model."):
    """msvqgan.py:326-374: coarse-to-fine pre-quant features; coarse scales nearest-upsampled; channel
    concat [coarse, ..., fine]."""
    p = _P(sd, prefix)
    n = len(cfg["embed_dim"])
    h_ms = encoder_forward(p, cfg["edconfig"], x)[::-1]      # coarse first
    prev, h_out = [], []
    for ii in range(n):
        if prev:
            for j in range(ii):
                up = F.conv_transpose2d(prev[j], p(f"upsample.{ii - 1}.weight"), p(f"upsample.{ii - 1}.bias"),
                                        stride=2, padding=1)
                prev[j] = _conv(p, f"shared_post_quant_conv.{ii - 1}", up, padding=0)
            q = torch.cat((*prev[:ii], h_ms[ii]), dim=1)
            sdd = dict(double_z=False, z_channels=sum(cfg["embed_dim"][:ii + 1]), resolution=256, in_channels=None,
                       out_ch=cfg["embed_dim"][0], ch=128, ch_mult=[1], num_res_blocks=2,
                       attn_resolutions=[2, 4, 8, 16, 32, 64], dropout=0.0)
            q = decoder_forward(p, sdd, q, prefix=f"shared_decoder.{ii - 1}")
        else:
            q = h_ms[ii]
        h = _conv(p, f"ms_quant_conv.{ii}", q, padding=0)
        h_out.append(h)
        zq, _ = quantize(p(f"ms_quantize.{ii}.embedding.weight"), h)
        prev.append(zq)
    h_out = h_out[::-1]
    for i in range(len(h_out)):
        for _ in range(i):
            h_out[i] = F.interpolate(h_out[i], scale_factor=2)
    return torch.cat(h_out[::-1], dim=1)
