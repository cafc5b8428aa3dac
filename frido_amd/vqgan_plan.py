"""Compiles the MS-VQGAN decode path into a HIP program:
  taming/models/msvqgan.py:376-399       VQModelInterface.decode (per-scale VQ, reversed concat, post_quant_conv)
  taming/modules/diffusionmodules/model.py:618-649   Decoder.forward (ResnetBlock 117-137, AttnBlock 168-192,
                                                     Upsample 49-53, Normalize eps 1e-6)
fused with the per-scale 1/scale_factor of frido/models/diffusion/frido.py:832-838.
No host round trip: code indices stay on the device (the reference's `.tolist()` sync, msvqgan.py:390,
is only materialised when the caller asks for return_code=True).
"""
import torch

from .arch import decoder_arch
from .builder import Builder, ACT_NONE, ACT_SILU
from .engine import rup


class _T:
    def __init__(self, t):
        self.t = t
        self.rows, self.C = t.shape[0], t.shape[1]

    @property
    def ptr(self):
        return self.t.data_ptr()

    def free(self):
        pass


def vq_res(b, B, blk_prefix, x, h, w):
    """taming ResnetBlock (model.py:117-137, temb None, eps 1e-6) on an NHWC activation."""
    HW = h * w
    pre = blk_prefix
    has_nin = (pre + ".nin_shortcut.weight") in b.w
    a1, raw = b.groupnorm(x, None, B, HW, pre + ".norm1", 1e-6, act=ACT_SILU, want_raw=has_nin)
    hmid = b.conv(a1, B, h, w, pre + ".conv1")
    a1.free()
    a2, _ = b.groupnorm(hmid, None, B, HW, pre + ".norm2", 1e-6, act=ACT_SILU)
    hmid.free()
    if has_nin and raw.K % 64 == 0 and a2.K % 64 == 0:
        out = b.conv_plus_skip(a2, raw, B, h, w, pre + ".conv2", pre + ".nin_shortcut")
        raw.free()
    elif has_nin:
        res = b.linear(raw, pre + ".nin_shortcut")
        raw.free()
        out = b.conv(a2, B, h, w, pre + ".conv2", residual=res, out=("f32", res))
    else:
        out = b.conv(a2, B, h, w, pre + ".conv2", residual=x)
    a2.free()
    return out


def vq_attn(b, B, blk_prefix, C, x, h, w):
    """taming AttnBlock (model.py:168-192): biased 1x1 q/k/v, single head, softmax over keys, proj_out + residual."""
    HW = h * w
    pre = blk_prefix
    a0, _ = b.groupnorm(x, None, B, HW, pre + ".norm", 1e-6, act=ACT_NONE)
    # scores: q' = a0 (W_q^T W_k) + W_k^T b_q against the raw rows of a0 (key-side bias terms cancel in the softmax)
    wq, bq = b.folded_qk_weight(pre + ".q", pre + ".k", "q")
    vkey = ("vT", C, HW, B)
    if vkey not in b._wcache:
        b._wcache[vkey] = b.persistent_op(C, rup(HW, 32), batch=B, zero=True)
    vT = b._wcache[vkey]
    # proj_out folded into v (single head): PV + (W_o b_v + b_o) + x lands directly on the residual stream
    wvo, bvo = b.folded_vo_weight(pre + ".v", pre + ".proj_out")
    b.prog.sync(0, 1)                  # V^T on the side stream, q' on the main one (both read a0), joined before the core
    with b.prog.side():
        b.v_transposed(a0, C, wvo, B, HW, C, out=vT)
    qp = b.linear(a0, None, wop=wq, bias_ptr=bq, bias=False, out="op")
    b.prog.sync(1, 0)
    out = b.attention(qp, C, a0, C, vT, B, HW, HW, C, bias_ptr=bvo, residual=x, stream=True)
    qp.free()
    a0.free()
    return out


def vq_run_blocks(b, B, blocks, cur, h, w, keep=()):
    """Run a list of VBlk on `cur`; returns (out, h, w).  `cur` is freed unless it is in `keep`."""
    for blk in blocks:
        if blk.kind == "res":
            nxt = vq_res(b, B, blk.prefix, cur, h, w)
        elif blk.kind == "attn":
            nxt = vq_attn(b, B, blk.prefix, blk.cin, cur, h, w)
        elif blk.kind == "up":       # model.py:49-53
            xo = b.to_operand(cur)
            nxt = b.upsample_conv(xo, B, h, w, blk.prefix + ".conv")
            xo.free()
            h, w = h * 2, w * 2
        else:                        # 'down' (model.py:68-72): zero-pad right/bottom by one, conv3x3 stride 2 pad 0
            xo = b.to_operand(cur)
            nxt = b.conv(xo, B, h, w, blk.prefix + ".conv", stride=2, pad=0, Ho=h // 2, Wo=w // 2)
            xo.free()
            h, w = h // 2, w // 2
        if not any(cur is k for k in keep):
            cur.free()
        cur = nxt
    return cur, h, w


def decoder_body(b, B, dd, prefix, z_op, h, w, out):
    """Decoder.forward (model.py:618-649) from an operand latent to `out` (("f32", buffer) or "f32_strict")."""
    a = decoder_arch(dd, prefix)
    cur = b.conv(z_op, B, h, w, prefix + ".conv_in")
    cur, h, w = vq_run_blocks(b, B, a.body, cur, h, w)
    ao, _ = b.groupnorm(cur, None, B, h * w, prefix + ".norm_out", 1e-6, act=ACT_SILU)
    cur.free()
    res = b.conv(ao, B, h, w, prefix + ".conv_out", out=out)
    ao.free()
    return res, h, w


class VQDecodePlan:
    def __init__(self, b: Builder, ddconfig, embed_dim, n_embed, *, B, h, w, z_state, inv_scale, forced=False, u8_mode=0):
        """z_state: device f32 [B][h*w][sum(embed_dim)] NHWC latent (already in diffusion scale);
        inv_scale[i] multiplies scale i before quantisation."""
        self.b = b
        a = self.a = decoder_arch(ddconfig, "decoder")
        self.B, self.h, self.w = B, h, w
        dev = b.device
        Ct = sum(embed_dim)
        hw = h * w
        up_levels = sum(1 for blk in a.body if blk.kind == "up")
        self.H, self.W = h << up_levels, w << up_levels
        # u8_mode 1 / 2: conv_out's epilogue writes the uint8 HWC image of scripts/sample_diffusion.py:103-121 (custom_to_np /
        # custom_to_pil) instead of the f32 plane -- no f32 image tensor exists in that plan at all
        self.out_u8 = torch.zeros(B * self.H * self.W, a.out_ch, dtype=torch.uint8, device=dev) if u8_mode else None
        self.out_nhwc = None if u8_mode else torch.zeros(B * self.H * self.W, a.out_ch, dtype=torch.float32, device=dev)
        self.idx = [torch.zeros(B * hw, dtype=torch.int64, device=dev) for _ in embed_dim]
        self.quant = torch.zeros(B * hw, Ct, dtype=torch.float32, device=dev)
        self.force_idx = [torch.zeros(B * hw, dtype=torch.int64, device=dev) for _ in embed_dim] if forced else None
        prog = self.prog = b.new_prog()
        start = 0
        for i, e in enumerate(embed_dim):
            cb = b.dev_f32(f"ms_quantize.{i}.embedding.weight")
            prog.emit("FRIDO_OP_VQ", x=z_state.data_ptr(), npix=B * hw, Cx=Ct, c0=start, e=e, inv_scale=float(inv_scale[i]),
                      codebook=cb.data_ptr(), n_codes=n_embed[i], zq=self.quant.data_ptr(), Cq=Ct,
                      q0=sum(embed_dim[i + 1:]), idx=self.idx[i].data_ptr(),
                      force_idx=self.force_idx[i].data_ptr() if forced else None)
            start += e
        q_op = b.pack(self.quant.data_ptr(), 1, B * hw, Ct, 0, Ct)
        pq = b.linear(q_op, "post_quant_conv", out="f32_strict")
        q_op.free()
        z_op = b.pack(pq.ptr, 1, B * hw, pq.C, 0, pq.C)
        pq.free()
        decoder_body(b, B, ddconfig, "decoder", z_op, h, w, ("u8", self.out_u8, u8_mode) if u8_mode else ("f32", _T(self.out_nhwc)))
        z_op.free()


class VQEncodePlan:
    """VQModelInterface.encode (msvqgan.py:326-374): MSEncoder (model.py:512-546) -> coarse-to-fine pre-quant features
    (quant_conv, VQ, ConvTranspose upsample, shared decoder) -> nearest-upsampled channel concat [coarse .. fine],
    optionally scaled per scale like get_first_stage_encoding (frido.py:654-662)."""

    def __init__(self, b: Builder, vq_cfg, *, B, H, W, x_in, scale):
        from .arch import encoder_arch
        from .holders import shared_decoder_cfg
        self.b = b
        ed, embed, n_embed = vq_cfg["edconfig"], vq_cfg["embed_dim"], vq_cfg["n_embed"]
        a = encoder_arch(ed, "encoder")
        n = a.multiscale
        dev = b.device
        prog = self.prog = b.new_prog()
        x_op = b.pack(x_in.data_ptr(), B, H * W, ed["in_channels"], 0, ed["in_channels"], nchw=True)
        cur = b.conv(x_op, B, H, W, "encoder.conv_in")
        x_op.free()
        h, w = H, W
        level_out = []
        nlev = len(a.down)
        for li, blocks in enumerate(a.down):
            body = [blk for blk in blocks if blk.kind != "down"]
            cur, h, w = vq_run_blocks(b, B, body, cur, h, w)
            kept = li >= nlev - n          # the last `multiscale` level outputs feed the heads (model.py:525-531)
            if kept:
                level_out.append((cur, h, w))
            downs = [blk for blk in blocks if blk.kind == "down"]
            if downs:
                cur, h, w = vq_run_blocks(b, B, downs, cur, h, w, keep=[cur] if kept else [])
        # heads (fine first, like the reference's out_h); h_ms = reversed -> coarse first
        heads = []
        for i in range(n):
            feat, fh, fw = level_out[-(n - i)]
            hcur, _, _ = vq_run_blocks(b, B, a.heads[i], feat, fh, fw, keep=[feat])
            ao, _ = b.groupnorm(hcur, None, B, fh * fw, f"encoder.norm_out_ms.{i}", 1e-6, act=ACT_SILU)
            hcur.free()
            heads.append((ao, fh, fw, a.z_channels[i]))
        heads = heads[::-1]          # coarse first
        self.h_out = []
        prev = []                    # quantised maps of coarser scales: (f32 tensor [B*hw][e], h, w)
        for ii in range(n):
            ao, fh, fw, zc = heads[ii]
            hw = fh * fw
            if ii == 0:
                feat = b.conv(ao, B, fh, fw, f"encoder.conv_out_ms.{n - 1 - ii}", out="f32_strict")
                ao.free()
                q_in, q_c = feat, zc
            else:
                ctot = sum(embed[:ii]) + zc
                cat = torch.zeros(B * hw, ctot, dtype=torch.float32, device=dev)
                b._persist.append(cat)
                col = 0
                for j in range(ii):          # msvqgan.py:334-337: every coarser quant is upsampled again at each scale
                    pt, ph, pw = prev[j]
                    up = torch.zeros(B * ph * pw * 4, embed[0], dtype=torch.float32, device=dev)
                    b._persist.append(up)
                    prog.emit("FRIDO_OP_CONVT", src=pt.data_ptr(), dst=up.data_ptr(), weight=b.dev_f32(f"upsample.{ii - 1}.weight").data_ptr(),
                              bias=b.bias(f"upsample.{ii - 1}.bias"), B=B, h=ph, w=pw, Cin=embed[0], Cout=embed[0])
                    up_op = b.pack(up.data_ptr(), 1, B * ph * pw * 4, embed[0], 0, embed[0])
                    pq = torch.zeros(B * ph * pw * 4, ed["z_channels"][0], dtype=torch.float32, device=dev)
                    b._persist.append(pq)
                    b.linear(up_op, f"shared_post_quant_conv.{ii - 1}", out=("f32", _T(pq)))
                    up_op.free()
                    prev[j] = (pq, ph * 2, pw * 2)
                    assert (ph * 2, pw * 2) == (fh, fw)
                    prog.emit("FRIDO_OP_RELAYOUT", src=pq.data_ptr(), dst=cat.data_ptr(), B=1, HW=B * hw, Csrc=pq.shape[1], c0=0,
                              Cuse=pq.shape[1], Cdst=ctot, d0=col, to_nchw=2)
                    col += pq.shape[1]
                fine = _T(torch.zeros(B * hw, zc, dtype=torch.float32, device=dev))
                b._persist.append(fine.t)
                b.conv(ao, B, fh, fw, f"encoder.conv_out_ms.{n - 1 - ii}", out=("f32", fine))
                ao.free()
                prog.emit("FRIDO_OP_RELAYOUT", src=fine.ptr, dst=cat.data_ptr(), B=1, HW=B * hw, Csrc=zc, c0=0, Cuse=zc, Cdst=ctot,
                          d0=col, to_nchw=2)
                cat_op = b.pack(cat.data_ptr(), 1, B * hw, ctot, 0, ctot)
                sd, _, _ = decoder_body(b, B, shared_decoder_cfg(embed, ii - 1), f"shared_decoder.{ii - 1}", cat_op, fh, fw,
                                        "f32_strict")
                cat_op.free()
                q_in, q_c = sd, embed[0]
            q_op = b.pack(q_in.ptr, 1, B * hw, q_c, 0, q_c)
            q_in.free()
            hq = torch.zeros(B * hw, embed[ii], dtype=torch.float32, device=dev)
            b._persist.append(hq)
            b.linear(q_op, f"ms_quant_conv.{ii}", out=("f32", _T(hq)))
            q_op.free()
            self.h_out.append((hq, fh, fw))
            if ii < n - 1:
                zq = torch.zeros(B * hw, embed[ii], dtype=torch.float32, device=dev)
                b._persist.append(zq)
                cb = b.dev_f32(f"ms_quantize.{ii}.embedding.weight")
                prog.emit("FRIDO_OP_VQ", x=hq.data_ptr(), npix=B * hw, Cx=embed[ii], c0=0, e=embed[ii], inv_scale=1.0,
                          codebook=cb.data_ptr(), n_codes=n_embed[ii], zq=zq.data_ptr(), Cq=embed[ii], q0=0, idx=None)
                prev.append((zq, fh, fw))
        for t, _, _ in level_out:
            t.free()
        # channel concat [coarse, ..., fine], every scale nearest-upsampled to the finest grid
        fh, fw = self.h_out[-1][1], self.h_out[-1][2]
        self.out = torch.zeros(B, sum(embed), fh, fw, dtype=torch.float32, device=dev)
        d0 = 0
        for i, (hq, hh, ww) in enumerate(self.h_out):
            up = (fh // hh).bit_length() - 1
            prog.emit("FRIDO_OP_PLACE", src=hq.data_ptr(), dst=self.out.data_ptr(), B=B, h=hh, w=ww, Csrc=embed[i], c0=0,
                      Cuse=embed[i], Cdst=sum(embed), d0=d0, up_shift=up, scale=float(scale[i]))
            d0 += embed[i]
