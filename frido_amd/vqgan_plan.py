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


class VQDecodePlan:
    def __init__(self, b: Builder, ddconfig, embed_dim, n_embed, *, B, h, w, z_state, inv_scale):
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
        self.out_nhwc = torch.zeros(B * self.H * self.W, a.out_ch, dtype=torch.float32, device=dev)
        self.idx = [torch.zeros(B * hw, dtype=torch.int64, device=dev) for _ in embed_dim]
        self.quant = torch.zeros(B * hw, Ct, dtype=torch.float32, device=dev)
        prog = self.prog = b.new_prog()
        start = 0
        for i, e in enumerate(embed_dim):
            cb = b.dev_f32(f"ms_quantize.{i}.embedding.weight")
            prog.emit("FRIDO_OP_VQ", x=z_state.data_ptr(), npix=B * hw, Cx=Ct, c0=start, e=e, inv_scale=float(inv_scale[i]),
                      codebook=cb.data_ptr(), n_codes=n_embed[i], zq=self.quant.data_ptr(), Cq=Ct,
                      q0=sum(embed_dim[i + 1:]), idx=self.idx[i].data_ptr())
            start += e
        q_op = b.pack(self.quant.data_ptr(), 1, B * hw, Ct, 0, Ct)
        pq = b.linear(q_op, "post_quant_conv", out="f32_strict")
        q_op.free()
        z_op = b.pack(pq.ptr, 1, B * hw, pq.C, 0, pq.C)
        pq.free()
        cur = b.conv(z_op, B, h, w, "decoder.conv_in")
        z_op.free()
        ch, cw = h, w
        for blk in a.body:
            if blk.kind == "res":
                nxt = self._res(blk, cur, ch, cw)
            elif blk.kind == "attn":
                nxt = self._attn(blk, cur, ch, cw)
            else:  # up
                xo = b.to_operand(cur)
                nxt = b.conv(xo, B, ch, cw, blk.prefix + ".conv", up=1)
                xo.free()
                ch, cw = ch * 2, cw * 2
            cur.free()
            cur = nxt
        ao, _ = b.groupnorm(cur, None, B, ch * cw, "decoder.norm_out", 1e-6, act=ACT_SILU)
        cur.free()
        b.conv(ao, B, ch, cw, "decoder.conv_out", out=("f32", _T(self.out_nhwc)))
        ao.free()

    def _res(self, blk, x, h, w):
        b, B, HW = self.b, self.B, h * w
        pre = blk.prefix
        has_nin = (pre + ".nin_shortcut.weight") in b.w
        a1, raw = b.groupnorm(x, None, B, HW, pre + ".norm1", 1e-6, act=ACT_SILU, want_raw=has_nin)
        hmid = b.conv(a1, B, h, w, pre + ".conv1")
        a1.free()
        a2, _ = b.groupnorm(hmid, None, B, HW, pre + ".norm2", 1e-6, act=ACT_SILU)
        hmid.free()
        if has_nin:
            res = b.linear(raw, pre + ".nin_shortcut")
            raw.free()
            out = b.conv(a2, B, h, w, pre + ".conv2", residual=res, out=("f32", res))
        else:
            out = b.conv(a2, B, h, w, pre + ".conv2", residual=x)
        a2.free()
        return out

    def _attn(self, blk, x, h, w):
        b, B, HW, C = self.b, self.B, h * w, blk.cin
        pre = blk.prefix
        a0, _ = b.groupnorm(x, None, B, HW, pre + ".norm", 1e-6, act=ACT_NONE)
        wqk = b.cat_lin_weight(("vqk", pre), [pre + ".q.weight", pre + ".k.weight"])
        key = ("vqk_bias", pre)
        if key not in b._wcache:
            b._wcache[key] = torch.cat([b.w[pre + ".q.bias"].float(), b.w[pre + ".k.bias"].float()]).contiguous()
        qk = b.op(B * HW, 2 * C)
        b.linear(a0, None, wop=wqk, bias_ptr=b._wcache[key].data_ptr(), out=("op", qk))
        vT = b.persistent_op(C, rup(HW, 32), batch=B, zero=True)
        b.v_transposed(a0, C, b.lin_weight(pre + ".v.weight"), B, HW, C, bias_ptr=b.bias(pre + ".v.bias"), out=vT)
        a0.free()
        o = b.attention(qk, 2 * C, qk, 2 * C, vT, B, HW, HW, C, q_off=0, k_off=C)
        qk.free()
        out = b.linear(o, pre + ".proj_out", residual=x)
        o.free()
        return out
