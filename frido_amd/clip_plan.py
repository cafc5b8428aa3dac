"""Compiles the text tower of OpenAI CLIP (the `encode_text` path FrozenCLIPTextEmbedder calls,
frido/modules/encoders/modules.py:188-219 -> clip/model.py `CLIP.encode_text`) into a HIP program:

    x = token_embedding[tokens] + positional_embedding                       (embed kernel)
    12 x ResidualAttentionBlock: x += out_proj(MHA_causal(ln_1(x)));  x += c_proj(QuickGELU(c_fc(ln_2(x))))
    z = ln_final(x)[b, argmax(tokens[b])] @ text_projection                  (row gather + GEMM)
    z = z / ||z||_2                                                          (l2norm kernel, `normalize=True`)

The attention is the (batch x head) QK^T GEMM -> causal row softmax -> PV GEMM chain the BERT plan uses; the causal mask is a
per-row key count inside the softmax kernel (FridoSoftmax.causal_nq).  The `clip` package is not a file of the reference
(it is a pip dependency, unpinned: environment.yaml `git+https://github.com/openai/CLIP.git@main`), so the algorithm is
restated from its published model.py; parity is pinned against oracle/clip_text.py only (no reference goldens exist)."""
import torch

from .builder import Builder
from .engine import rup

ACT_QUICKGELU = 4


class ClipTextPlan:
    def __init__(self, b: Builder, *, B, n, width, layers, heads, vocab, embed_dim, normalize=True, prefix="model."):
        self.b = b
        dev = b.device
        H, dh = heads, width // heads
        self.tokens = torch.zeros(B * n, dtype=torch.int64, device=dev)
        self.eot_rows = torch.zeros(B, dtype=torch.int64, device=dev)        # b * n + argmax(tokens[b]): filled per call
        self.out = torch.zeros(B, embed_dim, dtype=torch.float32, device=dev)
        prog = self.prog = b.new_prog()
        p = prefix
        x = b.f32_strict(B * n, width)
        prog.emit("FRIDO_OP_EMBED", tokens=self.tokens.data_ptr(), tok=b.dev_f32(p + "token_embedding.weight").data_ptr(),
                  pos=b.dev_f32(p + "positional_embedding").data_ptr(), out=x.ptr, rows=B * n, n=n, D=width, vocab=vocab)
        Np = rup(n, 32)
        vT = b.persistent_op(width, Np, batch=B, zero=True)
        for layer in range(layers):
            r = f"{p}transformer.resblocks.{layer}"
            hn = b.layernorm(x, r + ".ln_1")
            inb = b.dev_f32(r + ".attn.in_proj_bias")
            wqk = b.lin_weight(r + ".attn.in_proj_weight", rows=(0, 2 * width))
            qk = b.op(B * n, 2 * width)
            b.linear(hn, None, wop=wqk, bias_ptr=inb.data_ptr(), out=("op", qk))
            wv = b.lin_weight(r + ".attn.in_proj_weight", rows=(2 * width, 3 * width))
            b.v_transposed(hn, width, wv, B, n, width, bias_ptr=inb.data_ptr() + 4 * 2 * width, out=vT)
            hn.free()
            s = b.f32_strict(B * H * n, n)
            prog.gemm(n, n, dh, qk, (qk.ptr + 2 * width, qk.lo), batch=B * H, batch_inner=H, lda=2 * width, ldb=2 * width,
                      a_bs=n * 2 * width, a_bs2=dh, b_bs=n * 2 * width, b_bs2=dh, alpha=float(dh) ** -0.5,
                      out_f32=s.ptr, of_bs=H * n * n, of_bs2=n * n, ldo=n)
            pr = b.op(B * H * n, Np)
            prog.emit("FRIDO_OP_SOFTMAX", x=s.ptr, rows=B * H * n, N=n, ld=n, Npad=Np, nsplit=b.nsplit, out_op=pr.ptr, out_lo=pr.lo,
                      causal_nq=n)
            s.free()
            o = b.op(B * n, width)
            prog.gemm(n, dh, Np, pr, vT, batch=B * H, batch_inner=H, lda=Np, ldb=Np, a_bs=H * n * Np, a_bs2=n * Np,
                      b_bs=width * Np, b_bs2=dh * Np, out_op=o.ptr, oo_bs=n * width, oo_bs2=dh, ldoo=width, oo_lo=o.lo)
            pr.free()
            qk.free()
            x2 = b.linear(o, r + ".attn.out_proj", residual=x, out="f32_strict")
            o.free()
            x.free()
            hn = b.layernorm(x2, r + ".ln_2")
            h1 = b.linear(hn, r + ".mlp.c_fc", act=ACT_QUICKGELU, out="op")
            hn.free()
            x = b.linear(h1, r + ".mlp.c_proj", residual=x2, out="f32_strict")
            h1.free()
            x2.free()
        xf = b.f32_strict(B * n, width)
        prog.emit("FRIDO_OP_LAYERNORM", x=x.ptr, rows=B * n, C=width, eps=1e-5, weight=b.bias(p + "ln_final.weight"),
                  bias=b.bias(p + "ln_final.bias"), nsplit=b.nsplit, out_f32=xf.ptr, x_bf16=0)
        x.free()
        # the end-of-text token's row of every caption (the highest token id: clip/model.py encode_text), then the projection
        e = b.f32_strict(B, width)
        prog.emit("FRIDO_OP_EMBED", tokens=self.eot_rows.data_ptr(), tok=xf.ptr, pos=None, out=e.ptr, rows=B, n=1, D=width, vocab=B * n)
        xf.free()
        eo = b.pack(e.ptr, 1, B, width, 0, width)
        e.free()
        key = ("clip_proj", p)
        if key not in b._wcache:            # x @ text_projection  ==  Linear with weight text_projection^T
            from .engine import pack_matrix
            b._wcache[key] = pack_matrix(b.w[p + "text_projection"].float().t().contiguous(), b.nsplit)
        z = b.linear(eo, None, wop=b._wcache[key], bias=False, out="f32_strict")
        eo.free()
        if normalize:
            prog.emit("FRIDO_OP_L2NORM", x=z.ptr, out=self.out.data_ptr(), rows=B, C=embed_dim)
        else:
            prog.emit("FRIDO_OP_COPY", src=z.ptr, dst=self.out.data_ptr(), n=rup(B * embed_dim * 4, 16))
        z.free()
