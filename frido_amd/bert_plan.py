"""Compiles the cond-stage encoder (BERTEmbedder, frido/modules/encoders/modules.py:85-114 over
frido/modules/x_transformer.py:215-366,481-531,598-623) into a HIP program: embedding gather, then per layer
LayerNorm -> fused QK projection + transposed V projection -> (batch x head) QK^T, softmax, PV on the MFMA GEMM
-> output projection (+residual) -> LayerNorm -> Linear+GELU -> Linear (+residual); final LayerNorm -> f32 context."""
import torch

from .builder import Builder
from .engine import rup

ACT_GELU = 3


class BertPlan:
    def __init__(self, b: Builder, *, B, n, dim, depth, vocab, heads=8, dim_head=64):
        self.b = b
        dev = b.device
        H, dh = heads, dim_head
        inner = H * dh
        self.tokens = torch.zeros(B * n, dtype=torch.int64, device=dev)
        self.out = torch.zeros(B * n, dim, dtype=torch.float32, device=dev)
        prog = self.prog = b.new_prog()
        tr = "transformer."
        x = b.f32_strict(B * n, dim)          # the 32-layer residual stream stays f32 (tiny: B*n rows)
        prog.emit("FRIDO_OP_EMBED", tokens=self.tokens.data_ptr(), tok=b.dev_f32(tr + "token_emb.weight").data_ptr(),
                  pos=b.dev_f32(tr + "pos_emb.emb.weight").data_ptr(), out=x.ptr, rows=B * n, n=n, D=dim, vocab=vocab)
        Np = rup(n, 32)
        vT = b.persistent_op(inner, Np, batch=B, zero=True)
        for layer in range(depth):
            a, f = f"{tr}attn_layers.layers.{2 * layer}", f"{tr}attn_layers.layers.{2 * layer + 1}"
            hn = b.layernorm(x, a + ".0")
            wqk = b.cat_lin_weight(("bert_qk", a), [a + ".1.to_q.weight", a + ".1.to_k.weight"])
            qk = b.op(B * n, 2 * inner)
            b.linear(hn, None, wop=wqk, bias=False, out=("op", qk))
            b.v_transposed(hn, dim, b.lin_weight(a + ".1.to_v.weight"), B, n, inner, out=vT)
            hn.free()
            # scores[b][h] = q[b][:, h*dh:(h+1)*dh] @ k[b][:, h*dh:...]^T * dh^-0.5
            s = b.f32_strict(B * H * n, n)
            prog.gemm(n, n, dh, qk, (qk.ptr + 2 * inner, qk.lo), batch=B * H, batch_inner=H, lda=2 * inner, ldb=2 * inner,
                      a_bs=n * 2 * inner, a_bs2=dh, b_bs=n * 2 * inner, b_bs2=dh, alpha=float(dh) ** -0.5,
                      out_f32=s.ptr, of_bs=H * n * n, of_bs2=n * n, ldo=n)
            pr = b.softmax(s, B * H * n, n, n, Np)
            s.free()
            o = b.op(B * n, inner)
            prog.gemm(n, dh, Np, pr, vT, batch=B * H, batch_inner=H, lda=Np, ldb=Np, a_bs=H * n * Np, a_bs2=n * Np,
                      b_bs=inner * Np, b_bs2=dh * Np, out_op=o.ptr, oo_bs=n * inner, oo_bs2=dh, ldoo=inner, oo_lo=o.lo)
            pr.free()
            qk.free()
            x2 = b.linear(o, a + ".1.to_out", residual=x, out="f32_strict")
            o.free()
            x.free()
            hn = b.layernorm(x2, f + ".0")
            h1 = b.linear(hn, f + ".1.net.0.0", act=ACT_GELU, out="op")
            hn.free()
            x = b.linear(h1, f + ".1.net.2", residual=x2, out="f32_strict")
            h1.free()
            x2.free()
        # final LayerNorm -> f32 context
        prog.emit("FRIDO_OP_LAYERNORM", x=x.ptr, rows=B * n, C=dim, eps=1e-5, weight=b.bias(tr + "norm.weight"),
                  bias=b.bias(tr + "norm.bias"), nsplit=b.nsplit, out_f32=self.out.data_ptr(), x_bf16=0)
        x.free()
