"""ORACLE (test infrastructure only): CPU restatement of the cond-stage text/layout encoder
frido/modules/encoders/modules.py:85-114 (BERTEmbedder with use_tokenizer=False) =
frido/modules/x_transformer.py TransformerWrapper(Encoder(dim, depth)) with return_embeddings=True:
token + absolute position embedding (548-622), `depth` x [pre-LN 8-head attention (215-366, dim_head 64, no qkv bias),
pre-LN FeedForward Linear-GELU-Linear (194-211)] with residuals (481-531), final LayerNorm (623).
Pinned against the `c` arrays in tests/golden/sampler_small*.npz (captured from the reference's BERTEmbedder)."""
import torch
import torch.nn.functional as F


@torch.no_grad()
def bert_embed(sd, tokens, depth, heads=8, prefix="cond_stage_model."):
    p = lambda n: sd[prefix + "transformer." + n]
    B, n = tokens.shape
    x = p("token_emb.weight")[tokens] + p("pos_emb.emb.weight")[:n][None]
    dim = x.shape[-1]
    ln = lambda name, z: F.layer_norm(z, (dim,), p(name + ".weight"), p(name + ".bias"), 1e-5)
    for layer in range(depth):
        a, f = f"attn_layers.layers.{2 * layer}", f"attn_layers.layers.{2 * layer + 1}"
        h = ln(a + ".0", x)
        q, k, v = (F.linear(h, p(f"{a}.1.to_{t}.weight")) for t in "qkv")
        dh = q.shape[-1] // heads
        q, k, v = (t.view(B, n, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        dots = torch.einsum("bhid,bhjd->bhij", q, k) * dh ** -0.5
        o = torch.einsum("bhij,bhjd->bhid", dots.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(B, n, heads * dh)
        x = F.linear(o, p(a + ".1.to_out.weight"), p(a + ".1.to_out.bias")) + x
        h = ln(f + ".0", x)
        h = F.gelu(F.linear(h, p(f + ".1.net.0.0.weight"), p(f + ".1.net.0.0.bias")))
        x = F.linear(h, p(f + ".1.net.2.weight"), p(f + ".1.net.2.bias")) + x
    return ln("norm", x)
