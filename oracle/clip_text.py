"""ORACLE (test infrastructure only): CPU restatement of the text tower of OpenAI CLIP as FrozenCLIPTextEmbedder uses it
(frido/modules/encoders/modules.py:188-219: `clip.load(version)` -> `model.encode_text(tokens)` -> optional L2 normalisation ->
`encode` adds a token axis and repeats it n_repeat times).

The algorithm lives in a THIRD-PARTY dependency that is absent from /root/reference: the `clip` package, installed from
`git+https://github.com/openai/CLIP.git@main` (environment.yaml:47, unpinned).  Restated here from its published
clip/model.py (CLIP.encode_text, Transformer / ResidualAttentionBlock / QuickGELU, build_attention_mask):

    x = token_embedding(text) + positional_embedding                  # [B, 77, width]
    per layer:  x = x + out_proj(MHA(ln_1(x), causal additive mask -inf above the diagonal))      (nn.MultiheadAttention)
                x = x + c_proj(QuickGELU(c_fc(ln_2(x))))               # QuickGELU(x) = x * sigmoid(1.702 x)
    x = ln_final(x);  z = x[arange(B), text.argmax(-1)] @ text_projection

PARITY UNPINNED: neither the package nor its weights exist here, so no golden vector pins this file to the reference's
runtime; the HIP tower (frido_amd/clip_plan.py) is checked against this restatement only."""
import torch
import torch.nn.functional as F


@torch.no_grad()
def clip_encode_text(sd, tokens, heads, prefix="cond_stage_model.model.", normalize=True):
    p = lambda n: sd[prefix + n]
    B, n = tokens.shape
    x = p("token_embedding.weight")[tokens] + p("positional_embedding")[:n][None]
    width = x.shape[-1]
    dh = width // heads
    mask = torch.full((n, n), float("-inf")).triu_(1)
    layer = 0
    while f"{prefix}transformer.resblocks.{layer}.ln_1.weight" in sd:
        r = f"transformer.resblocks.{layer}."
        h = F.layer_norm(x, (width,), p(r + "ln_1.weight"), p(r + "ln_1.bias"), 1e-5)
        q, k, v = F.linear(h, p(r + "attn.in_proj_weight"), p(r + "attn.in_proj_bias")).chunk(3, dim=-1)
        q, k, v = (t.view(B, n, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        att = (torch.einsum("bhid,bhjd->bhij", q * dh ** -0.5, k) + mask).softmax(dim=-1)
        o = torch.einsum("bhij,bhjd->bhid", att, v).permute(0, 2, 1, 3).reshape(B, n, width)
        x = x + F.linear(o, p(r + "attn.out_proj.weight"), p(r + "attn.out_proj.bias"))
        h = F.layer_norm(x, (width,), p(r + "ln_2.weight"), p(r + "ln_2.bias"), 1e-5)
        h = F.linear(h, p(r + "mlp.c_fc.weight"), p(r + "mlp.c_fc.bias"))
        x = x + F.linear(h * torch.sigmoid(1.702 * h), p(r + "mlp.c_proj.weight"), p(r + "mlp.c_proj.bias"))
        layer += 1
    x = F.layer_norm(x, (width,), p("ln_final.weight"), p("ln_final.bias"), 1e-5)
    z = x[torch.arange(B), tokens.argmax(dim=-1)] @ p("text_projection")
    if normalize:
        z = z / torch.linalg.norm(z, dim=1, keepdim=True)
    return z
