"""CPU restatement of the ControlLDM path's text conditioner (test infrastructure; see oracle/__init__.py).

In-tree part, followed literally: FrozenOpenCLIPEmbedder.encode_with_transformer / text_transformer_forward
(/root/reference/ldm/modules/encoders/modules.py:176-193): token_embedding + positional_embedding, the resblocks up to
len(resblocks) - layer_idx (layer "penultimate": the last block is skipped, configs/cldm.yaml:86-90), ln_final.

Third-party part: the blocks themselves are open_clip's (`open_clip_torch`, not pinned in requirements.txt, neither in the tree nor installed
here), so **parity is UNPINNED** for them; they are restated from the published open_clip implementation (src/open_clip/transformer.py:
ResidualAttentionBlock = x + attn(ln_1(x), attn_mask) ; x + mlp(ln_2(x)) with nn.MultiheadAttention (packed in_proj_weight / in_proj_bias,
out_proj), mlp = c_fc -> nn.GELU -> c_proj; model.py: build_attention_mask = -inf above the diagonal) and anchored on the reference's
call site above. Parameter names are open_clip's (`transformer.resblocks.{i}.attn.in_proj_weight`, ...).
"""
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(width=1024, heads=16, layers=24, vocab_size=49408, context_length=77, mlp_ratio=4.0, layer="penultimate")


def state_dict_shapes(cfg=None):
    c = dict(DEFAULT_CFG, **(cfg or {}))
    d, f = c["width"], int(c["width"] * c["mlp_ratio"])
    s = {"token_embedding.weight": (c["vocab_size"], d), "positional_embedding": (c["context_length"], d), "ln_final.weight": (d,), "ln_final.bias": (d,)}
    for i in range(c["layers"]):
        p = f"transformer.resblocks.{i}"
        s.update({p + ".ln_1.weight": (d,), p + ".ln_1.bias": (d,), p + ".ln_2.weight": (d,), p + ".ln_2.bias": (d,),
                  p + ".attn.in_proj_weight": (3 * d, d), p + ".attn.in_proj_bias": (3 * d,), p + ".attn.out_proj.weight": (d, d), p + ".attn.out_proj.bias": (d,),
                  p + ".mlp.c_fc.weight": (f, d), p + ".mlp.c_fc.bias": (f,), p + ".mlp.c_proj.weight": (d, f), p + ".mlp.c_proj.bias": (d,)})
    return s


@torch.no_grad()
def encode_with_transformer(sd, tokens, cfg=None):
    """tokens: int64 [B, context_length] -> [B, context_length, width]."""
    c = dict(DEFAULT_CFG, **(cfg or {}))
    d, H, T = c["width"], c["heads"], tokens.shape[1]
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][:T]
    mask = torch.full((T, T), float("-inf")).triu_(1)
    n_run = c["layers"] - (1 if c["layer"] == "penultimate" else 0)
    for i in range(n_run):
        p = f"transformer.resblocks.{i}"
        h = F.layer_norm(x, (d,), sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"], 1e-5)
        qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
        q, k, v = (t.view(x.shape[0], T, H, d // H).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        a = torch.softmax(q @ k.transpose(-1, -2) * (d // H) ** -0.5 + mask, dim=-1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(x.shape[0], T, d), sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        h = F.layer_norm(x, (d,), sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"], 1e-5)
        x = x + F.linear(F.gelu(F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])), sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    return F.layer_norm(x, (d,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
