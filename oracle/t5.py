"""CPU fp32 restatement of the T5 v1.1 ENCODER that produces the prompt embeddings.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never imported by instarevive_amd.

The reference wraps a third-party model: diffusion/model/t5.py:82-101 calls `transformers.T5EncoderModel(input_ids, attention_mask)
['last_hidden_state']` on DeepFloyd/t5-v1_1-xxl (d_model 4096, 64 heads of 64, d_ff 10240, 24 blocks, gated-GELU, 32 relative
position buckets, max distance 128) after tokenising with max_length padding (t5.py:85-93; test_scripts/test_controlnet.py:383-402
writes the result to the prompt .pth file). The algorithm itself lives in `transformers` (not vendored in /root/reference; the
container has transformers 5.15.0, which tests/golden/make_golden.py runs on a reduced configuration to pin this restatement):
T5Stack = embedding -> N x [RMSNorm -> self-attention with an additive relative-position bias shared by all blocks and NO 1/sqrt(d)
scaling -> residual; RMSNorm -> wo(gelu_new(wi_0 x) * wi_1 x) -> residual] -> final RMSNorm; no biases anywhere.
"""
import math

import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=24, vocab_size=32128, relative_attention_num_buckets=32,
                   relative_attention_max_distance=128, layer_norm_epsilon=1e-6)


def relative_position_bucket(rel, num_buckets=32, max_distance=128):
    """Bidirectional bucket of rel = key position - query position (transformers T5Attention._relative_position_bucket)."""
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def position_bias(table, T, num_buckets=32, max_distance=128):
    """table: relative_attention_bias.weight [num_buckets, heads] -> additive bias [heads, T, T] (query, key)."""
    pos = torch.arange(T)
    buckets = relative_position_bucket(pos[None, :] - pos[:, None], num_buckets, max_distance)  # [q, k]
    return table[buckets].permute(2, 0, 1).contiguous()


def rmsnorm(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


@torch.no_grad()
def t5_encode(sd, input_ids, attention_mask=None, cfg=None):
    """sd: HF T5EncoderModel state dict (fp32). input_ids [B,T] long, attention_mask [B,T] (1 = token). Returns [B,T,d_model]."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg["layer_norm_epsilon"]
    B, T = input_ids.shape
    x = sd["shared.weight"][input_ids].float()
    bias = position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float(), T,
                         cfg["relative_attention_num_buckets"], cfg["relative_attention_max_distance"])[None]   # [1,H,T,T]
    if attention_mask is not None:
        bias = bias + (1.0 - attention_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer."
        h = rmsnorm(x, sd[p + "0.layer_norm.weight"], eps)
        q, k, v = (F.linear(h, sd[p + f"0.SelfAttention.{n}.weight"]).view(B, T, H, dk).transpose(1, 2) for n in ("q", "k", "v"))
        a = torch.softmax(q @ k.transpose(-1, -2) + bias, dim=-1) @ v          # no 1/sqrt(d_kv): folded into the initialisation of q
        x = x + F.linear(a.transpose(1, 2).reshape(B, T, H * dk), sd[p + "0.SelfAttention.o.weight"])
        h = rmsnorm(x, sd[p + "1.layer_norm.weight"], eps)
        g = gelu_new(F.linear(h, sd[p + "1.DenseReluDense.wi_0.weight"])) * F.linear(h, sd[p + "1.DenseReluDense.wi_1.weight"])
        x = x + F.linear(g, sd[p + "1.DenseReluDense.wo.weight"])
    return rmsnorm(x, sd["encoder.final_layer_norm.weight"], eps)


def state_dict_shapes(cfg=None):
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    D, H, dk, F_ = cfg["d_model"], cfg["num_heads"], cfg["d_kv"], cfg["d_ff"]
    s = {"shared.weight": (cfg["vocab_size"], D), "encoder.final_layer_norm.weight": (D,),
         "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": (cfg["relative_attention_num_buckets"], H)}
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer."
        for n in ("q", "k", "v"):
            s[p + f"0.SelfAttention.{n}.weight"] = (H * dk, D)
        s[p + "0.SelfAttention.o.weight"] = (D, H * dk)
        s[p + "0.layer_norm.weight"] = (D,)
        s[p + "1.DenseReluDense.wi_0.weight"] = (F_, D)
        s[p + "1.DenseReluDense.wi_1.weight"] = (F_, D)
        s[p + "1.DenseReluDense.wo.weight"] = (D, F_)
        s[p + "1.layer_norm.weight"] = (D,)
    return s
