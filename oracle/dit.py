"""CPU restatement of the PixArt-alpha DiT forward as the CLI runs it (test infrastructure; see oracle/__init__.py).

The CLI executes diffusers==0.30.0 `Transformer2DModel` (norm_type="ada_norm_single"); its source is not in the tree.
Block wiring follows the in-tree twin /root/reference/diffusion/model/nets/PixArtMS.py:71-79,165-211,236-248,
PixArt_blocks.py:24-25 (t2i_modulate), :43-58 (cross-attn), :123-158 (self-attn), :259-275 (final layer), :336-358
(timestep embedder, cos||sin), :439-463 (caption projection), PixArt.py:258-307 (2-D sincos table), and the key map /
hyper-parameters of tools/convert_pixart_to_diffusers.py:30-180. Where the twin and diffusers differ the diffusers
behaviour is restated, because that is what produces the reference outputs (UNPINNED here):
  * encoder_attention_mask with ndim == 3 (test_scripts/inference.py:274-277) skips diffusers' (1-m)*-10000 conversion
    and is ADDED to the cross-attention logits as is (+1 on real tokens, +0 on padding); a 2-D mask is converted.
  * the caption tokens are not dropped; all n_tok keys take part.
Weights use the diffusers Transformer2DModel key names (the InstaRevive_v1.ckpt surface); `pixart_to_diffusers`
restates the converter's renaming so the file can be pinned against the imported in-tree PixArtMS.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(num_layers=28, num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2,
                   sample_size=64, caption_channels=4096, interpolation_scale=1.0)


def sincos_pos_embed(embed_dim, grid_hw, base_size, interpolation_scale=1.0):
    """get_2d_sincos_pos_embed (PixArt.py:258-307 == diffusers embeddings.get_2d_sincos_pos_embed): [H*W, D] float64."""
    gh, gw = grid_hw
    grid_h = np.arange(gh, dtype=np.float32) / (gh / base_size) / interpolation_scale
    grid_w = np.arange(gw, dtype=np.float32) / (gw / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, gw, gh])

    def one(d, pos):
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    return np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)


def timestep_embedding(t, dim=256):
    """cos || sin, max_period 10000 (PixArt_blocks.py:336-351 == diffusers Timesteps(flip_sin_to_cos=True, shift=0))."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def size_embedding(sd, p, s):
    """SizeEmbedder.forward for ONE scalar per batch row (PixArt_blocks.py:384-396: sinusoid of the value, Linear -> SiLU -> Linear; the module
    flattens a [B, dims] input to B * dims scalars and concatenates their embeddings per row - the caller concatenates here). `p`: key prefix of
    the two linears in the diffusers layout (adaln_single.emb.resolution_embedder / aspect_ratio_embedder: TimestepEmbedding(256 -> C / 3))."""
    return _lin(sd, p + ".linear_2", F.silu(_lin(sd, p + ".linear_1", timestep_embedding(torch.as_tensor(s, dtype=torch.float32).reshape(-1)))))


def micro_condition(sd, B, H, W):
    """What sample_size == 128 models add to the timestep embedding: generate.py:56-62 passes resolution = (height, width) of the LATENT and
    aspect_ratio = height / width; diffusers' PixArtAlphaCombinedTimestepSizeEmbeddings (UNPINNED: diffusers is not in the tree) embeds the two
    resolution scalars with ONE embedder and the ratio with another and concatenates [res(h) | res(w) | ar] - the wiring of the in-tree twin,
    diffusion/model/nets/controlnet.py:189-191 (csize_embedder / ar_embedder, t = t + cat([csize, ar], dim=1)), which IS pinned (size_embedder.npz)."""
    r, a = "adaln_single.emb.resolution_embedder", "adaln_single.emb.aspect_ratio_embedder"
    parts = [size_embedding(sd, r, [float(H)] * B), size_embedding(sd, r, [float(W)] * B), size_embedding(sd, a, [float(H) / float(W)] * B)]
    return torch.cat(parts, dim=1)


def _heads(x, nh):
    B, T, C = x.shape
    return x.view(B, T, nh, C // nh).transpose(1, 2)


def cross_attention_bias(mask):
    """diffusers Transformer2DModel.forward: a 2-D mask becomes (1-m)*-10000 [B,1,L]; a 3-D mask is used as given."""
    if mask is None:
        return None
    if mask.ndim == 2:
        return ((1 - mask.to(torch.float32)) * -10000.0).unsqueeze(1)
    return mask.to(torch.float32)


@torch.no_grad()
def dit_forward(sd, latents, timestep, encoder_hidden_states, encoder_attention_mask=None, cfg=None, c=None):
    """latents [B,4,h,w], timestep [B] (or scalar), encoder_hidden_states [B|1,L,cap], mask [B|1,1,L] or [B|1,L].
    Returns the model output [B,8,h,w] (eps || sigma), i.e. Transformer2DModel(...).sample.
    c [B,4,h,w] (optional): the control latent of the ControlNet-Half variant; sd then also holds the
    `controlnet.{i}.{copied_block.*, after_proj.*}` / `controlnet.0.before_proj.*` tensors and the base model's keys unprefixed."""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    nh, hd, ps = cfg["num_attention_heads"], cfg["attention_head_dim"], cfg["patch_size"]
    C = nh * hd
    B, _, H, W = latents.shape
    gh, gw = H // ps, W // ps
    x = F.conv2d(latents, sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], stride=ps).flatten(2).transpose(1, 2)
    pos = sincos_pos_embed(C, (gh, gw), cfg["sample_size"] // ps, cfg["interpolation_scale"])
    x = x + torch.from_numpy(pos).float().unsqueeze(0)
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    emb = _lin(sd, "adaln_single.emb.timestep_embedder.linear_2", F.silu(_lin(sd, "adaln_single.emb.timestep_embedder.linear_1", timestep_embedding(t))))
    if cfg["sample_size"] == 128:   # use_additional_conditions
        emb = emb + micro_condition(sd, B, H, W)
    t6 = _lin(sd, "adaln_single.linear", F.silu(emb))  # [B, 6C]
    y = _lin(sd, "caption_projection.linear_2", F.gelu(_lin(sd, "caption_projection.linear_1", encoder_hidden_states), approximate="tanh"))
    y = y.reshape(-1, y.shape[-2], C)
    if y.shape[0] != B:
        y = y.expand(B, -1, -1)
    bias = cross_attention_bias(encoder_attention_mask)
    if bias is not None:
        bias = bias.reshape(-1, 1, 1, bias.shape[-1]).expand(B, 1, 1, -1)
    kvc = cfg.get("kv_compress") or {}
    def compress(p, tok):
        """AttentionKVCompress.downsample_2d (PixArt_blocks.py:97-121) on the tokens of ONE of k / v: [B, N, C] -> [B, N / r^2, C]. 'conv': a depthwise
        r x r / stride r convolution over the token grid (self.sr) followed by LayerNorm (self.norm, eps 1e-5); 'uniform': every r-th row and column;
        'ave': F.interpolate(scale_factor=1 / r, mode='nearest'), which for whole ratios picks the same tokens as 'uniform'."""
        r, mode = int(kvc["scale_factor"]), kvc["sampling"]
        g = tok.reshape(B, gh, gw, C).permute(0, 3, 1, 2)
        if mode == "conv":
            g = F.conv2d(g, sd[p + "attn1.sr.weight"], sd[p + "attn1.sr.bias"], stride=r, groups=C)
            t = g.reshape(B, C, -1).permute(0, 2, 1)
            return F.layer_norm(t, (C,), sd[p + "attn1.norm.weight"], sd[p + "attn1.norm.bias"], eps=1e-5)
        if mode == "uniform":
            g = g[:, :, ::r, ::r]
        elif mode == "ave":
            g = F.interpolate(g, scale_factor=1 / r, mode="nearest")
        else:
            raise ValueError(mode)
        return g.permute(0, 2, 3, 1).reshape(B, -1, C)

    def block(p, x):  # one BasicTransformerBlock (ada_norm_single), weights under prefix p
        sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = (sd[p + "scale_shift_table"][None] + t6.reshape(B, 6, -1)).chunk(6, dim=1)
        h = F.layer_norm(x, (C,), eps=1e-6) * (1 + sc_msa) + sh_msa
        q1, k1, v1 = _lin(sd, p + "attn1.to_q", h), _lin(sd, p + "attn1.to_k", h), _lin(sd, p + "attn1.to_v", h)
        if cfg.get("qk_norm"):   # AttentionKVCompress.q_norm / k_norm: LayerNorm over ALL C channels of the token, before the heads are split (:136-137)
            q1 = F.layer_norm(q1, (C,), sd[p + "attn1.q_norm.weight"], sd[p + "attn1.q_norm.bias"], eps=1e-5)
            k1 = F.layer_norm(k1, (C,), sd[p + "attn1.k_norm.weight"], sd[p + "attn1.k_norm.bias"], eps=1e-5)
        layer = int(p.split(".")[-2]) if p.startswith("transformer_blocks.") else -1
        if kvc and layer in kvc.get("layers", ()) and int(kvc.get("scale_factor", 1)) > 1:   # :140-142: k and v through the SAME sr / norm modules
            k1, v1 = compress(p, k1), compress(p, v1)
        a = F.scaled_dot_product_attention(_heads(q1, nh), _heads(k1, nh), _heads(v1, nh))
        x = x + g_msa * _lin(sd, p + "attn1.to_out.0", a.transpose(1, 2).reshape(B, -1, C))
        a = F.scaled_dot_product_attention(_heads(_lin(sd, p + "attn2.to_q", x), nh), _heads(_lin(sd, p + "attn2.to_k", y), nh),
                                           _heads(_lin(sd, p + "attn2.to_v", y), nh), attn_mask=bias)
        x = x + _lin(sd, p + "attn2.to_out.0", a.transpose(1, 2).reshape(B, -1, C))
        h = F.layer_norm(x, (C,), eps=1e-6) * (1 + sc_mlp) + sh_mlp
        return x + g_mlp * _lin(sd, p + "ff.net.2", F.gelu(_lin(sd, p + "ff.net.0.proj", h), approximate="tanh"))

    if c is None:
        for l in range(cfg["num_layers"]):
            x = block(f"transformer_blocks.{l}.", x)
    else:
        # ControlNet-Half (diffusion/model/nets/transformer_controlnet.py:98-173, pixart_controlnet.py:17-50): the first
        # copy_blocks_num blocks have trainable copies that process the control tokens c = pos_embed(c_latent); copy i feeds
        # base block i+1 through the zero-initialised after_proj; copy 0 starts from x + before_proj(c).
        ncopy = cfg.get("copy_blocks_num", 13)
        cs = F.conv2d(c, sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], stride=ps).flatten(2).transpose(1, 2)
        cs = cs + torch.from_numpy(pos).float().unsqueeze(0)
        x = block("transformer_blocks.0.", x)
        for i in range(1, ncopy + 1):
            q = f"controlnet.{i - 1}."
            if i == 1:
                cs = x + _lin(sd, q + "before_proj", cs)
            cs = block(q + "copied_block.", cs)
            x = block(f"transformer_blocks.{i}.", x + _lin(sd, q + "after_proj", cs))
        for l in range(ncopy + 1, cfg["num_layers"]):
            x = block(f"transformer_blocks.{l}.", x)
    shift, scale = (sd["scale_shift_table"][None] + emb[:, None]).chunk(2, dim=1)
    x = _lin(sd, "proj_out", F.layer_norm(x, (C,), eps=1e-6) * (1 + scale) + shift)
    oc = cfg["out_channels"]
    x = x.reshape(B, gh, gw, ps, ps, oc)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, oc, gh * ps, gw * ps)


def _block_to_diffusers(sd, q, p, o):
    """One block: in-tree keys under prefix q -> diffusers BasicTransformerBlock keys under prefix p (converter :62-154)."""
    o[p + "scale_shift_table"] = sd[q + "scale_shift_table"]
    for n, w, b in zip(("to_q", "to_k", "to_v"), sd[q + "attn.qkv.weight"].chunk(3, 0), sd[q + "attn.qkv.bias"].chunk(3, 0)):
        o[p + f"attn1.{n}.weight"], o[p + f"attn1.{n}.bias"] = w, b
    o[p + "attn1.to_out.0.weight"], o[p + "attn1.to_out.0.bias"] = sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"]
    o[p + "attn2.to_q.weight"], o[p + "attn2.to_q.bias"] = sd[q + "cross_attn.q_linear.weight"], sd[q + "cross_attn.q_linear.bias"]
    for n, w, b in zip(("to_k", "to_v"), sd[q + "cross_attn.kv_linear.weight"].chunk(2, 0), sd[q + "cross_attn.kv_linear.bias"].chunk(2, 0)):
        o[p + f"attn2.{n}.weight"], o[p + f"attn2.{n}.bias"] = w, b
    o[p + "attn2.to_out.0.weight"], o[p + "attn2.to_out.0.bias"] = sd[q + "cross_attn.proj.weight"], sd[q + "cross_attn.proj.bias"]
    o[p + "ff.net.0.proj.weight"], o[p + "ff.net.0.proj.bias"] = sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]
    o[p + "ff.net.2.weight"], o[p + "ff.net.2.bias"] = sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"]


def control_to_diffusers(sd, copy_blocks_num):
    """ControlPixArtHalf state dict (pixart_controlnet.py:55-69: `controlnet.{i}.copied_block.<in-tree block keys>`, `.after_proj`,
    `controlnet.0.before_proj`) -> the key names of the diffusers flavour ControlTransformerHalf (transformer_controlnet.py:62-76)."""
    o = {}
    for i in range(copy_blocks_num):
        _block_to_diffusers(sd, f"controlnet.{i}.copied_block.", f"controlnet.{i}.copied_block.", o)
        for n in ("after_proj",) + (("before_proj",) if i == 0 else ()):
            o[f"controlnet.{i}.{n}.weight"], o[f"controlnet.{i}.{n}.bias"] = sd[f"controlnet.{i}.{n}.weight"], sd[f"controlnet.{i}.{n}.bias"]
    return o


def pixart_to_diffusers(sd, num_layers):
    """tools/convert_pixart_to_diffusers.py:30-154 restated: in-tree PixArt(MS) keys -> diffusers Transformer2DModel keys."""
    o = {"pos_embed.proj.weight": sd["x_embedder.proj.weight"], "pos_embed.proj.bias": sd["x_embedder.proj.bias"]}
    for a, b in (("caption_projection.linear_1", "y_embedder.y_proj.fc1"), ("caption_projection.linear_2", "y_embedder.y_proj.fc2"),
                 ("adaln_single.emb.timestep_embedder.linear_1", "t_embedder.mlp.0"), ("adaln_single.emb.timestep_embedder.linear_2", "t_embedder.mlp.2"),
                 ("adaln_single.linear", "t_block.1"), ("proj_out", "final_layer.linear")):
        o[a + ".weight"], o[a + ".bias"] = sd[b + ".weight"], sd[b + ".bias"]
    o["scale_shift_table"] = sd["final_layer.scale_shift_table"]
    for d in range(num_layers):
        p, q = f"transformer_blocks.{d}.", f"blocks.{d}."
        o[p + "scale_shift_table"] = sd[q + "scale_shift_table"]
        for n, w, b in zip(("to_q", "to_k", "to_v"), sd[q + "attn.qkv.weight"].chunk(3, 0), sd[q + "attn.qkv.bias"].chunk(3, 0)):
            o[p + f"attn1.{n}.weight"], o[p + f"attn1.{n}.bias"] = w, b
        o[p + "attn1.to_out.0.weight"], o[p + "attn1.to_out.0.bias"] = sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"]
        o[p + "attn2.to_q.weight"], o[p + "attn2.to_q.bias"] = sd[q + "cross_attn.q_linear.weight"], sd[q + "cross_attn.q_linear.bias"]
        for n, w, b in zip(("to_k", "to_v"), sd[q + "cross_attn.kv_linear.weight"].chunk(2, 0), sd[q + "cross_attn.kv_linear.bias"].chunk(2, 0)):
            o[p + f"attn2.{n}.weight"], o[p + f"attn2.{n}.bias"] = w, b
        o[p + "attn2.to_out.0.weight"], o[p + "attn2.to_out.0.bias"] = sd[q + "cross_attn.proj.weight"], sd[q + "cross_attn.proj.bias"]
        o[p + "ff.net.0.proj.weight"], o[p + "ff.net.0.proj.bias"] = sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"]
        o[p + "ff.net.2.weight"], o[p + "ff.net.2.bias"] = sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"]
        # KV compression / qk norm of the in-tree AttentionKVCompress (PixArt_blocks.py:60-96): no diffusers 0.30 counterpart, so the keys keep the
        # module's own names under the diffusers attention prefix (attn.sr -> attn1.sr, attn.norm -> attn1.norm, attn.q_norm / k_norm likewise)
        for n in ("sr", "norm", "q_norm", "k_norm"):
            for leaf in ("weight", "bias"):
                if f"{q}attn.{n}.{leaf}" in sd:
                    o[f"{p}attn1.{n}.{leaf}"] = sd[f"{q}attn.{n}.{leaf}"]
    return o


def state_dict_shapes(cfg=None, mlp_ratio=4):
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    C = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    ps, cap = cfg["patch_size"], cfg["caption_channels"]
    s = {"pos_embed.proj.weight": (C, cfg["in_channels"], ps, ps), "pos_embed.proj.bias": (C,),
         "caption_projection.linear_1.weight": (C, cap), "caption_projection.linear_1.bias": (C,),
         "caption_projection.linear_2.weight": (C, C), "caption_projection.linear_2.bias": (C,),
         "adaln_single.emb.timestep_embedder.linear_1.weight": (C, 256), "adaln_single.emb.timestep_embedder.linear_1.bias": (C,),
         "adaln_single.emb.timestep_embedder.linear_2.weight": (C, C), "adaln_single.emb.timestep_embedder.linear_2.bias": (C,),
         "adaln_single.linear.weight": (6 * C, C), "adaln_single.linear.bias": (6 * C,),
         "proj_out.weight": (ps * ps * cfg["out_channels"], C), "proj_out.bias": (ps * ps * cfg["out_channels"],), "scale_shift_table": (2, C)}
    if cfg["sample_size"] == 128:
        for e in ("resolution_embedder", "aspect_ratio_embedder"):
            s[f"adaln_single.emb.{e}.linear_1.weight"], s[f"adaln_single.emb.{e}.linear_1.bias"] = (C // 3, 256), (C // 3,)
            s[f"adaln_single.emb.{e}.linear_2.weight"], s[f"adaln_single.emb.{e}.linear_2.bias"] = (C // 3, C // 3), (C // 3,)
    for d in range(cfg["num_layers"]):
        p = f"transformer_blocks.{d}."
        s[p + "scale_shift_table"] = (6, C)
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                s[p + f"{a}.{n}.weight"], s[p + f"{a}.{n}.bias"] = (C, C), (C,)
        s[p + "ff.net.0.proj.weight"], s[p + "ff.net.0.proj.bias"] = (mlp_ratio * C, C), (mlp_ratio * C,)
        s[p + "ff.net.2.weight"], s[p + "ff.net.2.bias"] = (C, mlp_ratio * C), (C,)
        kvc = cfg.get("kv_compress") or {}
        if kvc.get("sampling") == "conv" and d in kvc.get("layers", ()) and int(kvc.get("scale_factor", 1)) > 1:
            r = int(kvc["scale_factor"])
            s[p + "attn1.sr.weight"], s[p + "attn1.sr.bias"] = (C, 1, r, r), (C,)
            s[p + "attn1.norm.weight"], s[p + "attn1.norm.bias"] = (C,), (C,)
        if cfg.get("qk_norm"):
            for n in ("q_norm", "k_norm"):
                s[p + f"attn1.{n}.weight"], s[p + f"attn1.{n}.bias"] = (C,), (C,)
    return s


def control_state_dict_shapes(cfg=None, mlp_ratio=4, copy_blocks_num=13):
    """Extra tensors of ControlTransformerHalf (transformer_controlnet.py:19-40,62-76), keyed as its state_dict names them."""
    base = state_dict_shapes(cfg, mlp_ratio)
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    C = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    s = {"controlnet.0.before_proj.weight": (C, C), "controlnet.0.before_proj.bias": (C,)}
    for i in range(copy_blocks_num):
        s[f"controlnet.{i}.after_proj.weight"], s[f"controlnet.{i}.after_proj.bias"] = (C, C), (C,)
        for k, v in base.items():
            if k.startswith("transformer_blocks.0."):
                s[f"controlnet.{i}.copied_block." + k[len("transformer_blocks.0."):]] = v
    return s
