"""Round-6 fixtures, generated like make_golden.py by RUNNING THE REFERENCE's own modules (imported read-only from /root/reference, same stubs):

    python tests/golden/make_golden_r6.py

  size_embedder.npz   the micro-conditioning of sample_size-128 models: two SizeEmbedder modules (PixArt_blocks.py:366-399; hidden size C / 3) on
                      c_size = (height, width) and ar = height / width, wired t + cat([csize, ar]) as diffusion/model/nets/controlnet.py:189-191
  dit_kvc_small.npz   PixArtMS (PixArtMS.py:82-248) with KV token compression in its self-attention (AttentionKVCompress, PixArt_blocks.py:60-158):
                      sampling 'conv' with scale factor 2 in one of two blocks; 'uniform' + qk_norm in the other variant

Nothing from the reference is copied: inputs, expected outputs and weight checksums only (weights come from tests/golden/_det.py)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
from tests.golden import make_golden as MG  # noqa: E402
from tests.golden._det import checksum, det_input, det_state_dict  # noqa: E402


@torch.no_grad()
def golden_size_embedder():
    from diffusion.model.nets.PixArt_blocks import SizeEmbedder
    S = 96   # C / 3 of the 4 x 72 test model
    cs, ar = SizeEmbedder(S).eval(), SizeEmbedder(S).eval()
    shapes = {k: tuple(v.shape) for k, v in cs.state_dict().items()}
    sd_c, sd_a = det_state_dict(shapes, seed=611), det_state_dict(shapes, seed=612)
    cs.load_state_dict(sd_c)
    ar.load_state_dict(sd_a)
    out = {}
    for name, (h, w) in (("16x24", (16, 24)), ("64x64", (64, 64)), ("128x96", (128, 96))):
        c_size = torch.tensor([[float(h), float(w)]])
        ratio = torch.tensor([[float(h) / float(w)]])
        out["add_" + name] = torch.cat([cs(c_size, 2), ar(ratio, 2)], dim=1)   # [2, 3 S]: what controlnet.py:191 adds to t
    MG.save("size_embedder.npz", wsum_c=checksum(sd_c), wsum_a=checksum(sd_a), **out,
            note="SizeEmbedder x 2 from the reference; keys mlp.0 / mlp.2 map to linear_1 / linear_2 of the diffusers TimestepEmbedding")


@torch.no_grad()
def golden_dit_kvc():
    from diffusion.model.nets.PixArtMS import PixArtMS
    from oracle.dit import pixart_to_diffusers
    depth, heads, hidden, cap, ntok = 2, 4, 288, 64, 20   # 4 heads of 72: a width the HIP path takes (hidden % 32 == 0), so that it can be held to this fixture
    lat = det_input(17, (2, 4, 16, 24), -2, 2)
    y = det_input(19, (1, 1, ntok, cap), -1, 1)
    t = torch.full((2,), 400.0)
    out = {}
    for name, kvc, qkn in (("conv", {"sampling": "conv", "scale_factor": 2, "kv_compress_layer": [1]}, False),
                           ("uniform_qknorm", {"sampling": "uniform", "scale_factor": 2, "kv_compress_layer": [0, 1]}, True),
                           ("ave", {"sampling": "ave", "scale_factor": 2, "kv_compress_layer": [0]}, False)):
        m = PixArtMS(input_size=16, patch_size=2, in_channels=4, hidden_size=hidden, depth=depth, num_heads=heads, caption_channels=cap,
                     model_max_length=ntok, kv_compress_config=kvc, qk_norm=qkn).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if k not in ("pos_embed", "y_embedder.y_embedding")}
        sd = det_state_dict(shapes, seed=909)
        for k in sd:   # norms with a visible affine part (det weights are centred on 0: a LayerNorm weight near 0 would hide the branch)
            if k.endswith(("attn.norm.weight", "attn.q_norm.weight", "attn.k_norm.weight")):
                sd[k] = sd[k] + 1.0
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and set(missing) <= {"pos_embed", "y_embedder.y_embedding"}, (missing, unexpected)
        yy = y.expand(2, -1, -1, -1).contiguous()
        out["out_" + name] = m(lat, t, yy, mask=None)
        out["wsum_" + name] = checksum(sd)
        out["wsum_diffusers_" + name] = checksum(pixart_to_diffusers(sd, depth))
        print(name, {k: v for k, v in shapes.items() if ".attn.sr." in k or ".attn.norm." in k or "q_norm" in k}.keys())
    base = PixArtMS(input_size=16, patch_size=2, in_channels=4, hidden_size=hidden, depth=depth, num_heads=heads, caption_channels=cap, model_max_length=ntok).eval()
    MG.save("dit_kvc_small.npz", lat=lat, y=y[0], **out,
            note="PixArtMS with kv_compress_config / qk_norm from the reference; Mlp / Attention-ctor / PatchEmbed / xformers attention come from make_golden.py's shims")


if __name__ == "__main__":
    MG.install_stubs()
    which = sys.argv[1:] or ["size", "kvc"]
    if "size" in which:
        golden_size_embedder()
    if "kvc" in which:
        golden_dit_kvc()
