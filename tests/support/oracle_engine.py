"""Test infrastructure: the five phases of tiled sampling (the engine protocol of instarevive_amd.parallel.sharded_tiled_process)
implemented with the ORACLE on the CPU, so that the tile-sharding logic (who computes which tile, what travels, in which order the
receiving side accumulates) can be checked bit for bit on gloo ranks without a GPU. Not part of the product."""
import numpy as np
import torch

from oracle import dit as odit
from oracle import glue as oglue
from oracle import swinir as oswin
from oracle import vae as ovae
from tests.golden._det import det_state_dict

SWIN = dict(embed_dim=60, depths=[2, 2], num_heads=[6, 6])
VAE = dict(ch=32)
DIT = dict(num_layers=2, num_attention_heads=4, attention_head_dim=72, sample_size=16, caption_channels=64)


class OracleTileEngine:
    def __init__(self, y, color_fix_type="wavelet", tile_size=64, tile_stride=40, sf=0.18215):
        self.sws = det_state_dict(oswin.state_dict_shapes(SWIN), seed=101)
        self.svae = det_state_dict(ovae.state_dict_shapes(VAE), seed=202)
        self.sdit = det_state_dict(odit.state_dict_shapes(DIT), seed=404)
        self.y, self.fix, self.sf = y, color_fix_type, sf
        self.tl, self.sl = tile_size // 8, tile_stride // 8
        self.acp = oglue.alphas_cumprod_diffusers()

    def dit_fn(self, lat, t, yy, mm):
        return odit.dit_forward(self.sdit, lat, t, yy, mm, DIT)

    def count(self, h, w):
        return len(oglue.sliding_windows(h // 8, w // 8, self.tl, self.sl))

    def encode(self, control_imgs):
        control = torch.tensor(np.stack(control_imgs) / 255.0, dtype=torch.float32).clamp_(0, 1).permute(0, 3, 1, 2).contiguous()
        self.control = oswin.swinir_forward(self.sws, control, SWIN)
        self.wins = oglue.sliding_windows(self.control.shape[-2] // 8, self.control.shape[-1] // 8, self.tl, self.sl)
        return self.control, ovae.vae_encode_mean(self.svae, self.control * 2 - 1, VAE) * self.sf

    def stage1(self):
        a = (self.control.permute(0, 2, 3, 1) * 255).numpy().clip(0, 255).astype(np.uint8)
        return list(a)

    def dit_tiles(self, init, first, step):
        mine = self.wins[first::step]
        if not mine:
            return torch.zeros((0, init.shape[0], 4, self.tl, self.tl))
        return torch.stack([oglue.generate_sample_1step(self.dit_fn, self.acp, init[:, :, a:b, c:d], 400, self.y, None) for a, b, c, d in mine])

    def blend_latent(self, x0_all):
        nb = torch.zeros((x0_all.shape[1], 4, self.control.shape[-2] // 8, self.control.shape[-1] // 8))
        cnt = torch.zeros_like(nb)
        for t, (a, b, c, d) in zip(x0_all, self.wins):
            nb[:, :, a:b, c:d] += t
            cnt[:, :, a:b, c:d] += 1
        return nb.div_(cnt)

    def decode_tiles(self, nb, control, first, step):
        out = []
        for a, b, c, d in self.wins[first::step]:
            t = ovae.vae_decode(self.svae, nb[:, :, a:b, c:d] / self.sf, VAE) / 2 + 0.5
            cond = control[:, :, a * 8:b * 8, c * 8:d * 8]
            if self.fix == "adain":
                t = oglue.adaptive_instance_normalization(t, cond)
            elif self.fix == "wavelet":
                t = oglue.wavelet_reconstruction(t, cond)
            out.append(t)
        return torch.stack(out) if out else torch.zeros((0, nb.shape[0], 3, self.tl * 8, self.tl * 8))

    def blend_pixels(self, px_all):
        img = torch.zeros_like(self.control)
        cnt = torch.zeros_like(img)
        for t, (a, b, c, d) in zip(px_all, self.wins):
            img[:, :, a * 8:b * 8, c * 8:d * 8] += t
            cnt[:, :, a * 8:b * 8, c * 8:d * 8] += 1
        img.div_(cnt)
        return list((img.clamp(0, 1).permute(0, 2, 3, 1) * 255).numpy().clip(0, 255).astype(np.uint8))

    def reference(self, control_imgs):
        """The whole of oracle.glue.process(tiled=True) in one piece, for comparison."""
        return oglue.process(control_imgs, lambda x: oswin.swinir_forward(self.sws, x, SWIN), lambda x: ovae.vae_encode_mean(self.svae, x, VAE),
                             self.dit_fn, lambda z: ovae.vae_decode(self.svae, z, VAE), self.acp, self.y, None, scaling_factor=self.sf,
                             color_fix_type=self.fix, tiled=True, tile_size=self.tl * 8, tile_stride=self.sl * 8)
