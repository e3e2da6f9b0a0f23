#!/usr/bin/env python3
"""Time the DiT step with and without the ControlNet-Half branch at the released architecture
(28 blocks of width 1152, 13 copied blocks), random weights: python tools/bench_dit_control.py [latent_side ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from instarevive_amd import weights as W  # noqa: E402
from instarevive_amd.models import ControlTransformerHalf, Transformer2DModel  # noqa: E402


def main():
    sides = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
    cfg = dict(num_layers=28, num_attention_heads=16, attention_head_dim=72, caption_channels=4096)
    base = Transformer2DModel()
    base.load_state_dict(bench.random_state_dict(W.dit_shapes(cfg), 3))
    base.to("cuda")
    ctl = ControlTransformerHalf(base, 13)
    sd = {k: (v if "copied_block" in k else torch.randn_like(v) * 0.02) for k, v in ctl._sd.items()}  # non-zero projections
    ctl.load_state_dict(dict({"base_model." + k: v for k, v in base._sd.items()}, **sd))
    y, mask = torch.randn(1, 300, 4096).cuda(), torch.ones(1, 1, 300).cuda()
    for s in sides:
        lat, c = torch.randn(1, 4, s, s).cuda(), torch.randn(1, 4, s, s).cuda()
        for name, fn in (("base", lambda: base.step(lat, 400.0, 0.5, y, mask)), ("control", lambda: ctl.step(lat, 400.0, 0.5, y, mask, c=c))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            print(f"latent {s}x{s} ({(s // 2) ** 2} tokens) {name:8s}: {(time.perf_counter() - t0) / n * 1e3:8.2f} ms", flush=True)


if __name__ == "__main__":
    main()
