"""SwinIR stage alone at full architecture (timing / profiling target).   python tools/bench_swin.py [H W [iters]]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from instarevive_amd import weights as W
from instarevive_amd.models import SwinIR

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 2048)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = dict(embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2)
m = SwinIR(img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8, window_size=8, mlp_ratio=2, sf=8, img_range=1.0,
           upsampler="nearest+conv", resi_connection="1conv", unshuffle=True, unshuffle_scale=8)
m.load_state_dict(bench.random_state_dict(W.swinir_shapes(cfg), 1), strict=False)
m.to("cuda")
x = torch.rand(1, 3, h, w, device="cuda")
for _ in range(2):
    y = m(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = m(x)
e1.record()
torch.cuda.synchronize()
print(f"SwinIR {h}x{w}: {e0.elapsed_time(e1) / iters:.3f} ms per forward")
