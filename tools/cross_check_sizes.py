#!/usr/bin/env python3
"""Cross-check of the fast kernels at awkward sizes: run process() at full architecture with the default kernels and again, in a second
process, with the ping-pong / big-tile kernels switched off (IR_NO_CONV_PP, IR_NO_GEMM_PP, IR_NO_PINGPONG: same arithmetic through the
older 4-wave kernels), and compare the uint8 results.   python tools/cross_check_sizes.py [HxW ...]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def worker(sizes, out):
    import torch
    import bench
    from instarevive_amd.pipeline import process
    log = lambda m: None
    swin, vae, dit, sched, sds = bench.build_models(torch.device("cuda", 0), log)
    y, mask = bench.synthetic_prompt()
    res = {}
    for s in sizes:
        h, w = (int(v) for v in s.split("x"))
        img = bench.synthetic_lq(1, h, w, 7)[0].numpy()
        pred, _ = process(dit, [img], 1, "wavelet", False, False, 512, 448, preprocess_model=swin, vae=vae, y=y.cuda(), y_mask=mask.cuda())
        res[s] = pred[0]
    np.savez(out, **res)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[3:], sys.argv[2])
        sys.exit(0)
    sizes = sys.argv[1:] or ["1088x1920", "1536x2048", "832x1216"]
    outs = []
    for tag, env in (("fast", {}), ("plain", {"IR_NO_CONV_PP": "1", "IR_NO_GEMM_PP": "1", "IR_NO_PINGPONG": "1"})):
        out = f"/tmp/cross_{tag}.npz"
        subprocess.run([sys.executable, __file__, "--worker", out] + sizes, check=True, env=dict(os.environ, **env))
        outs.append(np.load(out))
    for s in sizes:
        a, b = outs[0][s].astype(np.float64), outs[1][s].astype(np.float64)
        mse = ((a - b) ** 2).mean()
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        print(f"{s}: fast vs plain kernels PSNR {psnr:.2f} dB, max |diff| {int(np.abs(a - b).max())}, output std {a.std():.1f}", flush=True)
        assert psnr > 45.0 and a.std() > 1.0
    print("cross-check ok")
