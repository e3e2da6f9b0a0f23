#!/usr/bin/env python3
"""Paired metrics of an output folder against a ground-truth folder: the PSNR-Y / SSIM-Y half of the reference's evaluate_img.py
(/root/reference/evaluate_img.py:30-33 creates them, :40-57 averages them over the sorted file lists).

    python tools/evaluate_pairs.py -i results/ -r gt/ [--ntest N]

The reference takes both from pyiqa (`create_metric('psnr' | 'ssim', test_y_channel=True, color_space='ycbcr')`). pyiqa is not in this image
and not in the reference tree, so its definitions are RESTATED here from the published implementation - parity unpinned until a box with
pyiqa runs tools/repin_with_diffusers.py-style checks:
  * Y = 16 + 65.481 R + 128.553 G + 24.966 B (BT.601, R, G, B in [0, 1]), rounded to integers (pyiqa.utils.color_util.to_y_channel at
    out_data_range 255);
  * PSNR = 10 log10(255^2 / (mean((Yx - Yy)^2) + 1e-8));
  * SSIM: 11 x 11 Gaussian window (sigma 1.5), 'valid' filtering, C1 = (0.01 * 255)^2, C2 = (0.03 * 255)^2, the contrast-structure term
    clamped at 0, mean over the map; no down-sampling (pyiqa's `ssim` default).
The no-reference metrics of the same script (MANIQA, MUSIQ, NIQE, CLIPIQA) and LPIPS need pretrained networks that do not exist offline
and are not restated. Files are paired by sorted order exactly as the reference does (glob "*.[jpJP][pnPN]*[gG]")."""
import argparse
import sys
from pathlib import Path

import numpy as np


def to_y(img_rgb01: np.ndarray) -> np.ndarray:
    """HWC float RGB in [0, 1] -> Y of YCbCr (BT.601 studio swing) on the 0..255 scale, rounded (fp64)."""
    x = np.asarray(img_rgb01, np.float64)
    return np.round(16.0 + 65.481 * x[..., 0] + 128.553 * x[..., 1] + 24.966 * x[..., 2])


def psnr_y(a_rgb01, b_rgb01) -> float:
    d = to_y(a_rgb01) - to_y(b_rgb01)
    return float(10.0 * np.log10(255.0 ** 2 / (np.mean(d * d) + 1e-8)))


def _gauss_window(size=11, sigma=1.5):
    c = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(c * c) / (2 * sigma * sigma))
    g /= g.sum()
    return g


def _filter_valid(x, g):
    """Separable 'valid' correlation with the 1-D window g along both axes."""
    k = len(g)
    h, w = x.shape
    tmp = np.zeros((h - k + 1, w), np.float64)
    for i in range(k):
        tmp += g[i] * x[i:i + h - k + 1]
    out = np.zeros((h - k + 1, w - k + 1), np.float64)
    for i in range(k):
        out += g[i] * tmp[:, i:i + w - k + 1]
    return out


def ssim_y(a_rgb01, b_rgb01) -> float:
    x, y = to_y(a_rgb01), to_y(b_rgb01)
    if x.shape != y.shape:
        raise ValueError(f"image shapes differ: {x.shape} vs {y.shape}")
    if min(x.shape) < 11:
        raise ValueError("SSIM needs images of at least 11 x 11 pixels")
    g = _gauss_window()
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    mu1, mu2 = _filter_valid(x, g), _filter_valid(y, g)
    s11 = _filter_valid(x * x, g) - mu1 * mu1
    s22 = _filter_valid(y * y, g) - mu2 * mu2
    s12 = _filter_valid(x * y, g) - mu1 * mu2
    cs = np.maximum((2 * s12 + c2) / (s11 + s22 + c2), 0.0)
    return float(np.mean((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1) * cs))


def list_images(folder):
    return sorted(Path(folder).glob("*.[jpJP][pnPN]*[gG]"))


def evaluate(in_path, ref_path, ntest=None, log=print):
    from PIL import Image
    ins, refs = list_images(in_path), list_images(ref_path)
    if ntest is not None:
        ins, refs = ins[:ntest], refs[:ntest]
    if not ins or len(ins) != len(refs):
        raise SystemExit(f"{len(ins)} images in {in_path}, {len(refs)} in {ref_path}: the folders must pair up (sorted order, as evaluate_img.py)")
    log(f"Find {len(ins)} images in {in_path}")
    tot = {"psnr": 0.0, "ssim": 0.0}
    for fi, fr in zip(ins, refs):
        a = np.asarray(Image.open(fi).convert("RGB"), np.float32) / 255.0
        b = np.asarray(Image.open(fr).convert("RGB"), np.float32) / 255.0
        if a.shape != b.shape:
            raise SystemExit(f"{fi.name} {a.shape} and {fr.name} {b.shape} differ in size")
        tot["psnr"] += psnr_y(a, b)
        tot["ssim"] += ssim_y(a, b)
    res = {k: v / len(ins) for k, v in tot.items()}
    for k, v in res.items():
        log(f"{k}: {v:.5f}")
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--in_path", type=str, required=True)
    ap.add_argument("-r", "--ref_path", type=str, required=True)
    ap.add_argument("--ntest", type=int, default=None)
    a = ap.parse_args()
    evaluate(a.in_path, a.ref_path, a.ntest)


if __name__ == "__main__":
    sys.exit(main())
