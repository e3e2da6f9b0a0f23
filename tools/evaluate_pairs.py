#!/usr/bin/env python3
"""Paired metrics of an output folder against a ground-truth folder: the PSNR-Y / SSIM-Y / LPIPS half of the reference's evaluate_img.py
(/root/reference/evaluate_img.py:30-33 creates them, :40-57 averages them over the sorted file lists).

    python tools/evaluate_pairs.py -i results/ -r gt/ [--ntest N] [--lpips_alexnet alexnet-owt-7be5be79.pth --lpips_lin alex.pth]

The reference takes all three from pyiqa (`create_metric('psnr' | 'ssim', test_y_channel=True, color_space='ycbcr')`, `create_metric('lpips')`);
its training code wraps the `lpips` package itself (utils/metrics.py:41-66, `LPIPS(net="alex")`). Neither package is in this image or in the
reference tree, so their definitions are RESTATED here from the published implementations - parity unpinned until a box with pyiqa / lpips
runs tools/repin_with_diffusers.py-style checks:
  * Y = (16 + 65.481 R + 128.553 G + 24.966 B) / 255 (BT.601 studio swing; R, G, B in [0, 1]);
  * PSNR (pyiqa `psnr`, data_range 1 - its default, which evaluate_img.py does not override): Y on the UNIT scale, no rounding,
    10 log10(1 / (mean((Yx - Yy)^2) + 1e-8));
  * SSIM (pyiqa `ssim` works on the 0..255 scale whatever the input range): Y x 255 ROUNDED to integers, 11 x 11 Gaussian window (sigma 1.5),
    'valid' filtering, C1 = (0.01 * 255)^2, C2 = (0.03 * 255)^2, the contrast-structure term clamped at 0, mean over the map; no down-sampling;
  * LPIPS v0.1, net 'alex' (the default of both packages): inputs [0, 1] -> [-1, 1] (normalize=True), the scaling layer
    (x - (-.030, -.088, -.188)) / (.458, .448, .450), the five ReLU outputs of torchvision's AlexNet `features` (conv 11/4 pad 2 -> pool 3/2 ->
    conv 5 pad 2 -> pool 3/2 -> conv 3 -> conv 3 -> conv 3), each unit-normalised over channels (x / (|x|_2 + 1e-10)), squared differences
    weighted by a non-negative 1 x 1 "lin" layer per stage, averaged over space, summed over the five stages.
    The pretrained weights (torchvision's alexnet-owt-7be5be79.pth, 233 MB, and lpips/weights/v0.1/alex.pth, 6 KB) do not exist offline:
    LPIPS is computed only when the user passes them (--lpips_alexnet / --lpips_lin, or one full `lpips.LPIPS().state_dict()` file through
    --lpips_lin alone); the arithmetic is tested against an independent torch.nn construction on random weights (tests/test_host_cpu.py).
The no-reference metrics of the same script (MANIQA, MUSIQ, NIQE, CLIPIQA: pretrained networks) are out of scope.
Files are paired by sorted order exactly as the reference does (glob "*.[jpJP][pnPN]*[gG]")."""
import argparse
import sys
from pathlib import Path

import numpy as np


def to_y(img_rgb01: np.ndarray, data_range: float = 255.0) -> np.ndarray:
    """HWC float RGB in [0, 1] -> Y of YCbCr (BT.601 studio swing) on the 0..data_range scale (fp64); rounded to integers on the 255
    scale only (pyiqa.utils.color_util.to_y_channel rounds when out_data_range >= 255 and not otherwise)."""
    x = np.asarray(img_rgb01, np.float64)
    y = (16.0 + 65.481 * x[..., 0] + 128.553 * x[..., 1] + 24.966 * x[..., 2]) / 255.0 * data_range
    return np.round(y) if data_range >= 255 else y


def psnr_y(a_rgb01, b_rgb01) -> float:
    """pyiqa `psnr(test_y_channel=True, color_space='ycbcr')` at its default data_range 1: unit-scale Y, no rounding, eps on the unit MSE."""
    d = to_y(a_rgb01, 1.0) - to_y(b_rgb01, 1.0)
    return float(10.0 * np.log10(1.0 / (np.mean(d * d) + 1e-8)))


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


# ---------------------------------------------------------------------------------------------------------------- LPIPS (v0.1, alex)
ALEX_CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))  # features idx, cin, cout, k, stride, pad
ALEX_POOL_BEFORE = (False, True, True, False, False)   # MaxPool2d(3, 2) in front of conv2 and conv3 (torchvision features.2 / features.5)
# full-model key of each backbone conv in lpips.LPIPS().state_dict(): the slices keep torchvision's indices
LPIPS_SLICE_KEY = ("net.slice1.0", "net.slice2.3", "net.slice3.6", "net.slice4.8", "net.slice5.10")


class LPIPS:
    """lpips.LPIPS(net='alex', version='0.1') in eval mode, restated (utils/metrics.py:41-66 wraps it; evaluate_img.py:32 takes pyiqa's copy of
    the same network). Weights come from the user's files: `alexnet` = torchvision's AlexNet state dict (`features.N.weight|bias`) and `lin` = the
    lpips linear heads (`lin{k}.model.1.weight`, [1, C, 1, 1]); or `lin` alone holding a full lpips state dict (`net.slice*.N.*` + `lin*`)."""

    def __init__(self, alexnet=None, lin=None, device="cpu"):
        import torch
        sd = {}
        for src in (alexnet, lin):
            if src is None:
                continue
            part = torch.load(src, map_location="cpu") if isinstance(src, (str, Path)) else src
            sd.update(part.get("state_dict", part) if isinstance(part, dict) else part)
        self.convs, self.lins = [], []
        for k, (idx, cin, cout, ks, _, _) in enumerate(ALEX_CONVS):
            for base in (f"features.{idx}", LPIPS_SLICE_KEY[k]):
                if base + ".weight" in sd:
                    w, b = sd[base + ".weight"], sd[base + ".bias"]
                    break
            else:
                raise KeyError(f"AlexNet conv {k + 1}: neither features.{idx}.weight nor {LPIPS_SLICE_KEY[k]}.weight in the given files")
            lw = sd.get(f"lin{k}.model.1.weight")
            if lw is None:
                raise KeyError(f"lin{k}.model.1.weight missing: pass the lpips linear heads (lpips/weights/v0.1/alex.pth)")
            if tuple(w.shape) != (cout, cin, ks, ks) or tuple(lw.shape) != (1, cout, 1, 1):
                raise ValueError(f"stage {k}: conv {tuple(w.shape)} / lin {tuple(lw.shape)} are not AlexNet's")
            self.convs.append((w.to(device, torch.float32), b.to(device, torch.float32)))
            self.lins.append(lw.to(device, torch.float32))
        self.shift = torch.tensor([-.030, -.088, -.188], device=device).view(1, 3, 1, 1)
        self.scale = torch.tensor([.458, .448, .450], device=device).view(1, 3, 1, 1)
        self.device = device

    def features(self, x):
        import torch.nn.functional as F
        outs = []
        for (w, b), (_, _, _, _, stride, pad), pool in zip(self.convs, ALEX_CONVS, ALEX_POOL_BEFORE):
            if pool:
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(F.conv2d(x, w, b, stride=stride, padding=pad))
            outs.append(x)
        return outs

    def __call__(self, img1, img2, normalize=True):
        """img1, img2: NCHW RGB tensors, [0, 1] with normalize=True (pyiqa's and evaluate_img.py's use), [-1, 1] otherwise -> [N] distances."""
        import torch
        with torch.no_grad():
            a, b = (t.to(self.device, torch.float32) for t in (img1, img2))
            if normalize:
                a, b = 2 * a - 1, 2 * b - 1
            fa, fb = self.features((a - self.shift) / self.scale), self.features((b - self.shift) / self.scale)
            total = 0
            for xa, xb, lw in zip(fa, fb, self.lins):
                na = xa / (xa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                nb = xb / (xb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                total = total + ((na - nb) ** 2 * lw).sum(1, keepdim=True).mean((2, 3), keepdim=True)
            return total.reshape(-1)


def list_images(folder):
    return sorted(Path(folder).glob("*.[jpJP][pnPN]*[gG]"))


def evaluate(in_path, ref_path, ntest=None, log=print, lpips=None):
    from PIL import Image
    ins, refs = list_images(in_path), list_images(ref_path)
    if ntest is not None:
        ins, refs = ins[:ntest], refs[:ntest]
    if not ins or len(ins) != len(refs):
        raise SystemExit(f"{len(ins)} images in {in_path}, {len(refs)} in {ref_path}: the folders must pair up (sorted order, as evaluate_img.py)")
    log(f"Find {len(ins)} images in {in_path}")
    tot = {"psnr": 0.0, "ssim": 0.0}
    if lpips is not None:
        import torch
        tot["lpips"] = 0.0
    for fi, fr in zip(ins, refs):
        a = np.asarray(Image.open(fi).convert("RGB"), np.float32) / 255.0
        b = np.asarray(Image.open(fr).convert("RGB"), np.float32) / 255.0
        if a.shape != b.shape:
            raise SystemExit(f"{fi.name} {a.shape} and {fr.name} {b.shape} differ in size")
        tot["psnr"] += psnr_y(a, b)
        tot["ssim"] += ssim_y(a, b)
        if lpips is not None:
            tot["lpips"] += float(lpips(torch.from_numpy(a).permute(2, 0, 1)[None], torch.from_numpy(b).permute(2, 0, 1)[None], normalize=True)[0])
    res = {k: v / len(ins) for k, v in tot.items()}
    for k, v in res.items():
        log(f"{k}: {v:.5f}")
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--in_path", type=str, required=True)
    ap.add_argument("-r", "--ref_path", type=str, required=True)
    ap.add_argument("--ntest", type=int, default=None)
    ap.add_argument("--lpips_alexnet", type=str, default=None, help="torchvision AlexNet state dict (alexnet-owt-7be5be79.pth)")
    ap.add_argument("--lpips_lin", type=str, default=None, help="lpips v0.1 linear heads (lpips/weights/v0.1/alex.pth), or a full lpips.LPIPS() state dict")
    ap.add_argument("--device", type=str, default="cpu")
    a = ap.parse_args()
    net = LPIPS(a.lpips_alexnet, a.lpips_lin, a.device) if a.lpips_lin else None
    if net is None:
        print("lpips: skipped (no weights given: --lpips_lin [--lpips_alexnet])")
    evaluate(a.in_path, a.ref_path, a.ntest, lpips=net)


if __name__ == "__main__":
    sys.exit(main())
