"""north_star's acceptance form, made to bite: |PSNR(ours, GT) - PSNR(reference, GT)| <= 0.1 dB only constrains `ours` when the reference
itself scores in the range restorers reach (25-35 dB); against an unrelated "GT" (10 dB) a 47 dB perturbation moves nothing. With no real
ground truth offline (random weights), synthetic ground truths are built AROUND the oracle's output:

    GT = oracle + n,  |n| chosen so that PSNR(oracle, GT) = level            (PSNR as utils/metrics.py:8-38: fp64 MSE over [0, 1])

* independent noise (n white, seeded): MSE(ours, GT) = |e|^2 + |n|^2 - 2<e, n> with e = ours - oracle and <e, n> ~ 0, so
  dPSNR ~ 10 log10(1 + 10^((level - P_err) / 10)), P_err = PSNR(ours, oracle); it crosses 0.1 dB at level = P_err - 16.33 dB.
* worst case (n anti-parallel to e: the ground truth lies exactly on the far side of the oracle): dPSNR = 20 log10(1 + 10^((level - P_err) / 20)),
  0.1 dB at level = P_err - 38.7 dB - no finite-precision path meets that at 30 dB; reported, not asserted.
Test infrastructure (used by tests/ only)."""
import numpy as np

CROSS_INDEPENDENT_DB = -10.0 * np.log10(10 ** 0.01 - 1.0)    # 16.33: level = P_err - this  <=>  dPSNR = 0.1 dB
CROSS_WORST_DB = -20.0 * np.log10(10 ** 0.005 - 1.0)         # 38.7


def psnr01(a, b):
    """a, b: float arrays in [0, 1] units (any shape). 10 log10(1 / (mse + 1e-8)) on fp64, as utils/metrics.py."""
    mse = float(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean())
    return 10.0 * np.log10(1.0 / (mse + 1e-8))


def guard_table(ours_u8, oracle_u8, levels=(25.0, 30.0, 35.0), seed=7):
    """-> (P_err, rows): rows[level] = (d_independent, d_worst) in dB for uint8 images ours / oracle of equal shape."""
    o = np.asarray(oracle_u8, np.float64) / 255.0
    u = np.asarray(ours_u8, np.float64) / 255.0
    e = u - o
    p_err = psnr01(u, o)
    rng = np.random.Generator(np.random.PCG64(seed))
    white = rng.standard_normal(o.shape)
    white /= np.sqrt((white ** 2).mean())
    en = np.sqrt((e ** 2).mean())
    rows = {}
    for lv in levels:
        sigma = 10.0 ** (-lv / 20.0)
        gt_i = o + sigma * white
        d_i = abs(psnr01(u, gt_i) - psnr01(o, gt_i))
        d_w = float("nan")
        if en > 0:
            gt_w = o - (sigma / en) * e
            d_w = abs(psnr01(u, gt_w) - psnr01(o, gt_w))
        rows[lv] = (d_i, d_w)
    return p_err, rows


def crossing_level(p_err_db):
    """Reference quality (dB against the ground truth) up to which an error of p_err_db against the reference stays within 0.1 dB
    (independent error)."""
    return p_err_db - CROSS_INDEPENDENT_DB
