"""Deterministic pseudo-random tensors shared by make_golden.py and the tests (numpy PCG64 uniform stream, which is
version-stable), so golden fixtures only need to store inputs, expected outputs and a weight checksum."""
import math

import numpy as np
import torch


def det_tensor(rng, shape, lo, hi):
    return torch.from_numpy((rng.random(size=shape, dtype=np.float64) * (hi - lo) + lo).astype(np.float32))


def det_state_dict(shapes, seed):
    """shapes: {name: shape}. Weights ~ U(+-sqrt(3/fan_in)) (unit-variance preserving), biases U(+-0.1),
    norm scales U(0.8,1.2), tables U(+-0.3). Iteration order = sorted(names) so it does not depend on dict order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if "table" in name or "embedding" in name:
            t = det_tensor(rng, shape, -0.3, 0.3)
        elif leaf == "bias" or len(shape) == 1 and leaf != "weight":
            t = det_tensor(rng, shape, -0.1, 0.1)
        elif len(shape) == 1:  # norm scale
            t = det_tensor(rng, shape, 0.8, 1.2)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = math.sqrt(3.0 / fan_in)
            t = det_tensor(rng, shape, -a, a)
        sd[name] = t
    return sd


def checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def det_input(seed, shape, lo=0.0, hi=1.0):
    return det_tensor(np.random.Generator(np.random.PCG64(seed)), shape, lo, hi)
