import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from instarevive_amd import Context
    return Context(0)


@pytest.fixture(scope="session")
def full_models():
    """Full-size architectures (SwinIR 15.8 M, VAE 83.7 M, DiT 611 M parameters, 300 x 4096 prompt) with bench.py's seeded random
    weights, uploaded once per test session: (swin, vae, dit, state_dicts, y, mask)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    swin, vae, dit, _sched, sds = bench.build_models(torch.device("cuda", 0), lambda m: None)
    y, mask = bench.synthetic_prompt()

    class Full(tuple):  # unpacks like the old 6-tuple; also carries ONE device copy of the prompt (a stable object keeps the prompt cache
        pass            # and recorded hipGraphs valid from call to call)

    full = Full((swin, vae, dit, sds, y, mask))
    full.y_cuda, full.mask_cuda = y.cuda(), mask.cuda()
    return full
