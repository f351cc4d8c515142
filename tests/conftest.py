import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dev():
    assert torch.cuda.is_available()
    # the product path must be the HIP library, never a fallback
    from nerfacc_amd import _backend as B
    B.load()
    return torch.device("cuda:0")


def seeded_case(params):
    """Regenerate the inputs of a seeded traversal fixture (oracle/gen_golden.py: seeded_case)."""
    seed, R, res, occ, step, cone, G, near, inside = params
    seed, R, res, G = int(seed), int(R), int(res), int(G)
    r = np.random.default_rng(seed)
    o = r.standard_normal((R, 3)).astype(np.float32)
    if inside:
        o = (r.random((R, 3)).astype(np.float32) - 0.5)
    d = r.standard_normal((R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    b = r.random((G, res, res, res)) < occ
    base = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    ab = np.stack([np.concatenate([-(2.0 ** i) * np.ones(3), (2.0 ** i) * np.ones(3)]) for i in range(G)]).astype(np.float32)
    del base
    nearp = np.full(R, near, np.float32)
    return o, d, b, ab, nearp, float(step), float(cone)


def assert_close(a, b, atol=1e-6, rtol=1e-5, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = atol + rtol * np.abs(b.astype(np.float64))
    assert (err <= tol).all(), (what, "max abs err", float(err.max()), "worst err/tol", float((err / np.maximum(tol, 1e-300)).max()))
