import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def words_to_state(w):
    """fixture layout [stack words..., head_lo, head_hi] -> Python list with 64-bit head last."""
    return [int(x) for x in w[:-2]] + [int(w[-2]) | (int(w[-1]) << 32)]


def reference_init_state(n=10000, seed=100):
    """mnist_compress.py:158-159 after np.random.seed(100) (:94)."""
    np.random.seed(seed)
    s = list(map(int, np.random.randint(low=1 << 16, high=(1 << 32) - 1, size=n, dtype=np.uint32)))
    s[-1] = s[-1] << 32
    return s


def chain_tables(g):
    """Rebuild the float64 bin endpoints/centres of a chain fixture from its compact form
    (make_golden.py asserts that this reproduces the reference arrays exactly)."""
    cfg = g["cfg"]
    nz, q = int(cfg[1]), int(cfg[7])
    K = 1 << q
    D = g["z_mins"].shape[1]
    zend = np.zeros((nz, D, K - 1))
    zcen = np.zeros((nz, D, K))
    zend[nz - 1] = g["z_top_endpoints"][None]
    zcen[nz - 1] = g["z_top_centres"][None]
    for zi in range(nz - 1):
        edges = np.stack([np.linspace(a, b, K + 1) for a, b in zip(g["z_mins"][zi], g["z_maxs"][zi])])
        zend[zi] = edges[:, 1:-1]
        zcen[zi] = (edges[:, :-1] + edges[:, 1:]) / 2
    X = int(cfg[0]) * 1024
    xend = np.broadcast_to(((np.arange(1, 256) - 127.5) / 127.5 - 1. / 255.)[None], (X, 255))
    return zend, xend, zcen
