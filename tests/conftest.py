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


def linear_to_wave(cdf, K, bits=31, pad_rows_to=64):
    """Reference-layout cdf rows [D, >=K+1] (ANS.cdfs, mnist_compress.py:39-40) -> BS_LAYOUT_WAVE rows [D', K+64]
    following include/bitswap_hip.h (entry j at dword ((j/256)*64 + j%64)*4 + (j/64)%4, then the pivot words).
    The wave pop kernel wants whole 64-row chunks: rows are appended (they are popped FIRST, rows go D-1..0) that
    hold the one-symbol table f_0 = 2^bits, whose pop leaves head and stack untouched."""
    cdf = np.asarray(cdf).astype(np.uint32)
    D = cdf.shape[0]
    Dp = (D + pad_rows_to - 1) // pad_rows_to * pad_rows_to
    lin = np.full((Dp, K + 1), 1 << bits, dtype=np.uint32)
    lin[:, 0] = 0
    lin[:D] = cdf[:, : K + 1]
    j = np.arange(K)
    off = ((j // 256) * 64 + j % 64) * 4 + (j // 64) % 4
    out = np.full((Dp, K + 64), 0xffffffff, dtype=np.uint32)
    out[:, off] = lin[:, :K]
    nr = K // 64
    out[:, K: K + nr] = lin[:, 0:K:64]
    out[:, K + nr] = 1 << bits
    return out


# seeds of tests/golden/make_golden.py::_full_chain (torch.manual_seed before the model is built)
FULL_CHAIN_SEEDS = {"mnist": 55, "cifar": 56, "imagenet": 57}


def seeded_full_model(g, data, device="cpu", **kw):
    """The reference Model of a chain_<data>_full fixture rebuilt from its SEED instead of a stored state dict (179 MB at the
    cifar8 width): bitswap_amd.model.Model draws its parameters with the reference's calls in the reference's order, so
    torch.manual_seed(s) + the same perturbation of biases, gains and gen_std as make_golden.py gives the reference's tensors
    -- tests/test_host_cpu.py holds that against the imported reference where it is present; a wrong weight would also show as
    (mu, scale) far from the fixture's in the tests that use this."""
    import torch
    from bitswap_amd.model import Model
    cfg = g["cfg"]
    torch.manual_seed(FULL_CHAIN_SEEDS[data])
    m = Model(xs=(int(cfg[0]), 32, 32), nz=int(cfg[1]), zchannels=int(cfg[2]), nprocessing=int(cfg[3]),
              kernel_size=int(cfg[4]), resdepth=int(cfg[5]), reswidth=int(cfg[6]), root_process=False, **kw)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith(".b") or n.endswith("gen_std"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    return m.to(device).eval()


def load_golden_model(g, device="cpu", **kw):
    """bitswap_amd Model with the weights of a reference-generated model fixture (tests/golden/model_*.npz)."""
    import torch
    from bitswap_amd.model import Model
    cfg = g["cfg"]
    m = Model(xs=(int(cfg[0]), 32, 32), nz=int(cfg[1]), zchannels=int(cfg[2]), nprocessing=int(cfg[3]),
              kernel_size=int(cfg[4]), resdepth=int(cfg[5]), reswidth=int(cfg[6]), **kw)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    assert set(sd) == set(m.state_dict())          # identical state-dict keys as the reference Model
    m.load_state_dict(sd)
    return m.to(device).eval()
