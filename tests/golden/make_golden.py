#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE code (fhkingma/bitswap) in-process.

Run in the build container only (needs /root/reference; the GPU box has no copy):

    python tests/golden/make_golden.py

The reference ships no tests, golden vectors or checkpoints, so
the fixtures are outputs of the reference's own classes on seeded synthetic inputs:

  tables.npz      logistic_cdf (utils/torch/rand.py:67-68) + pmf assembly
                  (mnist_compress.py:183-185) + ANS.__init__ integer tables (:14-47)
  rans.npz        ANS.decode / ANS.encode (:49-68) word streams on those tables
  model_*.npz     reference Model.infer(i)/generate(i) outputs (model/mnist_train.py:315-438)
  chain_*.npz     sender (mnist_compress.py:176-251) and receiver (:284-354) loops replayed
                  around the imported ANS/Model/logistic_cdf on device "cpu", with every
                  (mu, scale, symbols, state) captured per coding operation
  bins.npz        Bins / ImageBins / discretize_kbins outputs (rand.py:78-153,
                  discretization.py:105-118)
  demo_surface.npz   tiling, demo container (.npy), pickle: outputs of the reference's own extract_blocks,
                  demo_compress.compress and demo_decompress.decompress functions (see make_surface_fixture)
  discretize_small.npz  the sampling procedure of discretize() with explicit seeds (see make_discretize_fixture)
  chain_{mnist,cifar,imagenet}_full_*.npz   the reference's sender at the BASELINE models' REAL width (configs[0]: 100 blocks;
                  configs[1]: 6 blocks; configs[2]/[4]: 4 blocks), compact per-op layout; the weights are not stored -- they are
                  the seeded default init + a seeded perturbation (tests/conftest.py::seeded_full_model rebuilds them)

The reference is imported unmodified; torchvision and tensorboardX (absent here) are
stubbed exactly as SURVEY.md section 8(c) describes.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("BITSWAP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_missing_modules():
    tv = types.ModuleType("torchvision")
    tv.__all__ = []
    for n in ("datasets", "transforms", "utils"):
        m = types.ModuleType("torchvision." + n)
        setattr(tv, n, m)
        sys.modules["torchvision." + n] = m
    sys.modules["torchvision"] = tv
    tb = types.ModuleType("tensorboardX")
    tb.SummaryWriter = object
    sys.modules["tensorboardX"] = tb
    # demo_compress.py:4,13,100: torchvision.transforms.Compose / ToTensor and terminaltables.AsciiTable (absent here).
    # ToTensor's documented behaviour on a uint8 HWC PIL image: CHW float32 in [0, 1].
    class Compose:
        def __init__(self, ops):
            self.ops = ops

        def __call__(self, x):
            for op in self.ops:
                x = op(x)
            return x

    class ToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.asarray(pic).copy()).permute(2, 0, 1).contiguous().float().div(255)
    tv.transforms.Compose, tv.transforms.ToTensor = Compose, ToTensor
    tt = types.ModuleType("terminaltables")
    tt.AsciiTable = object
    sys.modules["terminaltables"] = tt


_stub_missing_modules()
sys.path.insert(0, REF)
import mnist_compress as ref_mc  # noqa: E402
from discretization import discretize_kbins  # noqa: E402
from model.mnist_train import Model as RefModel  # noqa: E402
from model.imagenetcrop_train import Model as RefCropModel  # noqa: E402
from utils.torch.rand import Bins, ImageBins, logistic_cdf, logistic_eps, transform  # noqa: E402

ANS = ref_mc.ANS
F64 = torch.float64


def ref_pmfs(endpoints, mu, scale):
    """mnist_compress.py:183-185 verbatim."""
    cdfs = logistic_cdf(endpoints.t(), mu, scale).t()
    pmfs = cdfs[:, 1:] - cdfs[:, :-1]
    pmfs = torch.cat((cdfs[:, 0].unsqueeze(1), pmfs, 1. - cdfs[:, -1].unsqueeze(1)), dim=1)
    return cdfs, pmfs


def init_state(n=10000):
    """mnist_compress.py:158-159 (caller seeds numpy)."""
    state = list(map(int, np.random.randint(low=1 << 16, high=(1 << 32) - 1, size=n, dtype=np.uint32)))
    state[-1] = state[-1] << 32
    return state


def words(state):
    """Python-int state list -> uint32 words + (head_lo, head_hi)."""
    head = state[-1]
    return np.array(state[:-1] + [head & 0xFFFFFFFF, head >> 32], dtype=np.uint64).astype(np.uint32)


# --------------------------------------------------------------------------- tables + rans
def make_tables_and_rans():
    rng = np.random.RandomState(1234)
    out = {}
    cases = {}
    # latent op: K = 1024 equal-mass top-layer bins (discretization.py:25-27) and uniform bins
    D = 24
    top = Bins(torch.zeros((1, 1, D)), torch.ones((1, 1, D)), 10).endpoints()[0, 0].to(F64)
    lo = rng.uniform(-9, -3, size=D)
    hi = rng.uniform(3, 9, size=D)
    uni = torch.from_numpy(np.stack([np.linspace(a, b, 1025)[1:-1] for a, b in zip(lo, hi)]))
    mu = torch.from_numpy((rng.randn(D) * 0.7).astype(np.float32)).to(F64)
    sc = torch.from_numpy(rng.uniform(0.1, 1.0, size=D).astype(np.float32)).to(F64)
    # edge rows: far-off mean (saturated tails), minimum scale, exact ties at the mode
    mu[0], sc[0] = 30.0, 0.1
    mu[1], sc[1] = -30.0, 0.1
    mu[2], sc[2] = 0.0, 1.0
    mu[3], sc[3] = 0.0, 0.1
    cases["ztop"] = (top, mu, sc, 10)
    cases["zuni"] = (uni, mu.clone(), sc.clone(), 10)
    # pixel op: K = 256 (rand.py:134-153), quantbits = 8 (mnist_compress.py:203)
    Dx = 48
    xe = ImageBins(F64, "cpu", Dx).endpoints()
    mux = torch.from_numpy(rng.uniform(-1.2, 1.2, size=Dx).astype(np.float32)).to(F64)
    scx = torch.from_numpy(rng.choice([2. / 255. / 8., 0.02, 0.1, 0.7], size=Dx).astype(np.float32)).to(F64)
    cases["x"] = (xe, mux, scx, 8)

    np.random.seed(100)
    for name, (e, m, s, q) in cases.items():
        cdfs, pmfs = ref_pmfs(e, m, s)
        a = ANS(pmfs, bits=31, quantbits=q)
        out[f"{name}_endpoints"] = e.numpy()
        out[f"{name}_mu"] = m.numpy()
        out[f"{name}_scale"] = s.numpy()
        out[f"{name}_quantbits"] = np.int32(q)
        out[f"{name}_cdf_f64"] = cdfs.numpy()
        out[f"{name}_pmf_f64"] = pmfs.numpy()
        out[f"{name}_f"] = a.pmfs.astype(np.uint32)
        out[f"{name}_cdf"] = a.cdfs.astype(np.uint32)
        # rANS: pop D symbols, push them back, push fresh symbols
        st = init_state(200)
        out[f"{name}_state0"] = words(st)
        st, sym = a.decode(st)
        out[f"{name}_pop_sym"] = sym.numpy().astype(np.int32)
        out[f"{name}_state_after_pop"] = words(st)
        st = a.encode(st, sym)
        assert words(st).tolist() == out[f"{name}_state0"].tolist()
        fresh = torch.from_numpy(rng.randint(0, pmfs.shape[1], size=pmfs.shape[0]))
        st = a.encode(st, fresh)
        out[f"{name}_push_sym"] = fresh.numpy().astype(np.int32)
        out[f"{name}_state_after_push"] = words(st)
    # a hand-made pmf table with exact ties for the argmax rule (mnist_compress.py:36)
    tie = np.full((4, 16), 1.0 / 16)
    tie[1, 3] = tie[1, 9] = 0.2
    tie[1] /= tie[1].sum()
    tie[2] = 0.0
    tie[2, 15] = 1.0
    tie[3] = np.linspace(1, 16, 16)
    tie[3] /= tie[3].sum()
    a = ANS(torch.from_numpy(tie), bits=31, quantbits=4)
    out["tie_pmf_f64"] = tie
    out["tie_f"] = a.pmfs.astype(np.uint32)
    out["tie_cdf"] = a.cdfs.astype(np.uint32)
    np.savez_compressed(os.path.join(OUT, "tables_rans.npz"), **out)
    print("tables_rans.npz", {k: v.shape for k, v in out.items() if k.endswith("_f")})


# --------------------------------------------------------------------------- bins
def make_bins():
    out = {}
    b = Bins(torch.zeros((1, 1, 8)), torch.ones((1, 1, 8)), 10)
    out["top_endpoints_q10"] = b.endpoints().numpy()[0, 0, 0]   # identical for every dim
    out["top_centres_q10"] = b.centres().numpy()[0, 0, 0]
    b = Bins(torch.zeros((1, 1, 8)), torch.ones((1, 1, 8)), 8)
    out["top_endpoints_q8"] = b.endpoints().numpy()[0, 0, 0]
    out["top_centres_q8"] = b.centres().numpy()[0, 0, 0]
    ib = ImageBins(F64, "cpu", 5)
    out["x_endpoints"] = ib.endpoints().numpy()[0]
    out["x_centres"] = ib.centres().numpy()[0]
    rng = np.random.RandomState(7)
    samples = rng.randn(500, 2, 4, 4).astype(np.float16).astype(np.float64)  # float64 cast: SURVEY 7(f)

    class M:
        zdim = (2, 4, 4)
    e, c = discretize_kbins(M, samples, 6, strategy="uniform")
    out["kbins_samples"] = samples
    out["kbins_endpoints_q6"] = e
    out["kbins_centres_q6"] = c
    np.savez_compressed(os.path.join(OUT, "bins.npz"), **out)
    print("bins.npz")


# --------------------------------------------------------------------------- model + chains
def synth_images(rng, n, xs):
    """Smooth synthetic blocks: 8x8 Gaussian field upsampled + noise."""
    c, h, w = xs
    base = rng.randn(n, c, 8, 8).astype(np.float32)
    up = torch.nn.functional.interpolate(torch.from_numpy(base), size=(h, w), mode="bilinear", align_corners=False)
    img = 127.5 + 60.0 * up.numpy() + rng.randn(n, c, h, w) * 4.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_bins(model, nz, zdim_flat, quantbits, images, rng):
    """discretize() (discretization.py:9-99) with a synthetic 'dataset' and ppb reduced to 2."""
    K = 1 << quantbits
    zend = np.zeros((nz, zdim_flat, K - 1))
    zcen = np.zeros((nz, zdim_flat, K))
    zb = Bins(torch.zeros((1, 1, zdim_flat)), torch.ones((1, 1, zdim_flat)), quantbits)
    zend[nz - 1] = zb.endpoints().numpy()
    zcen[nz - 1] = zb.centres().numpy()
    nsamples = 2 * K
    bs = 128
    batches = nsamples // bs
    data = torch.from_numpy(images[rng.randint(0, len(images), size=nsamples)]).float()
    gen = np.zeros((nz, nsamples) + model.zdim, dtype=np.float16)
    gen[-1] = logistic_eps((nsamples,) + model.zdim, device="cpu", bound=1e-30).numpy()
    inf = np.zeros((nz, nsamples) + model.zdim, dtype=np.float16)
    with torch.no_grad():
        for zi in reversed(range(1, nz)):
            for bi in range(batches):
                mu, scale = model.generate(zi)(given=torch.from_numpy(gen[zi][bi * bs: bi * bs + bs]).float())
                gen[zi - 1][bi * bs: bi * bs + bs] = transform(logistic_eps(mu.shape, device="cpu", bound=1e-30), mu, scale)
            for bi in range(batches):
                given = (data[bi * bs: bi * bs + bs] if nz - zi - 1 == 0
                         else torch.from_numpy(inf[nz - zi - 2][bi * bs: bi * bs + bs]).float())
                mu, scale = model.infer(nz - zi - 1)(given=given)
                inf[nz - zi - 1][bi * bs: bi * bs + bs] = transform(logistic_eps(mu.shape, device="cpu", bound=1e-30), mu, scale).numpy()
    mins, maxs = [], []
    for zi in range(nz - 1):
        samples = np.concatenate([gen[zi], inf[zi]], axis=0).astype(np.float64)
        zend[zi], zcen[zi] = discretize_kbins(model, samples, quantbits, strategy="uniform")
        flat = samples.reshape(-1, zdim_flat)
        mins.append(flat.min(0))
        maxs.append(flat.max(0))
        # compact form must regenerate the reference arrays exactly (checked here, relied on by tests)
        edges = np.stack([np.linspace(a, b, K + 1) for a, b in zip(mins[-1], maxs[-1])])
        assert np.array_equal(edges[:, 1:-1], zend[zi])
        assert np.array_equal((edges[:, :-1] + edges[:, 1:]) / 2, zcen[zi])
    return zend, zcen, np.array(mins), np.array(maxs)


def replay_chain(model, zend, zcen, images, nz, bitswap, quantbits, xdim, zdim, cap):
    """Sender mnist_compress.py:164-263 and receiver :277-358 around the imported ANS/Model."""
    zendpoints, zcentres = torch.from_numpy(zend), torch.from_numpy(zcen)
    xbins = ImageBins(F64, "cpu", xdim)
    xendpoints, xcentres = xbins.endpoints(), xbins.centres()
    zrange, xrange = torch.arange(zdim), torch.arange(xdim)
    ansbits = 31
    model.compress()
    ops = []  # (kind, layer, quantbits, mu, scale, sym, words_after)

    def rec(kind, table, q, mu, scale, sym, state):
        ops.append(dict(kind=kind, table=table, q=q, mu=mu.numpy().astype(np.float32), scale=scale.numpy().astype(np.float32),
                        sym=np.asarray(sym, dtype=np.int32), nwords=len(state), head=state[-1]))

    np.random.seed(100)
    state = init_state(10000)
    initialstate = state.copy()
    restbits = None
    datapoints = [torch.from_numpy(im.astype(np.float32)).view(xdim) for im in images]
    nets, cma = [], []
    with torch.no_grad():
        for xi, x in enumerate(datapoints):
            if bitswap:
                for zi in range(nz):
                    input = zcentres[zi - 1, zrange, zsym] if zi > 0 else xcentres[xrange, x.long()]
                    mu, scale = model.infer(zi)(given=input)
                    _, pmfs = ref_pmfs(zendpoints[zi], mu, scale)
                    state, zsymtop = ANS(pmfs, bits=ansbits, quantbits=quantbits).decode(state)
                    rec(0, zi, quantbits, mu, scale, zsymtop, state)
                    if xi == zi == 0:
                        restbits = state.copy()
                        assert len(restbits) > 1
                    z = zcentres[zi, zrange, zsymtop]
                    mu, scale = model.generate(zi)(given=z)
                    _, pmfs = ref_pmfs(zendpoints[zi - 1] if zi > 0 else xendpoints, mu, scale)
                    sym = zsym if zi > 0 else x.long()
                    state = ANS(pmfs, bits=ansbits, quantbits=(quantbits if zi > 0 else 8)).encode(state, sym)
                    rec(1, zi - 1, quantbits if zi > 0 else 8, mu, scale, sym, state)
                    zsym = zsymtop
            else:
                zs = []
                for zi in range(nz):
                    input = zcentres[zi - 1, zrange, zsym] if zi > 0 else xcentres[xrange, x.long()]
                    mu, scale = model.infer(zi)(given=input)
                    _, pmfs = ref_pmfs(zendpoints[zi], mu, scale)
                    state, zsymtop = ANS(pmfs, bits=ansbits, quantbits=quantbits).decode(state)
                    rec(0, zi, quantbits, mu, scale, zsymtop, state)
                    zs.append(zsymtop)
                    zsym = zsymtop
                if xi == 0:
                    restbits = state.copy()
                    assert len(restbits) > 1
                for zi in range(nz):
                    zsymtop = zs.pop(0)
                    z = zcentres[zi, zrange, zsymtop]
                    mu, scale = model.generate(zi)(given=z)
                    _, pmfs = ref_pmfs(zendpoints[zi - 1] if zi > 0 else xendpoints, mu, scale)
                    sym = zsym if zi > 0 else x.long()
                    state = ANS(pmfs, bits=ansbits, quantbits=(quantbits if zi > 0 else 8)).encode(state, sym)
                    rec(1, zi - 1, quantbits if zi > 0 else 8, mu, scale, sym, state)
                    zsym = zsymtop
                assert zs == []
            _, pmfs = ref_pmfs(zendpoints[-1], torch.zeros(1, dtype=F64), torch.ones(1, dtype=F64))
            state = ANS(pmfs, bits=ansbits, quantbits=quantbits).encode(state, zsymtop)
            rec(1, nz - 1, quantbits, torch.zeros(zdim), torch.ones(zdim), zsymtop, state)
            totaladdedbits = (len(state) - len(initialstate)) * 32
            totalbits = (len(state) - (len(restbits) - 1)) * 32
            nets.append((totaladdedbits / xdim) - sum(nets))
            cma.append(totalbits / (xdim * (xi + 1)))
        sent = state.copy()

        # receiver (:277-358), decoded images and state unwinding asserted as the reference does
        for xi, x in enumerate(reversed(datapoints)):
            _, pmfs = ref_pmfs(zendpoints[-1], torch.zeros(1, dtype=F64), torch.ones(1, dtype=F64))
            state, zsymtop = ANS(pmfs, bits=ansbits, quantbits=quantbits).decode(state)
            if bitswap:
                for zi in reversed(range(nz)):
                    z = zcentres[zi, zrange, zsymtop]
                    mu, scale = model.generate(zi)(given=z)
                    _, pmfs = ref_pmfs(zendpoints[zi - 1] if zi > 0 else xendpoints, mu, scale)
                    state, sym = ANS(pmfs, bits=ansbits, quantbits=quantbits if zi > 0 else 8).decode(state)
                    input = zcentres[zi - 1, zrange, sym] if zi > 0 else xcentres[xrange, sym]
                    mu, scale = model.infer(zi)(given=input)
                    _, pmfs = ref_pmfs(zendpoints[zi], mu, scale)
                    state = ANS(pmfs, bits=ansbits, quantbits=quantbits).encode(state, zsymtop)
                    zsymtop = sym
                assert torch.all(x.long() == zsymtop)
            else:
                zs = [zsymtop]
                for zi in reversed(range(nz)):
                    z = zcentres[zi, zrange, zsymtop]
                    mu, scale = model.generate(zi)(given=z)
                    _, pmfs = ref_pmfs(zendpoints[zi - 1] if zi > 0 else xendpoints, mu, scale)
                    state, sym = ANS(pmfs, bits=ansbits, quantbits=quantbits if zi > 0 else 8).decode(state)
                    zs.append(sym)
                    zsymtop = sym
                zsymtop = zs.pop(0)
                for zi in reversed(range(nz)):
                    sym = zs.pop(0) if zi > 0 else zs[0]
                    input = zcentres[zi - 1, zrange, sym] if zi > 0 else xcentres[xrange, sym]
                    mu, scale = model.infer(zi)(given=input)
                    _, pmfs = ref_pmfs(zendpoints[zi], mu, scale)
                    state = ANS(pmfs, bits=ansbits, quantbits=quantbits).encode(state, zsymtop)
                    zsymtop = sym
                assert torch.all(x.long() == zs[0])
        assert initialstate == state
    return ops, sent, restbits, nets, cma


def make_model_and_chains():
    torch.manual_seed(50)
    rng = np.random.RandomState(11)
    # MNIST-shaped (zchannels=1, xs=(1,32,32)) with a narrow ResNet so the fixture stays small
    nz, quantbits = 2, 10
    xs, zch = (1, 32, 32), 1
    model = RefModel(xs=xs, nz=nz, zchannels=zch, nprocessing=1, kernel_size=3, resdepth=2, reswidth=12,
                     root_process=False)
    # perturb the data-independent init so layers are not degenerate (gain 0 / bias 0 everywhere)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".b") or n.endswith("gen_std"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    model.eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    xdim, zdim = int(np.prod(xs)), zch * 16 * 16
    images = synth_images(rng, 64, xs)
    zend, zcen, mins, maxs = synth_bins(model, nz, zdim, quantbits, images, rng)

    # model outputs (batched, non-compress mode is what the new framework's Model must reproduce)
    out = {"sd_" + k: v for k, v in sd.items()}
    out["cfg"] = np.array([xs[0], nz, zch, 1, 3, 2, 12], dtype=np.int64)
    with torch.no_grad():
        x = torch.from_numpy(images[:5].astype(np.float32))
        model.compress(False)
        mu, sc = model.infer(0)(given=x)
        out["infer0_in"], out["infer0_mu"], out["infer0_scale"] = images[:5], mu.numpy(), sc.numpy()
        z = torch.randn(5, *model.zdim)
        out["z_in"] = z.numpy()
        for i in range(nz):
            mu, sc = model.generate(i)(given=z)
            out[f"gen{i}_mu"], out[f"gen{i}_scale"] = mu.numpy(), np.broadcast_to(sc.numpy(), mu.shape).copy()
            if i > 0:
                mu, sc = model.infer(i)(given=z)
                out[f"infer{i}_mu"], out[f"infer{i}_scale"] = mu.numpy(), sc.numpy()
    np.savez_compressed(os.path.join(OUT, "model_mnist_small.npz"), **out)
    print("model_mnist_small.npz")

    nblocks = 3
    for bitswap in (1, 0):
        ops, sent, restbits, nets, cma = replay_chain(model, zend, zcen, images[:nblocks], nz, bitswap, quantbits,
                                                      xdim, zdim, cap=20000)
        o = {
            "cfg": np.array([xs[0], nz, zch, 1, 3, 2, 12, quantbits, bitswap, nblocks], dtype=np.int64),
            "images": images[:nblocks],
            "z_top_endpoints": zend[nz - 1][0], "z_top_centres": zcen[nz - 1][0],
            "z_mins": mins, "z_maxs": maxs,
            "sent_words": words(sent), "restbits_len": np.int64(len(restbits)),
            "nets": np.array(nets), "cma": np.array(cma),
            "op_kind": np.array([p["kind"] for p in ops], dtype=np.int8),
            "op_table": np.array([p["table"] for p in ops], dtype=np.int8),
            "op_q": np.array([p["q"] for p in ops], dtype=np.int8),
            "op_nwords": np.array([p["nwords"] for p in ops], dtype=np.int64),
            "op_head": np.array([p["head"] for p in ops], dtype=np.uint64),
        }
        for i, p in enumerate(ops):
            o[f"op{i}_mu"], o[f"op{i}_scale"], o[f"op{i}_sym"] = p["mu"], p["scale"], p["sym"].astype(np.int16)
        name = f"chain_mnist_small_{'bitswap' if bitswap else 'bbans'}.npz"
        np.savez_compressed(os.path.join(OUT, name), **o)
        print(name, "ops", len(ops), "words", len(sent), "cma", cma)

    # crop-model variant (gen_std is a conv, imagenetcrop_train.py:306-315,417): outputs only
    torch.manual_seed(51)
    cm = RefCropModel(xs=(3, 32, 32), nz=2, zchannels=2, nprocessing=1, kernel_size=3, resdepth=2, reswidth=10,
                      root_process=False)
    with torch.no_grad():
        for n, p in cm.named_parameters():
            if n.endswith(".b"):
                p.add_(torch.randn_like(p) * 0.3)
    cm.eval()
    out = {"sd_" + k: v.numpy() for k, v in cm.state_dict().items()}
    out["cfg"] = np.array([3, 2, 2, 1, 3, 2, 10], dtype=np.int64)
    with torch.no_grad():
        z = torch.randn(3, *cm.zdim)
        out["z_in"] = z.numpy()
        mu, sc = cm.generate(0)(given=z)
        out["gen0_mu"], out["gen0_scale"] = mu.numpy(), sc.numpy()
        x = torch.from_numpy(synth_images(rng, 3, (3, 32, 32)).astype(np.float32))
        out["infer0_in"] = x.numpy().astype(np.uint8)
        mu, sc = cm.infer(0)(given=x)
        out["infer0_mu"], out["infer0_scale"] = mu.numpy(), sc.numpy()
    np.savez_compressed(os.path.join(OUT, "model_crop_small.npz"), **out)
    print("model_crop_small.npz")


def make_bits_fixture():
    """The reference ANS class at precisions other than its hard-coded 31 (`bits` is a constructor argument,
    mnist_compress.py:14): tables and word streams at 16 / 24 / 28 bits, own file, own RNG."""
    rng = np.random.RandomState(4321)
    K, D, q = 256, 64, 8
    pm = torch.from_numpy(rng.dirichlet(np.full(K, 0.3), size=D))
    out = {"pmf_f64": pm.numpy(), "quantbits": np.int32(q)}
    np.random.seed(77)
    for bits in (16, 24, 28):
        a = ANS(pm, bits=bits, quantbits=q)
        out[f"b{bits}_f"] = a.pmfs.astype(np.uint32)
        out[f"b{bits}_cdf"] = a.cdfs.astype(np.uint32)
        st = init_state(300)
        out[f"b{bits}_state0"] = words(st)
        st, sym = a.decode(st)
        out[f"b{bits}_pop_sym"] = sym.numpy().astype(np.int32)
        out[f"b{bits}_state_after_pop"] = words(st)
        fresh = torch.from_numpy(rng.randint(0, K, size=D))
        st = a.encode(st, fresh)
        out[f"b{bits}_push_sym"] = fresh.numpy().astype(np.int32)
        out[f"b{bits}_state_after_push"] = words(st)
    np.savez_compressed(os.path.join(OUT, "rans_bits.npz"), **out)
    print("rans_bits.npz")


def make_rgb4_chain():
    """A deeper, colour chain: xs = (3,32,32), nz = 4, Bit-Swap, 2 blocks -- pins the layer ordering for
    zi > 1 (mnist_compress.py:176-205) and the 3072-dim pixel op; written to its own files so that the
    other fixtures never change when this one is regenerated."""
    torch.manual_seed(52)
    rng = np.random.RandomState(12)
    nz, quantbits = 4, 10
    xs, zch = (3, 32, 32), 2
    cfg = [xs[0], nz, zch, 1, 3, 4, 10]
    model = RefModel(xs=xs, nz=nz, zchannels=zch, nprocessing=1, kernel_size=3, resdepth=4, reswidth=10,
                     root_process=False)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".b") or n.endswith("gen_std"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    model.eval()
    xdim, zdim = int(np.prod(xs)), zch * 16 * 16
    images = synth_images(rng, 64, xs)
    zend, zcen, mins, maxs = synth_bins(model, nz, zdim, quantbits, images, rng)
    out = {"sd_" + k: v.numpy() for k, v in model.state_dict().items()}
    out["cfg"] = np.array(cfg, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "model_rgb4_small.npz"), **out)
    nblocks = 2
    for bitswap in (1, 0):
        _rgb4_chain(model, zend, zcen, mins, maxs, images, cfg, nz, quantbits, xdim, zdim, nblocks, bitswap)


def _rgb4_chain(model, zend, zcen, mins, maxs, images, cfg, nz, quantbits, xdim, zdim, nblocks, bitswap):
    ops, sent, restbits, nets, cma = replay_chain(model, zend, zcen, images[:nblocks], nz, bitswap, quantbits, xdim,
                                                  zdim, cap=30000)
    o = {
        "cfg": np.array(cfg + [quantbits, bitswap, nblocks], dtype=np.int64),
        "images": images[:nblocks],
        "z_top_endpoints": zend[nz - 1][0], "z_top_centres": zcen[nz - 1][0],
        "z_mins": mins, "z_maxs": maxs,
        "sent_words": words(sent), "restbits_len": np.int64(len(restbits)),
        "nets": np.array(nets), "cma": np.array(cma),
        "op_kind": np.array([p["kind"] for p in ops], dtype=np.int8),
        "op_table": np.array([p["table"] for p in ops], dtype=np.int8),
        "op_q": np.array([p["q"] for p in ops], dtype=np.int8),
        "op_nwords": np.array([p["nwords"] for p in ops], dtype=np.int64),
        "op_head": np.array([p["head"] for p in ops], dtype=np.uint64),
    }
    for i, p in enumerate(ops):
        o[f"op{i}_mu"], o[f"op{i}_scale"], o[f"op{i}_sym"] = p["mu"], p["scale"], p["sym"].astype(np.int16)
    name = f"chain_rgb4_small_{'bitswap' if bitswap else 'bbans'}.npz"
    np.savez_compressed(os.path.join(OUT, name), **o)
    print(name, "ops", len(ops), "words", len(sent), "cma", cma)


def make_surface_fixture():
    """The on-disk surface, produced by the reference's OWN functions (not a replay):
      * benchmark_compress.extract_blocks / unextract_blocks (:20-39) on a 70x100 image;
      * demo_compress.compress() (:72-162: sender loop, excess_state_len bookkeeping, `del state[0:excess - 1]`) and
        the container assembly of its main block (:272-283, restated below line by line);
      * demo_decompress main block (:214-227) + demo_decompress.decompress() (:69-148), asserted lossless here;
      * the per-experiment pickle of mnist_compress.py:265-267.
    The two functions hard-code the full-width crop model; `Model` in their module namespaces is replaced by a
    constructor that narrows it (reswidth 12, 1 processing layer, 4 ResNet layers) so that the fixture stays small.
    Checkpoint and bins reach them the way they do in the reference: as files `model/params/imagenetcrop/nz4` and
    `bins/imagenetcrop_nz4_z{endpoints,centres}10.pt` under the working directory."""
    import pickle
    import tempfile
    import demo_compress as ref_dc
    import demo_decompress as ref_dd
    import benchmark_compress as ref_bc

    torch.manual_seed(53)
    rng = np.random.RandomState(13)
    nz, quantbits, zch = 4, 10, 8
    cfg = [3, nz, zch, 1, 3, 4, 12]

    def small(**kw):
        kw.update(nprocessing=1, resdepth=4, reswidth=12, root_process=False)
        return RefCropModel(**kw)
    model = small(xs=(3, 32, 32), nz=nz, zchannels=zch, kernel_size=3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".b"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    model.eval()
    zdim = zch * 16 * 16
    images = synth_images(rng, 64, (3, 32, 32))
    zend, zcen, mins, maxs = synth_bins(model, nz, zdim, quantbits, images, rng)

    yy, xx = np.mgrid[0:70, 0:100]
    img = np.stack([127 + 80 * np.sin(yy / (7. + 3 * c) + c) * np.cos(xx / (9. + 2 * c)) for c in range(3)], -1)
    img = np.clip(np.rint(img + rng.randn(70, 100, 3) * 5), 0, 255).astype(np.uint8)
    blocks, h, w = ref_bc.extract_blocks(img)
    crop = ref_bc.unextract_blocks(blocks, h, w)
    assert np.array_equal(crop, img[:h, :w])

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "model/params/imagenetcrop"))
        os.makedirs(os.path.join(tmp, "bins"))
        torch.save(model.state_dict(), os.path.join(tmp, "model/params/imagenetcrop/nz4"))
        torch.save(torch.from_numpy(zend), os.path.join(tmp, f"bins/imagenetcrop_nz{nz}_zendpoints{quantbits}.pt"))
        torch.save(torch.from_numpy(zcen), os.path.join(tmp, f"bins/imagenetcrop_nz{nz}_zcentres{quantbits}.pt"))
        ref_dc.Model = ref_dd.Model = small
        os.chdir(tmp)
        try:
            np.random.seed(100)                                  # demo_compress.py:225
            state = ref_dc.compress(quantbits=quantbits, nz=nz, gpu=-1, blocks=blocks)
            trimmed = list(state)
            # demo_compress.py:272-283, verbatim
            state.append(state[-1] >> 32)
            state[-2] = state[-2] & ((1 << 32) - 1)
            state.append(blocks.shape[0])
            state.append(h)
            state.append(w)
            state_array = np.array(state, dtype=np.uint32)
            np.save(os.path.join(tmp, "img_bitswap"), state_array)
            npy_bytes = open(os.path.join(tmp, "img_bitswap.npy"), "rb").read()
            # demo_decompress.py:177-179 (reads the file), :214-227
            st = list(map(int, np.load(os.path.join(tmp, "img_bitswap.npy"))))
            w2 = st.pop()
            h2 = st.pop()
            nblocks = st.pop()
            st.append(st.pop() << 32 | st.pop())
            assert st == trimmed and (nblocks, h2, w2) == (len(blocks), h, w)
            out = ref_dd.decompress(quantbits=quantbits, nz=nz, gpu=-1, state=st, nblocks=nblocks)
            assert np.all(ref_bc.unextract_blocks(out, h2, w2) == crop)   # demo_decompress.py:235-236
        finally:
            os.chdir(cwd)
    # `st` is now what the receiver is left with: the untouched tail of the initial words (the reference never checks it)
    pk = pickle.dumps(trimmed)                                   # mnist_compress.py:265-267 writes pickle.dump(state, fp)
    # torch.sigmoid(float64) is third-party arithmetic whose last bit depends on the build / CPU dispatch:
    # a probe lets a test know whether the torch it runs on evaluates the CDF like the one that wrote this container
    probe_t = torch.from_numpy(np.random.RandomState(99).uniform(-40, 40, 1 << 16))
    probe = torch.sigmoid(probe_t).numpy()
    o = {"sd_" + k: v.numpy() for k, v in model.state_dict().items()}
    o.update(cfg=np.array(cfg + [quantbits], dtype=np.int64), image=img, blocks=blocks, hw=np.array([h, w]), crop=crop,
             container=state_array, container_npy=np.frombuffer(npy_bytes, dtype=np.uint8),
             state_pickle=np.frombuffer(pk, dtype=np.uint8), state_words=words(trimmed), rest_words=words(st),
             sigmoid_probe=probe,
             z_top_endpoints=zend[nz - 1][0], z_top_centres=zcen[nz - 1][0], z_mins=mins, z_maxs=maxs)
    np.savez_compressed(os.path.join(OUT, "demo_surface.npz"), **o)
    print("demo_surface.npz: blocks", blocks.shape, "container words", len(state_array), "of 10000 initial +",
          "bits/dim", 32 * (len(state_array) - 5 - 0) / (blocks.shape[0] * 3072))


def make_discretize_fixture():
    """discretize()'s sampling procedure (discretization.py:55-83) replayed around the imported reference pieces
    (Model, logistic_eps, transform, discretize_kbins) with EXPLICIT seeds, so that bitswap_amd.bins.discretize can
    be driven with the very same noise and data order on the GPU.  Noise: torch.manual_seed(777) right before the
    first draw, then the reference's call order -- gen_samples[-1], then for zi = nz-1..1: all generative batches,
    all inference batches.  Data: batch bi = images[order[bi*128 : (bi+1)*128]]."""
    torch.manual_seed(54)
    rng = np.random.RandomState(14)
    nz, quantbits, zch = 3, 8, 2
    cfg = [3, nz, zch, 1, 3, 3, 10]
    model = RefModel(xs=(3, 32, 32), nz=nz, zchannels=zch, nprocessing=1, kernel_size=3, resdepth=3, reswidth=10,
                     root_process=False)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".b") or n.endswith("gen_std"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    model.eval()
    zdim = zch * 16 * 16
    K = 1 << quantbits
    images = synth_images(rng, 96, (3, 32, 32))
    nsamples, bs = 2 * K, 128
    batches = nsamples // bs
    order = rng.randint(0, len(images), size=nsamples)
    data = torch.from_numpy(images[order]).float()
    torch.manual_seed(777)
    gen = np.zeros((nz, nsamples) + model.zdim, dtype=np.float16)
    gen[-1] = logistic_eps((nsamples,) + model.zdim, device="cpu", bound=1e-30).numpy()
    inf = np.zeros((nz, nsamples) + model.zdim, dtype=np.float16)
    with torch.no_grad():
        for zi in reversed(range(1, nz)):
            for bi in range(batches):
                mu, scale = model.generate(zi)(given=torch.from_numpy(gen[zi][bi * bs: bi * bs + bs]).float())
                gen[zi - 1][bi * bs: bi * bs + bs] = transform(logistic_eps(mu.shape, device="cpu", bound=1e-30), mu, scale)
            for bi in range(batches):
                given = (data[bi * bs: bi * bs + bs] if nz - zi - 1 == 0
                         else torch.from_numpy(inf[nz - zi - 2][bi * bs: bi * bs + bs]).float())
                mu, scale = model.infer(nz - zi - 1)(given=given)
                inf[nz - zi - 1][bi * bs: bi * bs + bs] = transform(logistic_eps(mu.shape, device="cpu", bound=1e-30), mu, scale).numpy()
    zend = np.zeros((nz, zdim, K - 1))
    zcen = np.zeros((nz, zdim, K))
    zb = Bins(torch.zeros((1, 1, zdim)), torch.ones((1, 1, zdim)), quantbits)
    zend[nz - 1], zcen[nz - 1] = zb.endpoints().numpy(), zb.centres().numpy()
    mins, maxs = [], []
    for zi in range(nz - 1):
        samples = np.concatenate([gen[zi], inf[zi]], axis=0).astype(np.float64)
        zend[zi], zcen[zi] = discretize_kbins(model, samples, quantbits, strategy="uniform")
        flat = samples.reshape(-1, zdim)
        mins.append(flat.min(0))
        maxs.append(flat.max(0))
    o = {"sd_" + k: v.numpy() for k, v in model.state_dict().items()}
    o.update(cfg=np.array(cfg + [quantbits], dtype=np.int64), images=images, order=order.astype(np.int64),
             noise_seed=np.int64(777), z_mins=np.array(mins), z_maxs=np.array(maxs),
             z_top_endpoints=zend[nz - 1][0], z_top_centres=zcen[nz - 1][0],
             zend_layer0_dim5=zend[0][5], zcen_layer0_dim5=zcen[0][5])
    np.savez_compressed(os.path.join(OUT, "discretize_small.npz"), **o)
    print("discretize_small.npz: mins", np.array(mins).min(), "maxs", np.array(maxs).max())


def make_mnist_full_chain(nblocks=100):
    """BASELINE configs[0] at its REAL width (mnist_compress.py:85-86,107: nz 2, reswidth 63, resdepth 8, nprocessing 4), ONE
    chain of `nblocks` = 100 blocks (one "experiment", :102-103), Bit-Swap and BB-ANS, the reference's sender and receiver
    on CPU.  What it is for: the HIP path evaluates the CDF with its own deterministic routine, torch.sigmoid differs from it
    in ~0.1 ppm of table entries -- this chain measures how far a teacher-forced HIP stream follows the reference's own words
    before such an entry forks it.  Stored compactly: per op (mu, scale) float32 as the reference's net emitted
    them, the symbols, the word count and the head after the op; the pixel scale is a parameter (mnist_train.py:411), stored
    once; the prior ops carry no arrays (mu 0, scale 1)."""
    _full_chain("mnist", (1, 32, 32), 2, 1, 63, nblocks, 55, 15)


def make_cifar_full_chain(nblocks=6):
    """BASELINE configs[1] -- the bench headline's model -- at its REAL width (cifar_compress.py:80-81,106: nz 8, zchannels 8,
    reswidth 252, resdepth 8, nprocessing 4; model/cifar_train.py's Model is mnist_train.py's class with another log directory),
    ONE Bit-Swap / BB-ANS chain of `nblocks` blocks through the reference's sender on CPU: 2048-dim latent rows with 1024 bins
    and 3072-dim pixel rows, the launch shapes of the headline.  Same compact layout as the MNIST chains."""
    _full_chain("cifar", (3, 32, 32), 8, 8, 252, nblocks, 56, 16)


def make_imagenet_full_chain(nblocks=4):
    """BASELINE configs[2] / [4] -- north_star's target shape -- at its REAL width (imagenet_compress.py:83-84: nz 4, zchannels 8,
    reswidth 254; model/imagenet_train.py's Model is the same class again): one Bit-Swap and one BB-ANS chain of `nblocks` blocks."""
    _full_chain("imagenet", (3, 32, 32), 4, 8, 254, nblocks, 57, 17)


def _full_chain(tag, xs, nz, zch, reswidth, nblocks, tseed, nseed):
    torch.manual_seed(tseed)
    rng = np.random.RandomState(nseed)
    quantbits = 10
    cfg = [xs[0], nz, zch, 4, 3, 8, reswidth]
    model = RefModel(xs=xs, nz=nz, zchannels=zch, nprocessing=4, kernel_size=3, resdepth=8, reswidth=reswidth, root_process=False)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".b") or n.endswith("gen_std"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    model.eval()
    xdim, zdim = int(np.prod(xs)), zch * 16 * 16
    images = synth_images(rng, 128 if tag == "mnist" else 32, xs)
    zend, zcen, mins, maxs = synth_bins(model, nz, zdim, quantbits, images, rng)
    for bitswap in (1, 0):
        ops, sent, restbits, nets, cma = replay_chain(model, zend, zcen, images[:nblocks], nz, bitswap, quantbits, xdim, zdim,
                                                      cap=0)
        prior = np.array([p["kind"] == 1 and p["table"] == nz - 1 for p in ops])
        zops = [p for p, pr in zip(ops, prior) if p["table"] >= 0 and not pr]
        xops = [p for p in ops if p["table"] < 0]
        assert all(np.array_equal(p["scale"], xops[0]["scale"]) for p in xops)
        o = {
            "cfg": np.array(cfg + [quantbits, bitswap, nblocks], dtype=np.int64),
            "images": images[:nblocks],
            "z_top_endpoints": zend[nz - 1][0], "z_top_centres": zcen[nz - 1][0], "z_mins": mins, "z_maxs": maxs,
            "sent_words": words(sent), "restbits_len": np.int64(len(restbits)), "nets": np.array(nets), "cma": np.array(cma),
            "op_kind": np.array([p["kind"] for p in ops], dtype=np.int8),
            "op_table": np.array([p["table"] for p in ops], dtype=np.int8),
            "op_prior": prior,
            "op_q": np.array([p["q"] for p in ops], dtype=np.int8),
            "op_nwords": np.array([p["nwords"] for p in ops], dtype=np.int64),
            "op_head": np.array([p["head"] for p in ops], dtype=np.uint64),
            "z_mu": np.stack([p["mu"] for p in zops]), "z_scale": np.stack([p["scale"] for p in zops]),
            "z_sym": np.stack([p["sym"] for p in zops]).astype(np.int16),
            "prior_sym": np.stack([p["sym"] for p, pr in zip(ops, prior) if pr]).astype(np.int16),
            "x_mu": np.stack([p["mu"] for p in xops]), "x_scale": xops[0]["scale"],
            "x_sym": np.stack([p["sym"] for p in xops]).astype(np.uint8),
        }
        name = f"chain_{tag}_full_{'bitswap' if bitswap else 'bbans'}.npz"
        np.savez_compressed(os.path.join(OUT, name), **o)
        print(name, "ops", len(ops), "words", len(sent), "cma", cma[-1])


def make_draws_fixture():
    """The numpy draws of the dataset scripts in the reference's order (mnist_compress.py:94,133-137,158; imagenet_compress.py:
    94,134,158): seed(100), choice(len(test_set), (100, 100), replace=False) -- the exists() guard at :133 never hits, np.save
    appends .npy -- then per experiment the 10000 initial words.  len(test_set) = 10000 (MNIST / CIFAR-10 test sets) and 50000
    (ImageNet 32x32 validation set).  Stored: the first indices and the first / last words of experiments 0-2."""
    import random
    out = {}
    for ntest in (10000, 50000):
        np.random.seed(100)
        random.seed(50)
        torch.manual_seed(50)
        experiments, ndatapoints = 100, 100
        randindices = np.random.choice(ntest, size=(experiments, ndatapoints), replace=False)
        out[f"n{ntest}_indices"] = randindices[:3, :8].astype(np.int64)
        for ei in range(3):
            state = list(map(int, np.random.randint(low=1 << 16, high=(1 << 32) - 1, size=10000, dtype=np.uint32)))
            state[-1] = state[-1] << 32
            initialstate = state.copy()
            out[f"n{ntest}_e{ei}_first4"] = np.array(initialstate[:4], dtype=np.uint64)
            out[f"n{ntest}_e{ei}_last2"] = np.array(initialstate[-2:], dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "draws.npz"), **out)
    print("draws.npz", out["n10000_e0_first4"], out["n50000_e0_first4"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mnist_full":
        make_mnist_full_chain()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cifar_full":
        make_cifar_full_chain()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "imagenet_full":
        make_imagenet_full_chain()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "draws":
        make_draws_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "surface":
        make_surface_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "discretize":
        make_discretize_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rgb4":
        make_rgb4_chain()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bits":
        make_bits_fixture()
        sys.exit(0)
    make_tables_and_rans()
    make_bins()
    make_model_and_chains()
    make_rgb4_chain()
    make_bits_fixture()
    make_surface_fixture()
    make_discretize_fixture()
    make_draws_fixture()
    make_mnist_full_chain()
    make_cifar_full_chain()
    make_imagenet_full_chain()
