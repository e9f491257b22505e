"""Latent discretization bins: mirror of the reference's discretization.py.

`discretize(nz, quantbits, type, device, model, dataset)` returns (zendpoints [nz, Z, 2^q - 1],
zcentres [nz, Z, 2^q]) exactly as discretization.py:9-99 does: the top layer has equal-mass bins
under Logistic(0,1) (computed in float32 like the reference, :25-27), every lower layer has
uniform-width bins per dimension between the minimum and maximum of samples drawn from the
generative model (ancestral) and the inference model (posterior on training images), stored as
float16 (:59-61).  File names of the cache (`bins/<ds>_nz<k>_z{endpoints,centres}<q>.pt`) are the
reference's, so its published bins drop in.

Differences: sampling runs on the device in one pass per layer; there are no datasets offline, so
the caller supplies training images (`data`); scikit-learn's KBinsDiscretizer(strategy='uniform')
is replaced by its closed form, numpy.linspace(min, max, K+1) on float64-cast samples (NumPy 2
would otherwise return float16 edges for float16 samples -- SURVEY 7f).
"""
import os

import numpy as np
import torch

from .rand import Bins, logistic_eps, transform


def uniform_bins(mins, maxs, quantbits):
    """Closed form of discretize_kbins(..., strategy='uniform') (discretization.py:105-118):
    per dimension edges = numpy.linspace(min, max, K+1); endpoints = edges[1:-1],
    centres = midpoints.  mins/maxs float64 [Z] -> ([Z, K-1], [Z, K]) float64."""
    K = 1 << quantbits
    mins, maxs = np.asarray(mins, dtype=np.float64), np.asarray(maxs, dtype=np.float64)
    edges = np.stack([np.linspace(a, b, K + 1) for a, b in zip(mins, maxs)])
    return edges[:, 1:-1], (edges[:, :-1] + edges[:, 1:]) / 2


def uniform_step(endpoints, tol_ulps=8.0):
    """Bin width per row if every row of `endpoints` [D, K-1] (float64 tensor or array, rows may be expanded views)
    is an arithmetic progression up to rounding -- what discretize_kbins(strategy='uniform') produces
    (discretization.py:105-118) -- else None.  h = (e[K-2] - e[0]) / (K - 2), two IEEE float64 operations on the host,
    so sender and receiver derive the same numbers from the same bins.  The rows qualify when no endpoint is further
    than tol_ulps units in the last place (of the row's largest magnitude) from e[0] + j*h: the deterministic CDF
    spec 2 (include/bitswap_hip.h) treats that distance as a first-order correction.
    -> float64 numpy array [D] or None."""
    e = endpoints.detach().cpu().numpy() if torch.is_tensor(endpoints) else np.asarray(endpoints)
    e = e.astype(np.float64, copy=False)
    if e.ndim != 2 or e.shape[1] < 3:
        return None
    n = e.shape[1] - 1
    with np.errstate(all="ignore"):
        h = (e[:, -1] - e[:, 0]) / np.float64(n)
        if not (np.all(np.isfinite(h)) and np.all(h > 0)):
            return None
        ideal = e[:, :1] + np.arange(n + 1, dtype=np.float64)[None, :] * h[:, None]
        ulp = np.spacing(np.maximum(np.abs(e[:, 0]), np.abs(e[:, -1])))
        if not np.all(np.abs(e - ideal) <= tol_ulps * ulp[:, None]):
            return None
    return np.ascontiguousarray(h)


def top_bins(zdim_flat, quantbits):
    """discretization.py:25-27 -- float32 zeros/ones on the CPU, like the reference."""
    zb = Bins(torch.zeros((1, 1, zdim_flat)), torch.ones((1, 1, zdim_flat)), quantbits)
    return zb.endpoints().numpy()[0, 0].astype(np.float64), zb.centres().numpy()[0, 0].astype(np.float64)


def _cache_names(cache_dir, dataset, nz, quantbits):
    return (os.path.join(cache_dir, f"{dataset}_nz{nz}_zendpoints{quantbits}.pt"),
            os.path.join(cache_dir, f"{dataset}_nz{nz}_zcentres{quantbits}.pt"))


def discretize(nz, quantbits, type, device, model, dataset, data=None, ppb=30, cache_dir="bins", batch=128,
               save=True, eps_fn=None, order=None):
    """Same positional signature as the reference.  `data`: uint8/float images [N, C, 32, 32] in
    [0, 255] used for the posterior samples when no cached bins exist.
    eps_fn(shape) -> Logistic(0,1) noise on `device` (default: rand.logistic_eps on the device, bound 1e-30 like
    discretization.py:62,70,78); order: indices into `data`, one per sample (default: random with replacement -- the
    reference walks a shuffled DataLoader, :45-53).  Both exist so that the sampling procedure can be replayed
    against the reference's with the same noise (tests/golden/make_golden.py::make_discretize_fixture)."""
    fe, fc = _cache_names(cache_dir, dataset, nz, quantbits)
    if os.path.exists(fe) and os.path.exists(fc):
        zendpoints = torch.load(fe, map_location="cpu")
        zcentres = torch.load(fc, map_location="cpu")
        return zendpoints.type(type).to(device), zcentres.type(type).to(device)

    K = 1 << quantbits
    Z = int(np.prod(model.zdim))
    nsamples = ppb * K
    zendpoints = np.zeros((nz, Z, K - 1))
    zcentres = np.zeros((nz, Z, K))
    zendpoints[nz - 1], zcentres[nz - 1] = top_bins(Z, quantbits)
    if nz > 1:
        if data is None:
            raise FileNotFoundError(f"{fe} not found and no training images given to sample the bins from")
        data = torch.as_tensor(data)
        was_compressing = model.compressing
        model.compress(False)
        dev = torch.device(device)
        if eps_fn is None:
            eps_fn = lambda shape: logistic_eps(shape, device=dev, bound=1e-30)
        nb = nsamples // batch                  # whole batches only, like the reference (:55-56)
        assert nb >= 1, "ppb * 2^quantbits must be at least one batch of 128 samples"
        with torch.no_grad():
            # float16 sample stores, like the reference (:59-61); the tail beyond nb*batch stays zero there too
            gen = torch.zeros((nz, nsamples) + tuple(model.zdim), dtype=torch.float16, device=dev)
            inf = torch.zeros((nz, nsamples) + tuple(model.zdim), dtype=torch.float16, device=dev)
            gen[-1] = eps_fn((nsamples,) + tuple(model.zdim)).half()
            idx = torch.as_tensor(order) if order is not None else torch.randint(0, data.shape[0], (nsamples,))
            for zi in reversed(range(1, nz)):
                for bi in range(nb):            # ancestral samples of z_{zi-1} (:66-70)
                    sl = slice(bi * batch, (bi + 1) * batch)
                    mu, scale = model.generate(zi)(given=gen[zi][sl].float())
                    gen[zi - 1][sl] = transform(eps_fn(tuple(mu.shape)), mu, scale).half()
                li = nz - zi - 1                # posterior samples of z_{li+1} on the data (:73-78)
                for bi in range(nb):
                    sl = slice(bi * batch, (bi + 1) * batch)
                    given = data[idx[sl]].to(dev).float() if li == 0 else inf[li - 1][sl].float()
                    mu, scale = model.infer(li)(given=given)
                    inf[li][sl] = transform(eps_fn(tuple(mu.shape)), mu, scale).half()
            for zi in range(nz - 1):
                s = torch.cat([gen[zi], inf[zi]], dim=0).reshape(-1, Z).double()
                zendpoints[zi], zcentres[zi] = uniform_bins(s.min(0).values.cpu().numpy(),
                                                            s.max(0).values.cpu().numpy(), quantbits)
        model.compress(was_compressing)
    zendpoints, zcentres = torch.from_numpy(zendpoints), torch.from_numpy(zcentres)
    if save:
        os.makedirs(cache_dir, exist_ok=True)
        torch.save(zendpoints, fe)
        torch.save(zcentres, fc)
    return zendpoints.type(type).to(device), zcentres.type(type).to(device)
