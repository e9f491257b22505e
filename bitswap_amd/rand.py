"""Logistic-distribution helpers and bin tables: the host-side mirror of the reference's
utils/torch/rand.py (function and class names kept so reference call sites read the same).

Only `logistic_cdf` is on the coding hot path, and there the HIP kernels evaluate it fused with
the table build (bitswap_amd.hip.logistic_tables / logistic_fc); the torch version below is the
reference formula, used by the ELBO metric and by tests.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _softplus(x):
    return -F.logsigmoid(-x)


def transform(eps, mu, scale):
    """rand.py:6-8"""
    return mu + scale * eps


def logistic_eps(shape, device, bound=1e-5):
    """Logistic(0,1) noise through the inverse CDF of clamped uniforms (rand.py:11-21)."""
    u = torch.rand(shape, device=device).clamp_(min=bound, max=1 - bound)
    return torch.log(u) - torch.log1p(-u)


def logistic_logp(mu, scale, x):
    """rand.py:24-28"""
    y = -(x - mu) / scale
    return (-y - torch.log(scale) - 2 * _softplus(-y)).flatten(2)


def discretized_logistic_logp(mu, scale, x):
    """Discretized logistic log-pmf of x in {0..255} (rand.py:32-64, after PixelCNN++)."""
    xr = (x - 127.5) / 127.5
    inv = 1. / scale
    xc = xr - mu
    plus_in = inv * (xc + 1. / 255.)
    min_in = inv * (xc - 1. / 255.)
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)
    mid_in = inv * xc
    log_pdf_mid = mid_in - torch.log(scale) - 2. * _softplus(mid_in)
    inner = torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-12)),
                        log_pdf_mid - np.log(127.5))
    upper = torch.where(xr > .999, -_softplus(min_in), inner)
    return torch.where(xr < -.999, plus_in - _softplus(plus_in), upper).flatten(1)


def logistic_cdf(x, mu, scale):
    """rand.py:67-68"""
    return torch.sigmoid((x - mu) / scale)


def logistic_icdf(p, mu, scale):
    """rand.py:71-72"""
    return mu + scale * torch.log(p / (1. - p))


class Bins:
    """Equal-mass bins under Logistic(mu, scale) (rand.py:78-128).  endpoints(): shape(mu)+[2^p - 1]
    interior edges; centres(): shape(mu)+[2^p] bin medians.  Computed in mu.dtype -- the reference
    builds the top-layer bins from float32 zeros/ones (discretization.py:25), and so do we."""

    def __init__(self, mu, scale, precision):
        self.precision, self.nbins = precision, 1 << precision
        self.mu, self.scale = mu, scale
        self.type, self.device, self.shape = mu.dtype, mu.device, list(mu.shape)

    def _through_icdf(self, probs):
        nd = len(self.shape)
        probs = probs.view([-1] + [1] * nd).expand([-1] + self.shape)
        vals = logistic_icdf(probs, self.mu, self.scale)
        return vals.permute(list(range(1, nd + 1)) + [0])

    def endpoints(self):
        return self._through_icdf(torch.arange(1., self.nbins, dtype=self.type, device=self.device) / self.nbins)

    def centres(self):
        return self._through_icdf(
            (torch.arange(end=self.nbins, dtype=self.type, device=self.device) + .5) / self.nbins)


class ImageBins:
    """Pixel bins of the discretized logistic on [-1,1] (rand.py:134-153): 255 interior endpoints
    and 256 centres per dimension, returned as expanded (row-stride-0) views."""

    def __init__(self, type, device, shape):
        self.type, self.device, self.shape = type, device, [shape]

    # The arithmetic runs on the HOST and the result is copied to the device: torch divides by a scalar with IEEE
    # division on the CPU but with a multiplication by the rounded reciprocal on the GPU, which moves a third of these
    # endpoints by an ulp or two -- enough to flip a table entry once in ~1e8 bins, i.e. to make a GPU sender and a CPU
    # receiver (or the oracle) disagree on a stream.  Bins are data both ends must share bit for bit.
    def endpoints(self):
        e = torch.arange(1, 256, dtype=self.type, device="cpu")
        e = (((e - 127.5) / 127.5) - 1. / 255.).to(self.device)
        return e[None,].expand(self.shape + [-1])

    def centres(self):
        c = torch.arange(0, 256, dtype=self.type, device="cpu")
        c = ((c - 127.5) / 127.5).to(self.device)
        return c[None,].expand(self.shape + [-1])
