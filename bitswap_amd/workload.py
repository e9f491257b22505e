"""Synthetic workloads shaped like BASELINE.json's configs.

No datasets or checkpoints ship with the reference and there is no network, so
benchmarks, smoke tests and the CLIs' `--synthetic` mode use: seeded random-init weights of the
dataset's architecture (the reference's own initialisation, modules.py:68-73, plus a small seeded
perturbation of gains/biases so no layer is degenerate), smooth synthetic 32x32 blocks, and bins
drawn by the reference's sampling procedure with a reduced samples-per-bin count.
"""
import numpy as np
import torch

from .bins import discretize
from .model import preset

# name -> (dataset preset, nz): BASELINE.json configs[0..3]
WORKLOADS = {
    "mnist2": ("mnist", 2),
    "cifar8": ("cifar", 8),
    "imagenet4": ("imagenet", 4),
    "imagenetcrop4": ("imagenetcrop", 4),
}


def synthetic_blocks(n, xs, seed=0):
    """uint8 [n, C*H*W] blocks: an 8x8 Gaussian field bilinearly upsampled + N(0, 4^2) noise
    (SURVEY.md 8d (ii)), flattened in CHW order like the reference's x.view(xdim)."""
    g = torch.Generator().manual_seed(seed)
    c, h, w = xs
    base = torch.randn((n, c, 8, 8), generator=g)
    up = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)
    img = 127.5 + 60.0 * up + 4.0 * torch.randn((n, c, h, w), generator=g)
    return img.round().clamp(0, 255).to(torch.uint8).view(n, -1)


def synthetic_model(dataset, nz, device, seed=50, nn_batch=None, small=None):
    """Random-init Model of the dataset's architecture (seed 50 = the reference CLIs' torch seed,
    mnist_compress.py:96).  `small` = reswidth override for fast tests."""
    torch.manual_seed(seed)
    kw = {}
    m = preset(dataset, nz, nn_batch=nn_batch, **kw)
    if small:
        from .model import Model
        m = Model(xs=m.xs, nz=nz, zchannels=m.zchannels, nprocessing=1, kernel_size=3, resdepth=min(nz, 2),
                  reswidth=small, conditional_gen_std=m.conditional_gen_std, nn_batch=nn_batch)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith(".b"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith(".gain"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif n == "gen_std":
                p.add_(-3.0 + 0.1 * torch.randn(p.shape, generator=g))   # pixel scale ~ 0.05, as trained models have
    m = m.to(device).eval().fold()
    # HIP device: one fused epilogue launch per convolution (csrc/net_epilogue.hip)
    return m.fuse() if torch.device(device).type == "cuda" else m


def synthetic_bins(model, dataset, nz, quantbits, device, ppb=2, seed=7):
    torch.manual_seed(seed)
    data = synthetic_blocks(512, model.xs, seed=seed).view((-1,) + tuple(model.xs))
    return discretize(nz, quantbits, torch.float64, device, model, dataset, data=data, ppb=ppb, save=False,
                      cache_dir="/nonexistent")


def build(name, device, quantbits=10, nn_batch=None, small=None, ppb=2):
    dataset, nz = WORKLOADS[name]
    model = synthetic_model(dataset, nz, device, nn_batch=nn_batch, small=small)
    zend, zcen = synthetic_bins(model, dataset, nz, quantbits, device, ppb=ppb)
    return model, zend, zcen
