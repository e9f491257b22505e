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


def synthetic_model(dataset, nz, device, seed=50, nn_batch=None, small=None, regime=None):
    """Random-init Model of the dataset's architecture (seed 50 = the reference CLIs' torch seed,
    mnist_compress.py:96).  `small` = reswidth override for fast tests.  regime="lowrate": calibrate_lowrate()."""
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
    if regime == "lowrate":
        calibrate_lowrate(m)
    elif regime is not None:
        raise ValueError(f"unknown regime {regime!r}")
    m = m.to(device).eval().fold()
    # HIP device: one fused epilogue launch per convolution (csrc/net_epilogue.hip)
    return m.fuse() if torch.device(device).type == "cuda" else m


def synthetic_bins(model, dataset, nz, quantbits, device, ppb=2, seed=7):
    torch.manual_seed(seed)
    data = synthetic_blocks(512, model.xs, seed=seed).view((-1,) + tuple(model.xs))
    return discretize(nz, quantbits, torch.float64, device, model, dataset, data=data, ppb=ppb, save=False,
                      cache_dir="/nonexistent")


def calibrate_lowrate(m):
    """Turn the seeded random-init model into one that codes its OWN samples at a trained model's rate (3-5 bits/dim,
    BASELINE.md: 3.5-4.5) instead of the 26 bits/dim of random weights on unrelated images -- the regime of peaked tables,
    saturated tails (f = 1 over most of a row) and few renormalisations that no checkpoint is available to provide
    offline.  Done in the heads only, so the conv stacks run exactly as before:
      * every latent scale head is pushed to the clamp of the reference's parametrisation -- q: 0.1 + 0.9 sigmoid(. + 2)
        (mnist_train.py:349,368), p: 0.1 + 0.9 softplus(. + log(e - 1)) (:426) -- by a bias of -14 and a vanishing gain:
        scale = 0.1, where trained Bit-Swap models sit for most dimensions;
      * every latent mean head gets a vanishing gain (mu = its bias, |mu| ~ 0.1 = one scale): posterior and conditional
        prior of the lower layers differ by a fraction of a scale, the top layer pays KL(Logistic(mu, 0.1) || Logistic(0, 1));
      * pixels: mean head with a vanishing gain (a fixed mean image), scale 2/255/8 + softplus(-6) = 0.0035 (0.44 pixel
        levels; the reference's floor is 2/255/8 = 0.00098, mnist_train.py:411): 4-5 bins carry the mass of a 256-bin row.
    Blocks for it come from the model's own generative path: lowrate_blocks()."""
    from .model import WnConv2d
    tiny = -4.3          # softplus(-4.3) = 0.0135: gain 1/50 of the initial softplus(0)

    def head(seq_or_conv):
        return seq_or_conv[0] if isinstance(seq_or_conv, torch.nn.Sequential) else seq_or_conv
    with torch.no_grad():
        stds = [m.infer_std] + [head(q) for q in m.deepinfer_std] + [head(q) for q in m.deepgen_std]
        mus = [m.infer_mu] + [head(q) for q in m.deepinfer_mu] + [head(q) for q in m.deepgen_mu] + [head(m.gen_mu)]
        for c in stds:
            assert isinstance(c, WnConv2d)
            c.gain.fill_(tiny)
            c.b.fill_(-14.0)
        for c in mus:
            c.gain.fill_(tiny)
        if m.conditional_gen_std:
            c = head(m.gen_std)
            c.gain.fill_(tiny)
            c.b.fill_(-6.0)
        else:
            m.gen_std.fill_(-6.0)
    m.unfold()
    return m


def lowrate_blocks(model, n, seed=0, batch=64):
    """uint8 [n, X] blocks drawn from the model's own generative path (ancestral sampling, continuous latents):
    z_L ~ Logistic(0, 1), z_{i-1} ~ p(z_{i-1} | z_i), x ~ the discretized logistic p(x | z_1) -- the data a
    calibrate_lowrate() model codes at its ELBO."""
    dev = next(model.parameters()).device
    g = torch.Generator().manual_seed(seed)
    was = model.compressing
    model.compress(True)
    out = []

    def noise(shape):
        u = torch.rand(shape, generator=g, dtype=torch.float64).clamp_(1e-12, 1 - 1e-12)
        return (torch.log(u) - torch.log1p(-u)).float().to(dev)
    try:
        with torch.no_grad():
            for s in range(0, n, batch):
                k = min(batch, n - s)
                z = noise((k, model.zdim_flat))
                for i in reversed(range(1, model.nz)):
                    mu, sc = model.generate(i)(given=z)
                    z = mu + sc * noise(mu.shape)
                mu, sc = model.generate(0)(given=z)
                x = mu + sc * noise(mu.shape)
                out.append(((x * 127.5 + 127.5).round().clamp(0, 255)).to(torch.uint8).cpu())
    finally:
        model.compress(was)
    return torch.cat(out, 0)


def build(name, device, quantbits=10, nn_batch=None, small=None, ppb=2, regime=None):
    dataset, nz = WORKLOADS[name]
    model = synthetic_model(dataset, nz, device, nn_batch=nn_batch, small=small, regime=regime)
    zend, zcen = synthetic_bins(model, dataset, nz, quantbits, device, ppb=ppb)
    return model, zend, zcen
