#!/usr/bin/env python3
"""How wide is the window of bins a windowed table kernel would have to evaluate?  A bin with |t| >= 22 has f = 1
exactly, so a row only needs the bins within +-22 scale of mu: width = 44 scale / h bins.  Full-width cifar8 model on the CPU, the
calibrated low-rate regime and the random-weight regime, every table of a block step.  Result (profiles/r04_window_stats.txt): the
latent windows are 1600 - 2500 bins wide in BOTH regimes -- wider than the K = 1024 row, because the bins are fitted to samples of
the very distributions being coded -- so there is nothing to skip in a latent row; only the pixel rows of the low-rate regime are
peaked (21 of 256 bins), and the pixel tables are 1 % of a step.
"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from bitswap_amd import workload
from bitswap_amd.bins import uniform_step
torch.set_num_threads(8)
t0 = time.time()
for regime in ("lowrate", None):
    model, zend, zcen = workload.build("cifar8", "cpu", quantbits=10, regime=regime)
    print(regime, "built", round(time.time() - t0, 1), "s")
    model.compress(True)
    B = 8
    if regime == "lowrate":
        x = workload.lowrate_blocks(model, B, seed=3)
    else:
        x = workload.synthetic_blocks(B, model.xs, seed=3)
    xc = (x.float() - 127.5) / 127.5
    given = xc
    with torch.no_grad():
        for zi in range(model.nz):
            mu, sc = model.infer(zi)(given)
            e = zend[zi].numpy()
            h = uniform_step(e)
            if h is not None:
                w = 44.0 * sc.numpy() / h[None, :] + 2
                jl = (mu.numpy() - 22 * sc.numpy() - e[None, :, 0]) / h[None, :]
                print(f" infer{zi}: scale med {np.median(sc.numpy()):.3f} h med {np.median(h):.5f} window bins med {np.median(w):.0f}  frac<=496: {(w <= 496).mean():.3f} frac<=240: {(w<=240).mean():.3f}")
            else:
                print(f" infer{zi}: top layer (spec 1)")
            # sample z at the mean
            z = mu
            mu_g, sc_g = model.generate(zi)(z)
            if zi > 0:
                e = zend[zi - 1].numpy(); h = uniform_step(e)
                w = 44.0 * sc_g.numpy() / h[None, :] + 2
                print(f" gen{zi}:   scale med {np.median(sc_g.numpy()):.3f} window bins med {np.median(w):.0f}  frac<=496: {(w <= 496).mean():.3f} frac<=240: {(w<=240).mean():.3f}")
            else:
                w = 44.0 * sc_g.numpy() / (2 / 255.0) + 2
                print(f" gen0 (pixels): scale med {np.median(sc_g.numpy()):.4f} window bins med {np.median(w):.0f} frac<=64: {(w<=64).mean():.3f}")
            given = z
