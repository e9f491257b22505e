#!/usr/bin/env python3
"""CPU baseline of the REFERENCE's own Python path (cpu_baseline.kind "reference").

Replays the sender (mnist_compress.py:164-263) and the receiver (:277-358) loops verbatim -- the same replay
tests/golden/make_golden.py uses for the chain fixtures -- around the imported, unmodified reference classes
(ANS, Model, logistic_cdf, Bins, ImageBins) on device "cpu", at the reference's FULL model widths, and times them.
Needs the reference (/root/reference or $BITSWAP_REFERENCE).  bench.py runs it on the host of the GPU run when the
reference is there (`--config <workload> --where "same host as the GPU run"`); the GPU boxes of this project have no copy, so
there bench.py quotes the committed measurement from the build container, labelled as such.

    python tools/ref_cpu_baseline.py > profiles/r02_ref_cpu_baseline.json

No checkpoints or datasets exist offline: seeded random-init weights of the reference architecture, synthetic blocks,
top layer = the reference's analytic bins, lower layers = uniform bins over [-8, 8] (the timing does not depend on where
the bins sit; every run is asserted lossless and fully unwound by the replay itself).
"""
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402  (stubs torchvision/tensorboardX, imports the reference)


def bins(nz, zdim, q):
    K = 1 << q
    zend, zcen = np.zeros((nz, zdim, K - 1)), np.zeros((nz, zdim, K))
    zb = mg.Bins(torch.zeros((1, 1, zdim)), torch.ones((1, 1, zdim)), q)
    zend[nz - 1], zcen[nz - 1] = zb.endpoints().numpy(), zb.centres().numpy()
    edges = np.linspace(-8.0, 8.0, K + 1)
    for zi in range(nz - 1):
        zend[zi], zcen[zi] = edges[None, 1:-1], ((edges[:-1] + edges[1:]) / 2)[None]
    return zend, zcen


def run(name, xs, nz, zch, reswidth, nblocks, bitswap=1, q=10):
    torch.manual_seed(50)
    model = mg.RefModel(xs=xs, nz=nz, zchannels=zch, nprocessing=4, kernel_size=3, resdepth=8, reswidth=reswidth,
                        root_process=False)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".b"):
                p.add_(torch.randn_like(p) * 0.1)
            if n.endswith("gen_std"):
                p.add_(-3.0)
    model.eval()
    xdim, zdim = int(np.prod(xs)), zch * 256
    zend, zcen = bins(nz, zdim, q)
    images = mg.synth_images(np.random.RandomState(11), nblocks, xs)
    t0 = time.perf_counter()
    ops, sent, restbits, nets, cma = mg.replay_chain(model, zend, zcen, images, nz, bitswap, q, xdim, zdim, cap=0)
    dt = time.perf_counter() - t0
    return {"config": name, "blocks": nblocks, "seconds": round(dt, 2), "pixels_per_s": round(nblocks * 1024 / dt, 1),
            "bits_per_dim": round(float(cma[-1]), 3), "lossless_and_unwound": True}


CONFIGS = {"mnist2": ("configs[0]: MNIST nz=2 Bit-Swap, reswidth 63, Z=256, X=1024", (1, 32, 32), 2, 1, 63, 20),
           "imagenet4": ("configs[2] shape: ImageNet32 nz=4 Bit-Swap, reswidth 254, Z=2048, X=3072", (3, 32, 32), 4, 8, 254, 4),
           "cifar8": ("configs[1] shape: CIFAR-10 nz=8 Bit-Swap, reswidth 252, Z=2048, X=3072", (3, 32, 32), 8, 8, 252, 3)}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="one configuration only (bench.py's same-host leg)")
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--where", default="build container (not the GPU box)",
                    help="label of the host this runs on (bench.py passes 'same host as the GPU run')")
    a = ap.parse_args()
    threads = torch.get_num_threads()
    names = [a.config] if a.config else ["mnist2", "imagenet4", "cifar8"]
    res = []
    for n in names:
        title, xs, nz, zch, w, blocks = CONFIGS[n]
        res.append(run(title, xs, nz, zch, w, a.blocks or blocks))
    out = {"kind": "reference", "what": "fhkingma/bitswap Python path (imported ANS/Model/logistic_cdf), sender + receiver loops "
                                        "of mnist_compress.py:164-358 replayed on CPU, one chain",
           "value": res[-1]["pixels_per_s"], "unit": "pixels/s (encode+decode)", "cores": threads,
           "host": f"{platform.processor() or platform.machine()}, {os.cpu_count()} logical CPUs, torch {torch.__version__} "
                   f"({threads} threads), {a.where}",
           "runs": res, "tool": "tools/ref_cpu_baseline.py"}
    print(json.dumps(out, indent=1))
