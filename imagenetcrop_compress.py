#!/usr/bin/env python3
"""imagenetcrop_compress.py -- unscaled ImageNet images cropped to multiples of 32 and coded as
grids of 32x32 blocks, one rANS chain per image (reference: imagenetcrop_compress.py:215-314),
Bit-Swap and BB-ANS.  Only --gpu in the reference; extras: --images DIR|.npy, --nimages, --synthetic.
Images are independent chains: under torchrun they are sharded over the GPUs by block count."""
import argparse
import os
import time

import numpy as np

from bitswap_amd import cli, dist, tiling


def synthetic_images(n, seed=100):
    """H, W ~ U{256..512} smooth RGB images."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        h, w = rng.randint(256, 513, size=2)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 90 * np.sin(yy / rng.uniform(8, 40) + rng.uniform(0, 6)) * np.cos(xx / rng.uniform(8, 40))
                        for _ in range(3)], -1) + rng.randn(h, w, 3) * 6
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return out


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpu', default=0, type=int)
    parser.add_argument('--images', default='model/data/imagenetfull/test/class')
    parser.add_argument('--nimages', default=100, type=int)
    parser.add_argument('--synthetic', action='store_true')
    parser.add_argument('--params', default=None)
    args = parser.parse_args()
    print(args)
    rank, world = dist.init()
    np.random.seed(100)
    if args.synthetic or not os.path.exists(args.images):
        if not args.synthetic:
            raise SystemExit(f"{args.images} not found -- pass --images or --synthetic")
        imgs = synthetic_images(args.nimages)
    else:
        from PIL import Image
        files = sorted(f for f in os.listdir(args.images) if os.path.isfile(os.path.join(args.images, f)))
        pick = np.random.choice(len(files), size=min(len(files), 10 * args.nimages), replace=False)
        imgs = []
        for i in pick:
            a = np.asarray(Image.open(os.path.join(args.images, files[i])))
            if a.ndim == 3 and a.shape[-1] == 3:          # skip non-RGB images like the reference (:228)
                imgs.append(a.astype(np.uint8))
            if len(imgs) == args.nimages:
                break
    blocks = [tiling.extract_blocks(a)[0] for a in imgs]
    mine = dist.shard_chains(len(blocks), world, rank, weights=[len(b) for b in blocks])
    gpu = args.gpu if world == 1 else rank
    setup = cli.crop_setup(gpu, nz=4, quantbits=10, synthetic=args.synthetic, params=args.params)
    res = {}
    nblk = sum(len(blocks[i]) for i in mine)
    for name, bs in (("bbans", 0), ("bitswap", 1)):
        t0 = time.perf_counter()
        out = cli.compress_images([blocks[i] for i in mine], quantbits=10, nz=4, bitswap=bs, gpu=gpu,
                                  hwc_quirk=True, setup=setup)
        dt = time.perf_counter() - t0        # compress_images ends with a device->host copy of the streams
        bpd = dist.gather_rows(np.array([[o[2]] for o in out]), mine, len(blocks))
        tot = dist.allreduce_sum([float(nblk)])
        if rank == 0:
            res[name] = bpd[:, 0]
            print(f"{name}: rank 0 coded {len(mine)} images / {nblk} blocks in {dt:.2f} s "
                  f"({nblk * 1024 / dt:.0f} pixels/s sender, this rank; {int(tot[0])} blocks on {world} GPU(s))")
    if rank == 0:
        print(f"bbans: {np.mean(res['bbans']):.2f} bits/dim")
        print(f"bitswap: {np.mean(res['bitswap']):.2f} bits/dim")
