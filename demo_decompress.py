#!/usr/bin/env python3
"""demo_decompress.py -- inverse of demo_compress.py (reference: demo_decompress.py:150-244): reads
<name>_bitswap.npy, decodes the blocks, checks them against <name>_uncompressed.npy and writes
<name>_recovered.png."""
import argparse
import os
import sys

import numpy as np

from bitswap_amd import cli, container, meta, tiling

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--file', default=None)
    ap.add_argument('--gpu', default=None, type=int)
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--params', default=None)
    ap.add_argument('--cdf-spec', default=None, type=int, choices=list(meta.CDF_SPECS),
                    help="deterministic CDF specification the stream was written with (1: streams from before round 2)")
    args = ap.parse_args()
    if args.gpu is None:
        print("Give GPU index (0, 1, 2 etc.).")
        sys.stdout.write("Index: ")
        args.gpu = int(input())
    if args.file is None:
        sys.stdout.write("Compressed file path: ")
        args.file = input()
    d, file = os.path.split(os.path.abspath(args.file))
    filename, ext = os.path.splitext(file)
    if ext != ".npy" or "_bitswap" not in filename:
        raise SystemExit("Expected a <name>_bitswap.npy file")
    cli.seed_everything()
    arr = np.load(args.file)
    state, nblocks, h, w = (container.unpack64 if container.is_pack64(arr) else container.unpack)(arr)
    setup = cli.crop_setup(args.gpu, nz=4, quantbits=10, synthetic=args.synthetic, params=args.params)
    side = os.path.join(d, f"{filename}.meta.json")
    expect = meta.load(side) if os.path.exists(side) else None
    if expect is None and not container.is_pack64(arr):
        print(f"no {os.path.basename(side)} next to the container (the reference writes none): decoding with this build's "
              f"defaults, CDF spec {args.cdf_spec or meta.DEFAULT_CDF_SPEC}; a stream written with other settings decodes to noise")
    blocks, _ = cli.decompress_image(state, nblocks, quantbits=10, nz=4, gpu=args.gpu, setup=setup, expect=expect,
                                     expect_word=container.fingerprint64(arr), cdf_spec=args.cdf_spec)
    img = tiling.unextract_blocks(blocks, h, w)
    ref = os.path.join(d, f"{filename.replace('_bitswap', '_uncompressed')}.npy")
    if os.path.exists(ref):
        assert np.all(img == np.load(ref)), "decompressed image differs from the uncompressed crop"
        print("matches the uncompressed crop bit for bit")
    from PIL import Image
    out = os.path.join(d, f"{filename.replace('_bitswap', '_recovered')}.png")
    Image.fromarray(img).save(out)
    print(f"Reconstructed image as {os.path.basename(out)}")
