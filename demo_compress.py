#!/usr/bin/env python3
"""demo_compress.py -- compress one image with Bit-Swap (nz=4, 10 bits) into <name>_bitswap.npy,
the reference's container format (demo_compress.py:159-160,272-283).  Interactive like the
reference, or non-interactive with --image/--gpu."""
import argparse
import os
import sys

import numpy as np

from bitswap_amd import cli, container, meta, tiling


def ask(prompt):
    sys.stdout.write(prompt)
    sys.stdout.flush()
    return input()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--image', default=None)
    ap.add_argument('--gpu', default=None, type=int)
    ap.add_argument('--synthetic', action='store_true', help="seeded synthetic weights (no checkpoint offline)")
    ap.add_argument('--params', default=None)
    ap.add_argument('--format', default="reference", choices=["reference", "wave64"],
                    help="reference: the reference's stream/container; wave64: the opt-in 64-state format (own container)")
    ap.add_argument('--cdf-spec', default=None, type=int, choices=list(meta.CDF_SPECS), help="deterministic CDF specification (see include/bitswap_hip.h)")
    args = ap.parse_args()
    if args.gpu is None:
        print("Give GPU index (0, 1, 2 etc.).")
        args.gpu = int(ask("Index: "))
    path = args.image or ask("Image path: ")
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB")).astype('uint8')
    d, file = os.path.split(os.path.abspath(path))
    filename = os.path.splitext(file)[0]
    cli.seed_everything()
    old_h, old_w, _ = img.shape
    blocks, h, w = tiling.extract_blocks(img, block_size=(32, 32))
    uncompressed = tiling.unextract_blocks(blocks, h, w)
    np.save(os.path.join(d, f"{filename}_uncompressed"), uncompressed)
    size_raw = os.path.getsize(os.path.join(d, f"{filename}_uncompressed.npy")) * 8
    print(f"Shape ({old_h}, {old_w}, 3) -> cropped to ({h}, {w}, 3); raw size {size_raw} bits")
    setup = cli.crop_setup(args.gpu, nz=4, quantbits=10, synthetic=args.synthetic, params=args.params)
    res = cli.compress_images([blocks], quantbits=10, nz=4, bitswap=1, gpu=args.gpu, setup=setup, fmt=args.format,
                              cdf_spec=args.cdf_spec)
    state, min_words, bpd = res[0]
    if args.format == "wave64":
        arr = container.pack64(state, min_words, blocks.shape[0], h, w, fingerprint=meta.word(res.fingerprint))
    else:
        arr = container.pack(state, min_words, blocks.shape[0], h, w)      # the reference's layout: no room for metadata
    # what demo_decompress.py must reproduce (stream format, CDF specification, conv route): the sidecar it checks
    meta.save(os.path.join(d, f"{filename}_bitswap.meta.json"), res.fingerprint, nblocks=int(blocks.shape[0]), image=file)
    np.save(os.path.join(d, f"{filename}_bitswap"), arr)
    size_bs = os.path.getsize(os.path.join(d, f"{filename}_bitswap.npy")) * 8
    print(f"Bit-Swap: {filename}_bitswap.npy, {size_bs} bits, ratio {100 * size_bs / size_raw:.2f} %, "
          f"savings {100 - 100 * size_bs / size_raw:.2f} %")
