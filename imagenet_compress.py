#!/usr/bin/env python3
"""imagenet_compress.py -- ImageNet 32x32 experiments; like the reference (imagenet_compress.py:382)
it ignores --nz and runs nz = 2 and 4."""
from bitswap_amd.cli import dataset_main

if __name__ == '__main__':
    dataset_main("imagenet", default_nz=2, nz_loop=[2, 4])
