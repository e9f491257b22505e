#!/usr/bin/env python3
"""cifar_compress.py -- same flags and outputs as the reference script of this name
(--gpu --nz --quantbits --bitswap); the work is done by bitswap_amd (HIP kernels, batched chains)."""
from bitswap_amd.cli import dataset_main

if __name__ == '__main__':
    dataset_main("cifar", default_nz=8)
