"""32x32 block tiling of arbitrary-size images (the reference's benchmark_compress.py:20-39).

extract_blocks crops the image to multiples of the block size (dropping bottom/right remainders)
and returns the blocks in row-major grid order; unextract_blocks is its inverse on the crop.
"""
import numpy as np


def extract_blocks(arr, block_size=(32, 32)):
    """HWC uint8 image -> ([n, bh, bw, C] blocks, cropped h, cropped w)."""
    bh, bw = block_size
    h, w, c = arr.shape
    h, w = h - h % bh, w - w % bw
    grid = arr[:h, :w].reshape(h // bh, bh, w // bw, bw, c)
    return grid.transpose(0, 2, 1, 3, 4).reshape(-1, bh, bw, c), h, w


def unextract_blocks(arr, h, w):
    """[n, bh, bw, C] blocks -> HWC image of the cropped size (h, w)."""
    n, bh, bw, c = arr.shape
    grid = arr.reshape(h // bh, w // bw, bh, bw, c)
    return grid.transpose(0, 2, 1, 3, 4).reshape(h, w, c)


def blocks_to_chw_flat(blocks):
    """[n, 32, 32, C] -> [n, C*32*32] in CHW order (ToTensor order, demo_compress.py:120)."""
    return np.ascontiguousarray(blocks.transpose(0, 3, 1, 2)).reshape(blocks.shape[0], -1)


def chw_flat_to_blocks(flat, c=3, bh=32, bw=32):
    """inverse of blocks_to_chw_flat (demo_decompress.py:146-147)."""
    return np.ascontiguousarray(np.asarray(flat).reshape(-1, c, bh, bw).transpose(0, 2, 3, 1)).astype(np.uint8)


def blocks_to_hwc_flat(blocks):
    """[n, 32, 32, C] -> [n, 32*32*C] in HWC order: what imagenetcrop_compress.py:129-130 feeds the
    model (it views the raw HWC block as CHW -- a quirk of the reference kept for bits/dim parity)."""
    return np.ascontiguousarray(blocks).reshape(blocks.shape[0], -1)
