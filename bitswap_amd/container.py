"""On-disk formats of the reference.

* demo container (demo_compress.py:159-160,272-283; demo_decompress.py:221-227): a uint32 .npy
  `[stack words that were touched ..., head_lo, head_hi, nblocks, h, w]`; initial words that were
  never popped are trimmed (`del state[0:excess_state_len - 1]`).
* experiment bitstreams (mnist_compress.py:265-272): the Python list `state` pickled as is.
"""
import pickle

import numpy as np


def pack(state, min_words, nblocks, h, w):
    """state: Python list [w0..w_{n-1}, head]; min_words: fewest stack words the chain held at any
    point while coding (= excess_state_len - 1 of the reference)."""
    words, head = state[:-1], state[-1]
    out = list(words[min_words:]) + [head & 0xFFFFFFFF, head >> 32, nblocks, h, w]
    return np.array(out, dtype=np.uint32)


def unpack(arr):
    """-> (state list, nblocks, h, w)"""
    arr = np.asarray(arr)
    if arr.dtype != np.uint32:
        raise ValueError("State streams must be 32 bits long.")  # demo_decompress.py:177-179
    vals = [int(v) for v in arr.tolist()]
    w, h, nblocks = vals.pop(), vals.pop(), vals.pop()
    hi, lo = vals.pop(), vals.pop()
    vals.append(hi << 32 | lo)
    return vals, nblocks, h, w


def save_state(path, state):
    with open(path, "wb") as fp:
        pickle.dump(state, fp)


def load_state(path):
    with open(path, "rb") as fp:
        return pickle.load(fp)
