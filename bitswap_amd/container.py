"""On-disk formats of the reference.

* demo container (demo_compress.py:159-160,272-283; demo_decompress.py:221-227): a uint32 .npy
  `[stack words that were touched ..., head_lo, head_hi, nblocks, h, w]`; initial words that were
  never popped are trimmed (`del state[0:excess_state_len - 1]`).
* experiment bitstreams (mnist_compress.py:265-272): the Python list `state` pickled as is.
* 64-state container (opt-in BS_FORMAT_WAVE64 streams, no reference counterpart): a uint32 .npy
  `[MAGIC64, version, 64, (fingerprint,) n_0 .. n_63, kept words of state 0, ..., of state 63, (head_lo, head_hi) x 64,
  nblocks, h, w]`; the same trimming rule as the reference's container, applied per state.  Version 2 (round 3) carries
  the CRC-32 of the stream fingerprint (bitswap_amd/meta.py) after the state count, so a receiver with another CDF
  specification or conv route refuses the stream instead of decoding garbage; version 1 files are still read.
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


MAGIC64 = 0x36575342     # "BSW6"
VERSION64 = 2


def pack64(state, min_words, nblocks, h, w, fingerprint=0):
    """state: 64 sub-state lists [w0 .. w_{n-1}, head]; min_words: per state the fewest stack words it ever held;
    fingerprint: meta.word() of the sender's codec."""
    assert len(state) == 64 and len(min_words) == 64
    kept = [list(sub[:-1][m:]) for sub, m in zip(state, min_words)]
    out = [MAGIC64, VERSION64, 64, int(fingerprint) & 0xFFFFFFFF] + [len(k) for k in kept]
    for k in kept:
        out += k
    for sub in state:
        out += [sub[-1] & 0xFFFFFFFF, sub[-1] >> 32]
    return np.array(out + [nblocks, h, w], dtype=np.uint32)


def _hdr64(arr):
    """Header words in front of the 64 state lengths: 3 (version 1) or 4 (version 2: + fingerprint)."""
    return 4 if int(arr[1]) >= 2 else 3


def is_pack64(arr):
    arr = np.asarray(arr)
    if arr.dtype != np.uint32 or len(arr) < 3 + 64 + 128 + 3 or int(arr[0]) != MAGIC64 or int(arr[2]) != 64:
        return False
    h = _hdr64(arr)
    return len(arr) >= h + 64 and h + 64 + int(arr[h:h + 64].astype(np.int64).sum()) + 128 + 3 == len(arr)


def fingerprint64(arr):
    """The fingerprint word of a 64-state container, or None (version 1)."""
    arr = np.asarray(arr)
    return int(arr[3]) if is_pack64(arr) and int(arr[1]) >= 2 else None


def unpack64(arr):
    """-> (64 sub-state lists, nblocks, h, w)"""
    arr = np.asarray(arr)
    if not is_pack64(arr):
        raise ValueError("not a 64-state Bit-Swap container")
    if int(arr[1]) not in (1, 2):
        raise ValueError(f"64-state container version {int(arr[1])} not supported")
    vals = [int(v) for v in arr.tolist()]
    h0 = _hdr64(arr)
    ns = vals[h0:h0 + 64]
    off = h0 + 64
    subs = []
    for n in ns:
        subs.append(vals[off: off + n])
        off += n
    for j in range(64):
        subs[j].append(vals[off + 2 * j] | (vals[off + 2 * j + 1] << 32))
    nblocks, h, w = vals[-3:]
    return subs, nblocks, h, w


def save_state(path, state):
    with open(path, "wb") as fp:
        pickle.dump(state, fp)


def load_state(path):
    with open(path, "rb") as fp:
        return pickle.load(fp)
