#!/usr/bin/env python3
"""Sender and receiver in DIFFERENT processes (tests/test_codec_gpu.py::test_sender_and_receiver_in_separate_processes):
    python tools/xproc_codec.py enc <file.npz>     # codes seeded synthetic blocks, stores the streams
    python tools/xproc_codec.py dec <file.npz>     # rebuilds model + bins from the same seeds, decodes, verifies
Everything the receiver knows comes from the file (streams, shapes) and from the seeds: the conv stacks' outputs must
come out bit-identical in a fresh process (MIOpen algorithm choice, BLAS heuristics, Winograd-domain GEMMs)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import workload  # noqa: E402
from bitswap_amd.codec import BitSwapCodec, initial_states  # noqa: E402

B, N, WORKLOAD, Q, WIDTH = 26, 2, "cifar8", 10, 40


def setup():
    model, zend, zcen = workload.build(WORKLOAD, "cuda", quantbits=Q, small=WIDTH)
    return model, BitSwapCodec(model, zend, zcen, quantbits=Q, bitswap=True)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    model, codec = setup()
    images = workload.synthetic_blocks(B * N, model.xs, seed=123).view(B, N, -1).to(torch.int32)
    if mode == "enc":
        state, met = codec.compress(images.cuda())
        np.savez(path, stack=state.stack.cpu().numpy(), len=state.len.cpu().numpy(), head=state.head.cpu().numpy())
        print("encoded", float(met["cma"][:, -1].mean()))
    else:
        f = np.load(path)
        state = codec.new_states(B, N)
        assert state.stack.shape == f["stack"].shape
        state.stack.copy_(torch.from_numpy(f["stack"])); state.len.copy_(torch.from_numpy(f["len"])); state.head.copy_(torch.from_numpy(f["head"]))
        out = codec.decompress(state, N)
        ok = bool(torch.equal(out.cpu(), images)) and state.to_lists() == initial_states(B)
        print("decoded ok" if ok else "DECODE MISMATCH")
        sys.exit(0 if ok else 1)
