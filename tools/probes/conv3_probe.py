"""Time bs_conv3_wino_f32 alone at the bench's shape (400 blocks, Cin 8 -> 256 channels, 16x16 planes) against the
route it replaces (MIOpen conv + k_wino_fused<0, 6>).  BITSWAP_CONV3_CPB forces the channels per block.
usage: python tools/probes/conv3_probe.py [N] [C]"""
import os
import sys

import torch

from bitswap_amd import hip


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn((N, 8, 16, 16), generator=g).cuda()
    w = (torch.randn((C, 8, 3, 3), generator=g) / 8).cuda()
    b = torch.randn(C, generator=g).cuda()
    out = {"cpb": os.environ.get("BITSWAP_CONV3_CPB", "auto"), "N": N, "C": C}
    for act, want in ((3, True), (1, True), (0, True), (3, False)):
        out[f"fused_act{act}_h{int(want)}_us"] = round(timeit(lambda: hip.conv3_wino(x, w, b, act, want, 6)), 1)
    torch.backends.cudnn.deterministic = True
    conv = lambda: torch.nn.functional.conv2d(x, w, None, padding=1)
    out["miopen_conv_us"] = round(timeit(conv), 1)
    c = conv()
    out["wino_fused_0_6_us"] = round(timeit(lambda: hip.wino_fused(c, (N, C, 16, 16), 0, b, None, 3, want_act=True, ts_out=6)), 1)
    print(out, flush=True)


if __name__ == "__main__":
    main()
