#!/usr/bin/env python3
"""bs_wino_gemm_f32 at the column counts of few chains per call (latency-bound launches): microseconds per launch, alone,
back to back on one stream.  `BITSWAP_GEMM_NS3_UNITS` (read once per process) picks the launches that run on three LDS
stages: 0 = never (round 3's double buffer), unset = default rule, a large number = always.
    python tools/gemm_small.py            (prints one JSON line)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402


def t_us(fn, n=200, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = "cuda"
    out = {"ns3_units": os.environ.get("BITSWAP_GEMM_NS3_UNITS", "default")}
    ref = {}
    for (T, Cout, Cin) in ((36, 256, 256), (64, 256, 256), (36, 16, 256)):
        for chains in (1, 13, 50, 100, 200, 500):
            cols = chains * 16
            torch.manual_seed(T + cols)
            U = torch.randn(T, Cout, Cin, device=dev)
            V = torch.randn(T, Cin, cols, device=dev)
            M = torch.empty(T, Cout, cols, device=dev)
            us = t_us(lambda: hip.wino_gemm(U, V, out=M))
            fl = 2.0 * T * Cout * Cin * cols
            out[f"T{T}_co{Cout}_cols{cols}"] = {"us": round(us, 2), "TFLOPs": round(fl / us / 1e6, 1),
                                                "checksum": float(M.double().sum().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
