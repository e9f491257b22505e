#!/usr/bin/env python3
"""Where the time of the wave-specialised bf16x3 GEMM goes: the kernel with pieces switched off (-DBS_GEMM_LAB build in /tmp, WRONG
results, timing only; BITSWAP_BF16X3_WSLAB bits: 1 producers load nothing, 2 consumers store nothing, 4 consumers load no U
fragments, 8 producers neither split nor write LDS, 16 one MFMA per tile instead of 12, 32 no s_barrier in the loop).
    python tools/gemm_ws_lab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BITSWAP_HIP_LIB"] = "/tmp/libbitswap_hip_lab.so"
from bitswap_amd import build, hip  # noqa: E402
build.HIPCC_FLAGS.append("-DBS_GEMM_LAB")
build.build_hip(force=True)


def t_us(fn, warm=80, reps=120):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


NAMES = {0: "everything on", 1: "no V loads", 2: "no M stores", 3: "no V loads, no M stores", 4: "no U fragment loads",
         7: "no global memory traffic at all", 8: "no split, no LDS writes", 9: "producers idle (no loads, no split)",
         15: "consumers alone: MFMAs + LDS reads + barrier", 16: "one MFMA per tile (1/12 of the multiplies)",
         23: "no global memory, 1/12 of the multiplies", 32: "no s_barrier", 47: "bare MFMAs + LDS reads (no memory, no producers, no barrier)"}
for (T, C, cols) in ((36, 256, 8000), (64, 256, 8000)):
    U = torch.randn(T, C, C, device="cuda")
    V = torch.randn(T, C, cols, device="cuda")
    M = torch.empty(T, C, cols, device="cuda")
    Uf = hip.frags_bf16x3(U)
    os.environ["BITSWAP_BF16X3_SHAPE"] = "3"
    os.environ.pop("BITSWAP_BF16X3_WSLAB", None)
    os.environ["BITSWAP_BF16X3_PERSISTENT"] = "0"
    base = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=M))
    print(f"T{T} x {C} x {C} x {cols}: product kernel (one unit per workgroup) {base:7.1f} us", flush=True)
    for lab in (0, 1, 2, 3, 4, 7, 8, 9, 15, 16, 23, 32, 47):
        os.environ["BITSWAP_BF16X3_WSLAB"] = str(lab)
        t = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=M))
        print(f"   lab {lab:2d} {NAMES[lab]:58s} {t:7.1f} us", flush=True)
    os.environ.pop("BITSWAP_BF16X3_WSLAB", None)
