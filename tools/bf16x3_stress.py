#!/usr/bin/env python3
"""Is bs_wino_gemm_bf16x3 deterministic beside other kernels?  The product alone vs the same product while a second stream
runs (a) the fp32 GEMM, (b) a transform pass, (c) another bf16x3 product -- bitwise comparison over many repetitions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
res = {}
for nprod in (6, 9):
    for (T, C, cols) in ((36, 256, 512), (64, 256, 512), (36, 256, 8000)):
        U = torch.randn(T, C, C, device=dev)
        V = torch.randn(T, C, cols, device=dev)
        Uf = hip.frags_bf16x3(U)
        ref = hip.wino_gemm_bf16x3(Uf, V, nprod).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        U2 = torch.randn(T, C, C, device=dev); V2 = torch.randn(T, C, cols, device=dev); Uf2 = hip.frags_bf16x3(U2)
        x = torch.randn(cols // 16, C, 16, 16, device=dev); b = torch.randn(C, device=dev)
        others = {"fp32_gemm": lambda: hip.wino_gemm(U2, V2), "fused": lambda: hip.wino_fused(x, tuple(x.shape), 0, b, None, True, ts_out=6),
                  "bf16x3": lambda: hip.wino_gemm_bf16x3(Uf2, V2, nprod), "head_gemm": lambda: hip.wino_gemm(U2[:, :16].contiguous(), V2)}
        for name, other in others.items():
            bad = 0
            for rep in range(30):
                with torch.cuda.stream(side):
                    for _ in range(4):
                        other()
                out = hip.wino_gemm_bf16x3(Uf, V, nprod)
                with torch.cuda.stream(side):
                    for _ in range(2):
                        other()
                torch.cuda.synchronize()
                bad += int(not torch.equal(out, ref))
            res[f"nprod{nprod}_T{T}_cols{cols}_beside_{name}"] = bad
            print(f"nprod {nprod} T{T} cols {cols} beside {name}: {bad}/30 runs differ", flush=True)
