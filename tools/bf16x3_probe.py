#!/usr/bin/env python3
"""bs_wino_gemm_bf16x3 (three bf16 limbs per operand, 6 or 9 limb products per k block) against bs_wino_gemm_f32 and a float64
product: error (max |d| / output range, rms / rms) and time at the bench's shapes.   python tools/bf16x3_probe.py [--quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402


def t_ms(fn, warm=30, reps=60):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = "cuda"
    out = {}
    shapes = [(36, 256, 256, 1600), (36, 256, 256, 8000), (64, 256, 256, 8000), (36, 256, 64, 528), (3, 96, 48, 100), (36, 256, 256, 208)]
    if "--quick" in sys.argv:
        shapes = shapes[:1] + shapes[3:5]
    for (T, Cout, Cin, cols) in shapes:
        torch.manual_seed(T * 1000 + cols)
        # weights ~ Winograd-domain filters, activations ~ transformed ELU outputs: wide dynamic range on purpose
        U = (torch.randn(T, Cout, Cin, device=dev) * torch.exp(torch.randn(T, 1, Cin, device=dev))).contiguous()
        V = (torch.randn(T, Cin, cols, device=dev) * torch.exp(0.5 * torch.randn(T, Cin, 1, device=dev))).contiguous()
        Ul = hip.split_bf16x3(U)
        assert torch.equal(Ul[0].float() + Ul[1].float() + Ul[2].float(), U), "limbs do not add up to U"
        Ul = hip.frags_bf16x3(U)
        ref = torch.bmm(U.double(), V.double())
        rng = float(ref.abs().max())
        rms = float(ref.pow(2).mean().sqrt())
        res = {}
        m32 = hip.wino_gemm(U, V)
        for name, fn in (("fp32_mfma", lambda: hip.wino_gemm(U, V)), ("bf16x3_6", lambda: hip.wino_gemm_bf16x3(Ul, V, 6)),
                         ("bf16x3_9", lambda: hip.wino_gemm_bf16x3(Ul, V, 9))):
            m = fn()
            d = (m.double() - ref)
            again = fn()
            ms = t_ms(fn) if cols >= 1000 or "--all-times" in sys.argv else None
            fl = 2.0 * T * Cout * Cin * cols
            res[name] = {"max_err_over_range": float(d.abs().max()) / rng, "rms_err_over_rms": float(d.pow(2).mean().sqrt()) / rms,
                         "repeatable": bool(torch.equal(m, again)), "ms": None if ms is None else round(ms, 4),
                         "TFLOPs_equiv": None if ms is None else round(fl / ms / 1e9, 1)}
            if name != "fp32_mfma":
                # batch invariance: a prefix of the columns gives the same bits
                sub = (cols // 2) // 4 * 4
                ms_ = hip.wino_gemm_bf16x3(Ul, V[:, :, :sub].contiguous(), int(name[-1]))
                res[name]["prefix_columns_same_bits"] = bool(torch.equal(ms_, m[:, :, :sub]))
        out[f"T{T}_{Cout}x{Cin}x{cols}"] = res
        print(f"T{T} {Cout}x{Cin}x{cols}", json.dumps(res), flush=True)
    json.dump(out, open(os.environ.get("BF16X3_OUT", "/tmp/bf16x3_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
