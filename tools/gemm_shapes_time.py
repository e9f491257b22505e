"""bs_wino_gemm_bf16x3 alone, launch shape by launch shape (environment switches of wino_gemm_bf16x3.hip), at the sizes of the
conv stacks; every shape is held to the bits of the first.  usage: python tools/gemm_shapes_time.py [name=SHAPE,PERSISTENT,ULDS ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402

VARIANTS = [a.split("=") for a in sys.argv[1:] if "=" in a] or [["ws", "3,0,0"], ["ws_persistent", "3,1,0"]]
SIZES = [(36, 256, 256, 8000), (64, 256, 256, 8000), (36, 256, 256, 12000), (36, 256, 256, 1600), (64, 256, 256, 1600), (36, 256, 256, 400),
         (36, 256, 256, 208)]


def t_us(fn, warm=60, reps=100):
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


for (T, Cout, Cin, cols) in SIZES:
    torch.manual_seed(T + cols)
    U = (torch.randn(T, Cout, Cin, device="cuda") * torch.exp(torch.randn(T, 1, Cin, device="cuda"))).contiguous()
    V = (torch.randn(T, Cin, cols, device="cuda") * torch.exp(0.5 * torch.randn(T, Cin, 1, device="cuda"))).contiguous()
    Uf = hip.frags_bf16x3(U)
    out = torch.empty(T, Cout, cols, device="cuda")
    fl = 2.0 * T * Cout * Cin * cols
    row, ref = {}, None
    for name, v in VARIANTS:
        shape, pers, ulds = v.split(",")
        os.environ["BITSWAP_BF16X3_SHAPE"], os.environ["BITSWAP_BF16X3_PERSISTENT"], os.environ["BITSWAP_BF16X3_ULDS"] = shape, pers, ulds
        m = hip.wino_gemm_bf16x3(Uf, V, 6).clone()
        ref = m if ref is None else ref
        assert torch.equal(m, ref), name
        row[name] = t_us(lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=out))
    print(f"T{T} {Cout}x{Cin}x{cols}: " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:6.1f} TF-eq)" for k, v in row.items()), flush=True)
