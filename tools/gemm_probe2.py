"""Batched GEMM shapes of Winograd-domain convolutions: U[t] [252 x 252] x V[t] [252 x tiles]."""
import torch
dev = "cuda"
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
C = 252
for n in (100, 200):
    for name, T, tiles, direct in (("F(4x4,3x3)", 36, 16, 9), ("F(2x2,3x3)", 16, 64, 9), ("F(2x2,5x5)", 36, 64, 25), ("F(4x4,5x5)", 64, 16, 25)):
        N = n * tiles
        u = torch.randn(T, C, C, device=dev) * 0.02
        v = torch.randn(T, C, N, device=dev)
        out = torch.empty(T, C, N, device=dev)
        t = bench(lambda: torch.bmm(u, v, out=out))
        fl = 2 * T * C * C * N
        dfl = 2 * n * 256 * C * C * direct
        print(f"n={n} {name}: bmm batch {T} [252x252]x[252x{N}] {t*1e3:.0f} us ({fl/t/1e9:.0f} TF real, {dfl/t/1e9:.0f} TF conv-equivalent); operand {T*C*N*4/1e6:.0f} MB", flush=True)
