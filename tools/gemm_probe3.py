"""The Winograd-domain GEMM U[t] [252x252] x V[t] [252xN] at batch 36 vs 64, per BLAS backend."""
import torch
dev = "cuda"
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
C = 252
for lib in ("default", "hipblaslt", "hipblas", "ck"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "unavailable:", str(e)[:80]); continue
    for T in (36, 64):
        for N in (3200, 6400, 12800):
            try:
                u = torch.randn(T, C, C, device=dev) * 0.02; v = torch.randn(T, C, N, device=dev); out = torch.empty(T, C, N, device=dev)
                t = bench(lambda: torch.bmm(u, v, out=out))
                print(f"{lib:10s} batch {T} N={N}: {t*1e3:.0f} us ({2*T*C*C*N/t/1e9:.0f} TF)", flush=True)
            except Exception as e:
                print(lib, T, N, "failed:", str(e)[:80])
# padded channels 256
torch.backends.cuda.preferred_blas_library("default")
for T in (36,):
    for N in (6400,):
        u = torch.randn(T, 256, 256, device=dev) * 0.02; v = torch.randn(T, 256, N, device=dev); out = torch.empty(T, 256, N, device=dev)
        t = bench(lambda: torch.bmm(u, v, out=out))
        print(f"C=256 batch {T} N={N}: {t*1e3:.0f} us ({2*T*256*256*N/t/1e9:.0f} TF)", flush=True)
