"""Which BLAS backend serves the 5x5-conv GEMMs best?  [252 x 1260] x [n][1260 x 256] (ld 320), fp32."""
import torch
dev = "cuda"
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
for lib in ("default", "hipblaslt", "rocblas"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "unavailable", e); continue
    for n in (50, 100, 200):
        C = 252
        ax = torch.randn(n, C * 5, 20, 16, device=dev)
        w = torch.randn(C, C * 5, device=dev) * 0.02
        out = torch.empty(n, C, 256, device=dev)
        a = ax[:, :, 1:17, :].flatten(2)
        t = bench(lambda: torch.bmm(w.unsqueeze(0).expand(n, C, C * 5), a, out=out))
        t2 = bench(lambda: out.baddbmm_(w.unsqueeze(0).expand(n, C, C * 5), a))
        # transposed formulation: pixels as rows
        at = a.transpose(1, 2)                      # [n, 256, 1260] view
        wt = w.t().contiguous()                     # [1260, 252]
        t3 = bench(lambda: torch.matmul(at, wt))    # -> [n, 256, 252] (NHWC-like result)
        fl = 2 * n * 256 * C * C * 5
        print(f"{lib:10s} n={n}: bmm {t*1e3:.0f} us ({fl/t/1e9:.0f} TF)  baddbmm_ {t2*1e3:.0f} us ({fl/t2/1e9:.0f} TF)  transposed matmul {t3*1e3:.0f} us ({fl/t3/1e9:.0f} TF)", flush=True)
