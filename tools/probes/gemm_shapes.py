"""Library GEMM throughput for the Winograd-domain shapes (batch 36 / 64 of [C x C] x [C x tiles]) per BLAS backend,\nC = 252 against the padded 256, and the transposed form.  Output of one run: profiles/r02m_gemm_shapes.txt."""
import torch, time, warnings
warnings.simplefilter("ignore")
dev="cuda"
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/it*1e-3
for be in ("default","ck","hipblaslt","hipblas"):
    try:
        prev=torch.backends.cuda.preferred_blas_library()
        if be!="default": torch.backends.cuda.preferred_blas_library(be)
    except Exception as e:
        print(be,"unavailable",e); continue
    for T,C,N in ((36,252,6400),(36,256,6400),(64,252,6400),(36,252,12800),(36,252,208)):
        U=torch.randn(T,C,C,device=dev); V=torch.randn(T,C,N,device=dev)
        t=bench(lambda: torch.bmm(U,V))
        Vt=V.transpose(1,2).contiguous(); Ut=U.transpose(1,2).contiguous()
        t2=bench(lambda: torch.bmm(Vt,Ut))
        fl=2*T*C*C*N
        print(f"{be:10s} T{T} C{C} N{N}: U@V {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF | V^T@U^T {t2*1e6:8.1f} us {fl/t2/1e12:6.1f} TF")
    try: torch.backends.cuda.preferred_blas_library(prev)
    except Exception: pass
