"""bs_wino_gemm_f32 against the library (torch.bmm on the backend the model uses) for the batched shapes of the bench:
sustained loops, TFLOP/s.  usage: python tools/gemm_probe.py"""
import torch

from bitswap_amd import hip


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


try:
    torch.backends.cuda.preferred_blas_library("ck")
except Exception as e:   # noqa
    print("ck backend unavailable", e)
for T, Cout, Cin, cols in ((36, 256, 256, 6400), (36, 256, 256, 1600), (64, 256, 256, 6400), (64, 256, 256, 1600),
                           (36, 256, 256, 25600), (36, 256, 256, 208), (36, 16, 256, 6400), (36, 16, 256, 1600)):
    U = torch.randn(T, Cout, Cin, device="cuda")
    V = torch.randn(T, Cin, cols, device="cuda")
    out = torch.empty(T, Cout, cols, device="cuda")
    fl = 2.0 * T * Cout * Cin * cols
    t1 = timeit(lambda: hip.wino_gemm(U, V, out=out))
    t2 = timeit(lambda: torch.bmm(U, V, out=out))
    print(f"T{T} Cout{Cout} Cin{Cin} cols{cols}: own {t1 * 1e6:8.1f} us {fl / t1 / 1e12:6.1f} TF | library {t2 * 1e6:8.1f} us {fl / t2 / 1e12:6.1f} TF", flush=True)
