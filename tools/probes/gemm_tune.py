"""What PyTorch's TunableOp (exhaustive search over the rocBLAS / hipBLASLt solutions) finds for the Winograd-domain
GEMM shapes of the bench, next to the library's default choice.  Writes the chosen solutions to gpurun_out/."""
import os
import sys
import time

import torch

dev = "cuda"


def bench(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3


shapes = ((36, 256, 6400), (64, 256, 6400), (36, 256, 25600))
ops = {}
for T, C, N in shapes:
    U = torch.randn(T, C, C, device=dev)
    V = torch.randn(T, C, N, device=dev)
    ops[(T, C, N)] = (U, V)
    t = bench(lambda: torch.bmm(U, V))
    print(f"default   T{T} C{C} N{N}: {t * 1e6:8.1f} us {2 * T * C * C * N / t / 1e12:6.1f} TF", flush=True)
import torch.cuda.tunable as tn
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(15)
tn.set_max_tuning_iterations(5)
tn.set_filename("gpurun_out/r02_tunableop.csv")
for (T, C, N), (U, V) in ops.items():
    t0 = time.time()
    torch.bmm(U, V)
    torch.cuda.synchronize()
    print(f"tuned in {time.time() - t0:.1f} s", flush=True)
    t = bench(lambda: torch.bmm(U, V))
    print(f"tunableop T{T} C{C} N{N}: {t * 1e6:8.1f} us {2 * T * C * C * N / t / 1e12:6.1f} TF", flush=True)
tn.write_file()
print(tn.get_results())
