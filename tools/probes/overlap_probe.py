"""Do the float64 table kernel (VALU issue bound) and the fp32 MFMA GEMM overlap when they run on two streams?
Both alone, then together for the same number of launches each; "serial" = sum of the alone times.
usage: python tools/probes/overlap_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bitswap_amd import hip  # noqa: E402
from bitswap_amd.bins import uniform_step  # noqa: E402

dev = "cuda"
rng = np.random.RandomState(0)
B, D, K = 400, 2048, 1024
lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
e_np = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
e = torch.from_numpy(e_np).to(dev)
step = torch.from_numpy(uniform_step(e_np)).to(dev)
mu = torch.from_numpy((rng.randn(B, D) * 0.5).astype(np.float32)).to(dev)
sc = torch.from_numpy(rng.uniform(0.1, 1.0, (B, D)).astype(np.float32)).to(dev)
wcdf = torch.empty((B, D, hip.wave_ld(K)), dtype=torch.int32, device=dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
U = torch.randn(36, 256, 256, device=dev)
V = torch.randn(36, 256, 6400, device=dev)
M = torch.empty(36, 256, 6400, device=dev)
x = torch.randn(400, 256, 16, 16, device=dev)
bias = torch.randn(256, device=dev)


pcdf = torch.empty((B, D, hip.PIVOT_LD), dtype=torch.int32, device=dev)
PIVOT = "--pivot" in sys.argv      # the big-batch hand-off: 64 cumulative values per row, no 3.57 GB row write


def tables():
    if PIVOT:
        hip.logistic_tables(e, mu, sc, 31, 10, out=pcdf, layout=hip.LAYOUT_PIVOT, step=step, status=status)
    else:
        hip.logistic_tables(e, mu, sc, 31, 10, out=wcdf, layout=hip.LAYOUT_WAVE, step=step, status=status)


def gemm():
    hip.wino_gemm(U, V, out=M)


def fused():
    hip.wino_fused(M, (400, 256, 16, 16), 6, bias, x, True, ts_out=6)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, na, fb=None, nb=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # interleave the enqueues so neither stream runs dry
    ia = ib = 0
    while ia < na or ib < nb:
        if ia < na:
            with torch.cuda.stream(s1):
                fa()
            ia += 1
        if fb is not None and ib * na < ia * nb:
            with torch.cuda.stream(s2):
                fb()
            ib += 1
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for per_cu in ("", "1"):
    if per_cu:
        os.environ["BITSWAP_GEMM_WGS_PER_CU"] = per_cu
    for _ in range(2):
        run(gemm, 50, tables, 20)
    ng, nt_, nf = 400, 160, 400
    tg = run(gemm, ng)
    tt = run(tables, nt_)
    tf = run(fused, nf)
    both = run(gemm, ng, tables, nt_)
    gf = run(gemm, ng, fused, nf)
    tfu = run(tables, nt_, fused, nf)
    print(f"GEMM wgs/cu={per_cu or 2}: gemm alone {tg / ng * 1e6:7.1f} us/launch, tables alone {tt / nt_ * 1e6:7.1f} us, fused alone {tf / nf * 1e6:6.1f} us")
    print(f"   gemm x{ng} + tables x{nt_}: serial {1e3 * (tg + tt):7.1f} ms, two streams {1e3 * both:7.1f} ms ({both / (tg + tt):.2f})")
    print(f"   gemm x{ng} + fused  x{nf}: serial {1e3 * (tg + tf):7.1f} ms, two streams {1e3 * gf:7.1f} ms ({gf / (tg + tf):.2f})")
    print(f"   tables x{nt_} + fused x{nf}: serial {1e3 * (tt + tf):7.1f} ms, two streams {1e3 * tfu:7.1f} ms ({tfu / (tt + tf):.2f})", flush=True)
    os.environ.pop("BITSWAP_GEMM_WGS_PER_CU", None)
