"""Does the serial rANS pop (one wavefront per chain, 400 chains = 400 of 1024 SIMDs, latency bound) run UNDER a GEMM / a
table kernel of another stream?  And do two pops on two streams overlap at all?
usage: python tools/probes/overlap_probe2.py"""
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


def fresh_state(nwords=60000):
    words = np.random.RandomState(1).randint(1 << 16, (1 << 32) - 1, size=(B, nwords), dtype=np.uint32)
    st = hip.RansState(B, nwords + 4 * D, dev)
    st.stack[:, :nwords] = torch.from_numpy(words.view(np.int32)).to(dev)
    st.len.fill_(nwords - 1)
    st.head.copy_(torch.from_numpy((words[:, -1].astype(np.uint64) << np.uint64(32)).view(np.int64)))
    return st


status = torch.zeros(B, dtype=torch.int32, device=dev)
cdfs = [hip.logistic_tables(e, mu, sc, 31, 10, layout=hip.LAYOUT_WAVE, step=step, status=status) for _ in range(2)]
states = [fresh_state(), fresh_state()]
U = torch.randn(36, 256, 256, device=dev)
V = torch.randn(36, 256, 6400, device=dev)
M = torch.empty(36, 256, 6400, device=dev)
wcdf = torch.empty((B, D, hip.wave_ld(K)), dtype=torch.int32, device=dev)


def gemm():
    hip.wino_gemm(U, V, out=M)


def tables():
    hip.logistic_tables(e, mu, sc, 31, 10, out=wcdf, layout=hip.LAYOUT_WAVE, step=step, status=status)


def pop(i):
    return lambda: hip.rans_pop(states[i], cdfs[i], K)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, na, fb=None, nb=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
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


for _ in range(2):
    run(gemm, 20, pop(0), 5)
n = 24                                  # 24 pops of 2048 symbols drain ~ 24 * 2048 * 10 bits of the 60000-word stacks
states[:] = [fresh_state(), fresh_state()]
tp = run(pop(0), n)
states[:] = [fresh_state(), fresh_state()]
tpp = run(pop(0), n, pop(1), n)
ng = 100
tg = run(gemm, ng)
tt = run(tables, 40)
states[:] = [fresh_state(), fresh_state()]
tgp = run(gemm, ng, pop(0), n)
states[:] = [fresh_state(), fresh_state()]
ttp = run(tables, 40, pop(0), n)
print(f"pop alone {tp / n * 1e6:7.1f} us/launch; two pops on two streams x{n} each: {tpp * 1e3:6.1f} ms vs {2 * tp * 1e3:6.1f} serial ({tpp / (2 * tp):.2f})")
print(f"gemm x{ng} ({tg * 1e3:6.1f} ms) + pop x{n} ({tp * 1e3:6.1f} ms): two streams {tgp * 1e3:6.1f} ms ({tgp / (tg + tp):.2f} of serial, max would be {max(tg, tp) / (tg + tp):.2f})")
print(f"tables x40 ({tt * 1e3:6.1f} ms) + pop x{n}: two streams {ttp * 1e3:6.1f} ms ({ttp / (tt + tp):.2f} of serial, max would be {max(tt, tp) / (tt + tp):.2f})")
