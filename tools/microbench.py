#!/usr/bin/env python3
"""Kernel-level timings of the hot path at BASELINE sizes (run on the GPU box)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=100)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda"
    rng = np.random.RandomState(0)
    res = {}
    for (name, D, K, q) in (("z", 2048, 1024, 10), ("x", 3072, 256, 8)):
        B = args.B
        lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
        e = torch.from_numpy(np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])).to(dev)
        mu = torch.from_numpy((rng.randn(B, D) * 0.5).astype(np.float32)).to(dev)
        sc = torch.from_numpy(rng.uniform(0.1, 1.0, (B, D)).astype(np.float32)).to(dev)
        np.random.seed(1)
        words = np.random.randint(1 << 16, (1 << 32) - 1, size=(B, 10000), dtype=np.uint32)
        st = hip.RansState(B, 10000 + 4 * D, dev)
        st.stack[:, :10000] = torch.from_numpy(words.view(np.int32)).to(dev)
        st.len.fill_(9999)
        st.head.copy_(torch.from_numpy((words[:, -1].astype(np.uint64) << np.uint64(32)).view(np.int64)))
        ld = hip.aligned_ld(K)
        cdf = torch.empty((B, D, ld), dtype=torch.int32, device=dev)
        t_tab = timeit(lambda: hip.logistic_tables(e, mu, sc, 31, q, out=cdf), args.iters)
        sym, _ = hip.rans_pop(st, cdf, K)
        fo = (torch.empty((B, D), dtype=torch.int32, device=dev), torch.empty((B, D), dtype=torch.int32, device=dev))
        t_fc = timeit(lambda: hip.logistic_fc(e, mu, sc, sym, st.status, 31, q, out=fo), args.iters)

        def poppush():
            s2, _ = hip.rans_pop(st, cdf, K)
            hip.rans_push(st, fo[0], fo[1])
        t_pp = timeit(poppush, args.iters)
        t_pop = timeit(lambda: (hip.rans_pop(st, cdf, K), hip.rans_push(st, fo[0], fo[1]))[0], 1, 0)  # placeholder
        # separate pop / push timings (state returns to the start after each pair)
        def only_pop():
            hip.rans_pop(st, cdf, K)
        def only_push():
            hip.rans_push(st, fo[0], fo[1])
        tp = []
        tq = []
        for _ in range(args.iters):
            tp.append(timeit(only_pop, 1, 0))
            tq.append(timeit(only_push, 1, 0))
        # wave-native hand-off layout
        wcdf = torch.empty((B, D, hip.wave_ld(K)), dtype=torch.int32, device=dev)
        t_wtab = timeit(lambda: hip.logistic_tables(e, mu, sc, 31, q, out=wcdf, layout=hip.LAYOUT_WAVE), args.iters)
        wcdf.bs_layout = hip.LAYOUT_WAVE
        tw = []
        for _ in range(args.iters):
            tw.append(timeit(lambda: hip.rans_pop(st, wcdf, K), 1, 0))
            hip.rans_push(st, fo[0], fo[1])
        st.check()
        rows = B * D
        alg = rows * ((K - 1) * 8 + 12)
        res[name] = dict(B=B, D=D, K=K, tables_s=t_tab, tables_wave_s=t_wtab, pop_wave_s=float(np.median(tw)), fc_s=t_fc, pop_s=float(np.median(tp)), push_s=float(np.median(tq)),
                         tables_rows_per_s=rows / t_tab, tables_alg_GBps=alg / t_tab / 1e9,
                         fc_alg_GBps=alg / t_fc / 1e9, sigmoids_per_s_tables=rows * (K - 1) / t_tab)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
