#!/usr/bin/env python3
"""Kernel-level timings of the hot path at BASELINE sizes, SURVEY 8(d) parameter regime (latents: mu ~ N(0, 0.5^2),
scale ~ U[0.1, 1], uniform bins over +-2..8; pixels: K = 256).  Run on the GPU box:
    python tools/microbench.py --B 400
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402
from bitswap_amd.bins import uniform_step  # noqa: E402


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


def fresh_state(B, D, dev, nwords=10000):
    np.random.seed(1)
    words = np.random.randint(1 << 16, (1 << 32) - 1, size=(B, nwords), dtype=np.uint32)
    st = hip.RansState(B, nwords + 4 * D, dev)
    st.stack[:, :nwords] = torch.from_numpy(words.view(np.int32)).to(dev)
    st.len.fill_(nwords - 1)
    st.head.copy_(torch.from_numpy((words[:, -1].astype(np.uint64) << np.uint64(32)).view(np.int64)))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=400)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda"
    rng = np.random.RandomState(0)
    res = {}
    for (name, D, K, q) in (("z", 2048, 1024, 10), ("x", 3072, 256, 8)):
        B = args.B
        lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
        e_np = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
        e = torch.from_numpy(e_np).to(dev)
        step = torch.from_numpy(uniform_step(e_np)).to(dev)
        mu = torch.from_numpy((rng.randn(B, D) * 0.5).astype(np.float32)).to(dev)
        sc = torch.from_numpy(rng.uniform(0.1, 1.0, (B, D)).astype(np.float32)).to(dev)
        st = fresh_state(B, D, dev)
        wcdf = torch.empty((B, D, hip.wave_ld(K)), dtype=torch.int32, device=dev)
        fo = (torch.empty((B, D), dtype=torch.int32, device=dev), torch.empty((B, D), dtype=torch.int32, device=dev))
        r = dict(B=B, D=D, K=K)
        for spec, stp in ((1, None), (2, step), (3, step), (4, step)):
            sp = None if stp is None else spec
            t_tab = timeit(lambda: hip.logistic_tables(e, mu, sc, 31, q, out=wcdf, layout=hip.LAYOUT_WAVE, step=stp,
                                                       status=st.status, spec=sp), args.iters)
            wcdf.bs_layout = hip.LAYOUT_WAVE
            sym, _ = hip.rans_pop(st, wcdf, K)
            t_fc = timeit(lambda: hip.logistic_fc(e, mu, sc, sym, st.status, 31, q, out=fo, step=stp, spec=sp), args.iters)
            hip.rans_push(st, fo[0], fo[1])
            tp, tq = [], []
            for _ in range(args.iters):
                tp.append(timeit(lambda: hip.rans_pop(st, wcdf, K), 1, 0))
                tq.append(timeit(lambda: hip.rans_push(st, fo[0], fo[1]), 1, 0))
            st.check()
            rows = B * D
            alg = rows * ((K - 1) * 8 + 12)
            r[f"spec{spec}"] = dict(tables_wave_s=t_tab, fc_s=t_fc, pop_wave_s=float(np.median(tp)),
                                    push_s=float(np.median(tq)), tables_alg_GBps=alg / t_tab / 1e9,
                                    fc_alg_GBps=alg / t_fc / 1e9, ns_per_row_tables=t_tab / rows * 1e9)
            if stp is not None:      # the production hand-off for uniform bins: 64 cumulative values per row
                pcdf = torch.empty((B, D, hip.PIVOT_LD), dtype=torch.int32, device=dev)
                t_ptab = timeit(lambda: hip.logistic_tables(e, mu, sc, 31, q, out=pcdf, layout=hip.LAYOUT_PIVOT, step=stp,
                                                            status=st.status, spec=sp), args.iters)
                tables = hip.logistic_tables(e, mu, sc, 31, q, out=pcdf, layout=hip.LAYOUT_PIVOT, step=stp, status=st.status, spec=sp)
                tpp = []
                for _ in range(args.iters):
                    tpp.append(timeit(lambda: hip.rans_pop(st, tables, K), 1, 0))
                    hip.rans_push(st, fo[0], fo[1])
                st.check()
                r[f"spec{spec}"].update(tables_pivot_s=t_ptab, pop_pivot_s=float(np.median(tpp)),
                                        handoff_bytes_per_row={"wave": 4 * (K + 64), "pivot": 4 * hip.PIVOT_LD})
            # 64-state format: table + rANS step fused, one launch per coding operation
            np.random.seed(1)
            words = np.random.randint(1 << 16, (1 << 32) - 1, size=(B, 64, 160), dtype=np.uint32)
            s64 = hip.RansState64(B, 160 + 4 * (D // 64) + 64, dev)
            s64.stack[:, :, :160] = torch.from_numpy(words.view(np.int32)).to(dev)
            s64.len64.fill_(159)
            s64.head.copy_(torch.from_numpy((words[:, :, -1].astype(np.uint64) << np.uint64(32)).view(np.int64)))
            cen = torch.from_numpy(rng.randn(D, K)).to(dev)
            tp64, tq64 = [], []
            for _ in range(args.iters):
                box = []
                tp64.append(timeit(lambda: box.append(hip.layer_pop64(s64, e, mu, sc, 31, q, centres=cen, step=stp, spec=sp)), 1, 0))
                sy = box[0][0]
                tq64.append(timeit(lambda: hip.layer_push64(s64, e, mu, sc, sy, 31, q, step=stp, spec=sp), 1, 0))
            s64.check()
            r[f"spec{spec}"].update(layer_pop64_s=float(np.median(tp64)), layer_push64_s=float(np.median(tq64)))
        res[name] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
