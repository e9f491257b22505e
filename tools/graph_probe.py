#!/usr/bin/env python3
"""Does hipGraph capture of the conv stacks work here, is it bitwise equal to eager, what does it save?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import workload  # noqa: E402

dev = torch.device("cuda")
torch.backends.cudnn.deterministic = True
torch.backends.cudnn.benchmark = False
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
model, zend, zcen = workload.build("cifar8", dev, quantbits=10)
model.compress(True)


def timeit(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it, (time.perf_counter() - t0) / it * 1e3


pool = torch.cuda.graph_pool_handle()
side = torch.cuda.Stream()
tot_e = tot_g = 0.0
keep = []
for kind in ("infer", "generate"):
    for i in (0, 1, 7):
        fn = getattr(model, kind)(i)
        D = model.xdim if (kind == "infer" and i == 0) else model.zdim_flat
        x = torch.randn(B, D, device=dev)
        with torch.no_grad():
            mu0, sc0 = fn(x)
            mu0, sc0 = mu0.clone(), sc0.clone()
            sin = x.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=side):
                mu1, sc1 = fn(sin)
            keep.append((g, sin, mu1, sc1))
            g.replay()
            torch.cuda.synchronize()
            same = torch.equal(mu0, mu1) and torch.equal(sc0, sc1.expand_as(mu1))
            te = timeit(lambda: fn(x))
            tg = timeit(lambda: g.replay())
        tot_e += te[0]
        tot_g += tg[0]
        print(f"{kind}({i}) B={B}: eager {te[0]:.3f} ms gpu / {te[1]:.3f} ms wall   graph {tg[0]:.3f} ms gpu / {tg[1]:.3f} ms wall   bitwise equal: {same}", flush=True)
print(f"sum eager {tot_e:.3f} ms, graph {tot_g:.3f} ms")
