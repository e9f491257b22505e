"""bs_logistic_tables alone (pivot hand-off, CDF spec 4) at the headline launch size: 500 chains x 2048 dims x 1024 bins of uniform width.
Used to compare builds of the table kernel (BITSWAP_HIPCC_EXTRA=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
B, D, K = 500, 2048, 1024
lo = -6.0 + torch.rand(D, generator=g, dtype=torch.float64)
hi = 6.0 + torch.rand(D, generator=g, dtype=torch.float64)
h = (hi - lo) / K
endpoints = (lo[:, None] + torch.arange(1, K, dtype=torch.float64)[None, :] * h[:, None]).to(dev)
step = h.to(dev)
mu = (torch.randn(B, D, generator=g) * 1.5).to(dev)
scale = (0.1 + 0.9 * torch.rand(B, D, generator=g)).to(dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
for layout, name in ((hip.LAYOUT_PIVOT, "pivot"),):
    out = hip.logistic_tables(endpoints, mu, scale, layout=layout, step=step, status=status, spec=4)
    for _ in range(5):
        hip.logistic_tables(endpoints, mu, scale, layout=layout, step=step, status=status, spec=4, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        hip.logistic_tables(endpoints, mu, scale, layout=layout, step=step, status=status, spec=4, out=out)
    b.record()
    torch.cuda.synchronize()
    print(f"k_logistic<{name}, spec 4>, {B} x {D} rows: {a.elapsed_time(b) / 30 * 1e3:.1f} us per launch   lib={os.path.basename(hip.load()._name)}  bad={int((status != 0).sum())}")
