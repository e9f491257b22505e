#!/bin/bash
# Does the power-management level change what the GEMM and the bench see?  (diagnostic; the bench never touches it)
export TMPDIR=/tmp
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$\|====" | head -20
python - <<'PY'
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from bitswap_amd import hip
U = torch.randn(36, 256, 256, device="cuda"); V = torch.randn(36, 256, 6400, device="cuda"); M = torch.empty(36, 256, 6400, device="cuda")
x = torch.randn(400, 256, 16, 16, device="cuda"); b = torch.randn(256, device="cuda")
def t(fn, n):
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / n * 1e3
g = lambda: hip.wino_gemm(U, V, out=M)
f = lambda: hip.wino_fused(M, (400, 256, 16, 16), 6, b, x, True, ts_out=6)
for rep in range(3):
    time.sleep(0.3)
    print("cold 5 launches: gemm %.1f us" % t(g, 5), "| after 300: %.1f us" % (t(g, 300) and t(g, 100)), "| alternating gemm/fused (in-situ like): gemm+fused %.1f us" % t(lambda: (g(), f()), 200), flush=True)
PY
