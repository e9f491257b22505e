#!/bin/bash
# round 4 visit H: bf16x3 GEMM with pieces switched off (timing only): where does a K step's time go
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export BITSWAP_HIP_LIB=/tmp/libbitswap_lab.so BITSWAP_HIPCC_EXTRA=-DBS_GEMM_LAB
python -m bitswap_amd.build > /dev/null 2>&1
for lab in "" 4 1; do
  BITSWAP_BF16X3_LAB=$lab python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bitswap_amd import hip
if not os.environ.get("BITSWAP_BF16X3_LAB"): os.environ.pop("BITSWAP_BF16X3_LAB", None)
dev = "cuda"
for (T, C, cols) in ((36, 256, 8000), (36, 256, 1600)):
    U = torch.randn(T, C, C, device=dev); V = torch.randn(T, C, cols, device=dev); M = torch.empty(T, C, cols, device=dev)
    Uf = hip.frags_bf16x3(U)
    f = lambda: hip.wino_gemm_bf16x3(Uf, V, 6, out=M)
    for _ in range(50): f()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(100): f()
    b.record(); torch.cuda.synchronize()
    print("lab", os.environ.get("BITSWAP_BF16X3_LAB", "product"), cols, "cols:", round(a.elapsed_time(b) / 100 * 1e3, 1), "us")
PY
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
