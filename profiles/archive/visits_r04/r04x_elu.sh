#!/bin/bash
# round 4 visit X2: the fixed-sequence packed ELU against the device library's expm1f (old library built beside the new one)
export TMPDIR=/tmp; OLD=$PWD/bitswap_amd/csrc/libbitswap_hip_old.so
timeout 900 python -m pytest tests -m gpu -x -q -k "wino or conv or fused or stack or model or lossless or bits_per_dim or elu" 2>&1 | tail -2
cat > /tmp/t_alone.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bitswap_amd import hip
N = 500
g = torch.Generator().manual_seed(0)
x8 = torch.randn((N, 8, 16, 16), generator=g).cuda(); w = (torch.randn((256, 8, 3, 3), generator=g) / 8).cuda(); b = torch.randn(256, generator=g).cuda()
M = torch.randn(36, 256, N * 16, device="cuda"); M8 = torch.randn(64, 256, N * 16, device="cuda"); x = torch.randn(N, 256, 16, 16, device="cuda")
def t(fn, n=100):
    for _ in range(20): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return round(a.elapsed_time(e) / n * 1e3, 1)
print(os.environ.get("BITSWAP_HIP_LIB", "new")[-10:], "alone, us: conv3_wino(act 3)", t(lambda: hip.conv3_wino(x8, w, b, 3, True, 6)), "fused<6,6> act 1 / 3", t(lambda: hip.wino_fused(M, (N, 256, 16, 16), 6, b, x, True, ts_out=6)),
      t(lambda: hip.wino_fused(M, (N, 256, 16, 16), 6, b, x, 3, want_act=True, ts_out=6)), "fused<8,8>", t(lambda: hip.wino_fused(M8, (N, 256, 16, 16), 8, b, x, True, ts_out=8)))
# accuracy of the new ELU against float64 expm1 on a dense grid (through k_bias_res_elu)
xs = torch.linspace(-30, 5, 2_000_000, device="cuda").view(1, 1, -1, 4).contiguous()
from bitswap_amd import hip as h
a = h.bias_res_elu(xs, None, None, want_act=True)[1] if hasattr(h, "bias_res_elu") else None
if a is not None:
    ref = torch.where(xs.double() > 0, xs.double(), torch.expm1(xs.double()))
    err = (a.double() - ref).abs(); ulp = torch.maximum(ref.abs(), torch.tensor(1e-30, device="cuda", dtype=torch.float64)) * 2.0 ** -23
    print("ELU max abs err %.3e, max err in ulp of the result %.2f" % (float(err.max()), float((err / ulp).max())))
PY
python /tmp/t_alone.py 2>&1 | grep -v Warn; BITSWAP_HIP_LIB=$OLD python /tmp/t_alone.py 2>&1 | grep -v Warn
run() { BITSWAP_HIP_LIB=$1 python bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 8 --warmup 2 $3 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 $3', d['ms_per_step'], d['value'], d['lossless'], round(d['bits_per_dim'],4))"; }
run "" new; run $OLD old; run "" new; run $OLD old
run "" new "--chains 13 --groups 1"; run $OLD old "--chains 13 --groups 1"; run "" new "--chains 100 --groups 1"; run $OLD old "--chains 100 --groups 1"
