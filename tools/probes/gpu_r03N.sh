#!/bin/bash
# visit N: the matrix-core input convolution (k_conv3_mfma) against the VALU one: tests, kernel time alone, bench A/B
TAG=${1:-r03N}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_codec_gpu.py -q -x -k "conv3_wino or full_width_winograd or winograd_convs" 2>&1 | tail -3
python - <<'PY'
import os, torch
from bitswap_amd import hip
N, C = 500, 256
x = torch.randn(N, 8, 16, 16, device="cuda"); w = torch.randn(C, 8, 3, 3, device="cuda") / 8; b = torch.randn(C, device="cuda")
def t(fn, warm=30, reps=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / reps * 1e3
print("k_conv3 (default = mfma): %.1f us per 500-block launch" % t(lambda: hip.conv3_wino(x, w, b, 3, True, 6)))
PY
BITSWAP_CONV3_MFMA=0 python - <<'PY'
import os, torch
from bitswap_amd import hip
N, C = 500, 256
x = torch.randn(N, 8, 16, 16, device="cuda"); w = torch.randn(C, 8, 3, 3, device="cuda") / 8; b = torch.randn(C, device="cuda")
def t(fn, warm=30, reps=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / reps * 1e3
print("k_conv3 (VALU kernel):    %.1f us per 500-block launch" % t(lambda: hip.conv3_wino(x, w, b, 3, True, 6)))
PY
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for rep in 1 2; do for m in 1 0; do
  echo -n "conv3 mfma $m: "; BITSWAP_CONV3_MFMA=$m $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
exit 0
