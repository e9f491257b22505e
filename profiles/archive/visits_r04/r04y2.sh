#!/bin/bash
# round 4 visit Y: the transform passes and the input conv with whole 16-byte LDS reads (tests, times alone, counters, step)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "wino or conv or fused or stack or model or lossless" 2>&1 | tail -2
cat > /tmp/s_loop.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
from bitswap_amd import hip
N = 500
g = torch.Generator().manual_seed(0)
x8 = torch.randn((N, 8, 16, 16), generator=g).cuda(); w = (torch.randn((256, 8, 3, 3), generator=g) / 8).cuda(); b = torch.randn(256, generator=g).cuda()
M = torch.randn(36, 256, N * 16, device="cuda"); M8 = torch.randn(64, 256, N * 16, device="cuda"); x = torch.randn(N, 256, 16, 16, device="cuda")
for _ in range(20):
    hip.conv3_wino(x8, w, b, 3, True, 6)
    hip.wino_fused(M, (N, 256, 16, 16), 6, b, x, True, ts_out=6)
    hip.wino_fused(M8, (N, 256, 16, 16), 8, b, x, True, ts_out=8)
torch.cuda.synchronize()
PY
: > $OUT/r04y2_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  ( cd /tmp && rm -rf pvs && R=$R timeout 300 rocprofv3 --pmc $set -d /tmp/pvs -o pv --output-format csv -- python /tmp/s_loop.py > /dev/null 2>$OUT/r04y2_pmc.err )
  python - <<PY | tee -a $OUT/r04y2_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pvs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:28]
        if "k_conv3_wino" in k or "k_wino_fused" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    print(k, {c: round(v[1] / v[0]) for c, v in d.items()})
PY
done
python - <<'PY' 2>&1 | grep -v Warn | tee -a $OUT/r04y2_counters.txt
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
print("alone, us: conv3_wino", t(lambda: hip.conv3_wino(x8, w, b, 3, True, 6)), "fused<6,6>", t(lambda: hip.wino_fused(M, (N, 256, 16, 16), 6, b, x, True, ts_out=6)),
      "fused<8,8>", t(lambda: hip.wino_fused(M8, (N, 256, 16, 16), 8, b, x, True, ts_out=8)))
PY
for i in 1 2; do python bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 8 --warmup 2 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chains 1000', d['ms_per_step'], d['value'], d['lossless'])"; done
for c in 13 100; do python bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 12 --warmup 3 --chains $c --groups 1 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chains $c', d['ms_per_step'], d['lossless'])"; done
