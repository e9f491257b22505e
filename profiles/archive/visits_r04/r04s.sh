#!/bin/bash
# round 4 visit S: what do k_conv3_wino and k_wino_fused spend their time on?  SQ counters of both kernels alone at 500 blocks
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
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
: > $OUT/r04s_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  ( cd /tmp && rm -rf pvs && R=$R timeout 300 rocprofv3 --pmc $set -d /tmp/pvs -o pv --output-format csv -- python /tmp/s_loop.py > /dev/null 2>$OUT/r04s_pmc.err )
  python - <<PY | tee -a $OUT/r04s_counters.txt
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
python - <<'PY' 2>&1 | grep -v Warn | tee -a $OUT/r04s_counters.txt
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
