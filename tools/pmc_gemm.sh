#!/bin/bash
# SQ counters of the conv GEMMs alone (how busy is the matrix pipe, what do the wavefronts wait for, at what clock): separate --pmc
# passes over 20 launches each of bs_wino_gemm_bf16x3 (the shapes named in SHAPES: "shape:persistent") and bs_wino_gemm_f32.
#   bash tools/pmc_gemm.sh <tag>
TAG=${1:-pmc_gemm}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cat > /tmp/x3loop.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
from bitswap_amd import hip
dev = "cuda"
T, C, cols = 36, 256, 8000
U = torch.randn(T, C, C, device=dev); V = torch.randn(T, C, cols, device=dev); M = torch.empty(T, C, cols, device=dev)
Uf = hip.frags_bf16x3(U)
for shape, pers in (("3", "0"), ("3", "1"), ("2", "0")):
    os.environ["BITSWAP_BF16X3_SHAPE"], os.environ["BITSWAP_BF16X3_PERSISTENT"] = shape, pers
    for _ in range(20): hip.wino_gemm_bf16x3(Uf, V, 6, out=M)
for _ in range(20): hip.wino_gemm(U, V, out=M)
torch.cuda.synchronize()
PY
: > $OUT/${TAG}_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  ( cd /tmp && rm -rf pvx && R=$R timeout 300 rocprofv3 --pmc $set -d /tmp/pvx -o pv --output-format csv -- python /tmp/x3loop.py > /dev/null 2>$OUT/${TAG}_pmc.err )
  python - >> $OUT/${TAG}_sq.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pvx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]
        if "k_wino_gemm" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    print(k, {c: round(v[1] / v[0]) for c, v in d.items()})
PY
done
cat $OUT/${TAG}_sq.txt
