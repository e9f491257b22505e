#!/bin/bash
# GEMM ablation lab on the GPU box: tools/probes/gemm_lab.hip, wall-clock round-robin + one PMC pass (cycles per variant).
TAG=${1:-r03c}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_lab tools/probes/gemm_lab.hip || exit 1
timeout 300 /tmp/gemm_lab 6400 > $OUT/${TAG}_lab_6400.txt 2>&1; cat $OUT/${TAG}_lab_6400.txt

( cd /tmp && rm -rf pv && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pv -o pv --output-format csv -- /tmp/gemm_lab 6400 5 > /dev/null 2>$OUT/${TAG}_lab_pmc.err )
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:28] + " grid" + r.get("Grid_Size", "?")
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$OUT/${TAG}_lab_pmc.json", "w"), indent=1)
for k, d in sorted(out.items()):
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8; m = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024
    print(f"{k:44s} cycles/XCD {g:10.0f}  mfma busy/SIMD {m:10.0f}  busy frac {m / g if g else 0:.3f}")
PY
exit 0
