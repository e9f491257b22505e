#!/bin/bash
TAG=${1:-r03e}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "wino_gemm or own_gemm" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/${TAG}_pytest.log
timeout 600 python tools/gemm_probe.py > $OUT/${TAG}_gemm_probe.txt 2>&1; echo "probe exit $?"; grep -v Warning $OUT/${TAG}_gemm_probe.txt
( cd /tmp && rm -rf pv && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pv -o pv --output-format csv -- python $R/tools/gemm_probe.py --quick --only-own > /dev/null 2>$OUT/${TAG}_pmc.err )
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40] + " grid" + r.get("Grid_Size", "?")
        if "wino_gemm" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    d = {c: v[1] / v[0] for c, v in d.items()}
    g = d.get("GRBM_GUI_ACTIVE", 0) / 8; m = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024
    print(f"{k:60s} cycles/XCD {g:10.0f}  mfma busy/SIMD {m:10.0f}  busy frac {m / g if g else 0:.3f}")
PY
exit 0
