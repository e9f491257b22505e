#!/bin/bash
# Round-3 visit A: parity tests, GEMM probe (new vs round-2 kernel vs library) + MFMA counters, default bench line.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r03a.sh r03a'
TAG=${1:-r03a}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/${TAG}_pytest.log
tail -25 $OUT/${TAG}_pytest.log
timeout 600 python tools/gemm_probe.py > $OUT/${TAG}_gemm_probe.txt 2>&1; echo "probe exit $?"; cat $OUT/${TAG}_gemm_probe.txt | grep -v Warning
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && rm -rf pv && timeout 300 rocprofv3 --pmc $set -d /tmp/pv -o pv --output-format csv -- python $R/tools/gemm_probe.py --quick --only-own > /dev/null 2>$OUT/${TAG}_pmc_$n.err )
  python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        if "wino_gemm" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$OUT/${TAG}_gemm_pmc_$n.json", "w"), indent=1)
for k, d in out.items():
    print(k[:60], {c: round(v) for c, v in d.items()})
PY
done
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
exit 0
