#!/bin/bash
# VALU / wait-state counters of the table kernels (separate --pmc pass, no tracing domains combined).
#   gpurun --timeout 900 -- 'bash tools/pmc_valu.sh r02p'
TAG=${1:-rXX}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && rm -rf pv && timeout 300 rocprofv3 --pmc $set -d /tmp/pv -o pv --output-format csv -- python $R/tools/microbench.py --B 500 --iters 3 > /dev/null 2>$OUT/${TAG}_pmc_$n.err )
  python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
        if "k_logistic" in k or "k_layer64" in k or "k_rans" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$OUT/${TAG}_pmc_$n.json", "w"), indent=1)
for k, d in out.items():
    print(k[:60], {c: round(v) for c, v in d.items()})
PY
done
