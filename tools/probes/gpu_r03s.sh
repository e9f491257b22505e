#!/bin/bash
TAG=${1:-r03s}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline $EXTRA > $OUT/${TAG}_$n.json 2>> $OUT/${TAG}.err; python - <<PY
import json
d=json.load(open("$OUT/${TAG}_$n.json")); print("$n:", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms", d["lossless"])
PY
}
for c in 800 1000 1200; do for g in 2 3; do EXTRA="--chains $c --groups $g" run gs_c${c}_g$g BITSWAP_GROUP_STREAMS=1; done; done
EXTRA="--chains 1600 --groups 4" run gs_c1600_g4 BITSWAP_GROUP_STREAMS=1
EXTRA="--chains 1000 --groups 2 --workload imagenet4" run gs_im4_c1000 BITSWAP_GROUP_STREAMS=1
EXTRA="--chains 800 --groups 2 --workload imagenet4" run gs_im4_c800 BITSWAP_GROUP_STREAMS=1
exit 0
