#!/bin/bash
TAG=${1:-r03r}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline $EXTRA > $OUT/${TAG}_$n.json 2>> $OUT/${TAG}.err; python - <<PY
import json
d=json.load(open("$OUT/${TAG}_$n.json")); print("$n:", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms")
PY
}
EXTRA="" run default A=1
EXTRA="--tables-on serial" run tables_serial A=1
EXTRA="" run group_streams BITSWAP_GROUP_STREAMS=1
EXTRA="" run one_bulk BITSWAP_BULK_PER_GROUP=0
EXTRA="--chains 1000" run chains1000 A=1
EXTRA="--chains 1200" run chains1200 A=1
exit 0
