#!/bin/bash
# round 5, visit t: table hand-off at 1000 chains with the spec 3 kernels: cumulative values (default) vs whole rows
TAG=${1:-r05t}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $OUT/${TAG}_${name}.json 2> $OUT/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${TAG}_${name}.json") if l.startswith("{")][-1])
    print("$name", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms lossless", d["lossless"])
except Exception as e:
    print("$name FAILED", e)
PY
}
run pivot_a X=1
run wholerows BITSWAP_PIVOT=0
run pivot_b X=1
