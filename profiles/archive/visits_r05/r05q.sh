#!/bin/bash
# round 5, visit q: chains per wavefront of the spec 3 table kernel in the pipeline (BITSWAP_TABLE_NB), smoke()
TAG=${1:-r05q}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
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
run nb8_a X=1
run nb4 BITSWAP_TABLE_NB=4
run nb16 BITSWAP_TABLE_NB=16
run nb8_b X=1
run nb12 BITSWAP_TABLE_NB=12
run nb6 BITSWAP_TABLE_NB=6
