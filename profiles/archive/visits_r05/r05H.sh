#!/bin/bash
# round 5, last sanity visit on the final library: GPU suite, smoke, the headline step without the extras
TAG=${1:-r05H}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -2 $OUT/${TAG}_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-extra --steps 6 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/${TAG}_bench.json") if l.startswith("{")][-1])
print(round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms lossless", d["lossless"], {k: d["roofline"].get(k) for k in ("frac","valu_issue_frac","avg_launch_ms")}, d["cpu_baseline"]["value"])
PY
