#!/bin/bash
# visit E: who occupies the wall time of a step (kernel-trace timeline of the timed region)
TAG=${1:-r03E}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
BCMD="python $PWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-roofline"
( cd /tmp && rm -rf prof_tl && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl --output-format csv -- $BCMD > $OUT/${TAG}_tl.log 2>&1 )
grep -h "metric" $OUT/${TAG}_tl.log | cut -c1-200
python tools/prof_summary.py timeline /tmp/prof_tl $OUT/${TAG}_timeline.json 0 0
python - <<'PY'
import json,sys
d=json.load(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03E_timeline.json"))
for n,v in d["kernel_ms_sum"].items(): print(f'{v["ms"]:9.1f} ms {v["calls"]:5d}  {n[:100]}')
PY
exit 0
