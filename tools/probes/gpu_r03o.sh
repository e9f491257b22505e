#!/bin/bash
TAG=${1:-r03o}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_codec_gpu.py -m gpu -q -x --timeout 300 -k "pivot or full_width_oracle or chain_replay or round_trip" > $OUT/${TAG}_pytest1.log 2>&1; echo "tests exit $?"; tail -4 $OUT/${TAG}_pytest1.log
for g in 2 3 4; do
  timeout 600 python bench.py --no-extra --no-cpu-baseline --groups $g > $OUT/${TAG}_bench_g$g.json 2>> $OUT/${TAG}_bench.err; python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench_g$g.json")); print("groups $g chains 800:", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms", {k: v for k, v in d["stream_time_fraction"].items() if k in ("net","pop_z","tables_z","fc_z")})
PY
done
timeout 600 python bench.py --no-extra --no-cpu-baseline --groups 3 --chains 1200 > $OUT/${TAG}_bench_g3c1200.json 2>> $OUT/${TAG}_bench.err; python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_g3c1200.json')); print('groups 3 chains 1200:', round(d['value']/1e6,3), d['ms_per_step'])"
timeout 600 python bench.py --no-extra --no-cpu-baseline --groups 4 --chains 1600 > $OUT/${TAG}_bench_g4c1600.json 2>> $OUT/${TAG}_bench.err; python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_g4c1600.json')); print('groups 4 chains 1600:', round(d['value']/1e6,3), d['ms_per_step'])"
exit 0
