#!/bin/bash
# round 4 visit F: the whole GPU suite + the default bench line (with extras) on the code of the visit
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r04f_pytest.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/r04f_pytest.log
timeout 1200 python bench.py > $OUT/r04f_bench.json 2> $OUT/r04f_bench.err
echo "bench exit $?"; tail -3 $OUT/r04f_bench.err
python - <<PY
import json
d = json.load(open("$OUT/r04f_bench.json"))
print("headline", round(d["value"]/1e6,3), d["ms_per_step"], d["lossless"], d["config"]["chain_groups"])
r = d["roofline"]; print({k: r[k] for k in ("frac","path_frac","hbm_survey_frac","valu_busy_pmc","avg_launch_ms")}, r["mfma"]["frac"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("host"))
for e in d["extra"]:
    print(e.get("config") or e, "|", e.get("value"), e.get("ms_per_step"), e.get("lossless"), e.get("forked_block_step"), e.get("error"))
PY
