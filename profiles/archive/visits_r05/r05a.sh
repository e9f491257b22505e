#!/bin/bash
# round 5, visit a: GPU suite on the CDF spec 3 code, kernel microbench (specs 1 / 2 / 3 side by side), headline bench with spec 3
# and spec 2 on the same box
TAG=${1:-r05a}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/${TAG}_pytest.log
timeout 300 python tools/microbench.py --B 500 > $OUT/${TAG}_micro.json 2> $OUT/${TAG}_micro.err; tail -2 $OUT/${TAG}_micro.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_micro.json"))
for n in ("z","x"):
    for sp in ("spec1","spec2","spec3"):
        r=d[n][sp]; print(n,sp,{k:round(v*1e6,1) for k,v in r.items() if k.endswith("_s")})
PY
for sp in 3 2 3 2; do
  timeout 600 python bench.py --no-extra --no-cpu-baseline --cdf-spec $sp > $OUT/${TAG}_bench_spec${sp}.json 2> $OUT/${TAG}_bench_spec${sp}.err
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/${TAG}_bench_spec${sp}.json") if l.startswith("{")][-1])
r=d["roofline"]
print("spec $sp", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms lossless", d["lossless"], "tables excl ms", r["avg_launch_ms"], "in pipeline", r["avg_launch_ms_in_pipeline"], "valu_issue_frac", r.get("valu_issue_frac"), "path_frac", r["frac"])
PY
done
