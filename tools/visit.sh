#!/bin/bash
# round 4 full visit on the final code: GPU suite, default bench line, microbench, kernel stats + PMC FETCH / WRITE passes,
# SQ counters of the table kernels (valu_busy.json keyed by flavour), BASELINE configs through the reference-named scripts
TAG=${1:-visit}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/${TAG}_pytest.log
timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?"; tail -2 $OUT/${TAG}_bench.err
timeout 300 python tools/microbench.py > $OUT/${TAG}_micro.json 2> $OUT/${TAG}_micro.err
timeout 300 python tools/microbench.py --B 13 > $OUT/${TAG}_micro13.json 2>> $OUT/${TAG}_micro.err
BCMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
( cd /tmp && rm -rf prof_stats prof_fetch prof_write
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st --output-format csv -- $BCMD > $OUT/${TAG}_prof_stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o pf --output-format csv -- $BCMD > $OUT/${TAG}_prof_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o pw --output-format csv -- $BCMD > $OUT/${TAG}_prof_write.log 2>&1 )
python tools/prof_summary.py stats /tmp/prof_stats $OUT/${TAG}_kernel_stats.txt > /dev/null
python tools/prof_summary.py pmc /tmp/prof_fetch FETCH_SIZE $OUT/${TAG}_pmc_FETCH_SIZE.json > /dev/null
python tools/prof_summary.py pmc /tmp/prof_write WRITE_SIZE $OUT/${TAG}_pmc_WRITE_SIZE.json > /dev/null
head -16 $OUT/${TAG}_kernel_stats.txt
rm -f $OUT/${TAG}_traffic.json
python tools/prof_summary.py traffic $OUT/${TAG}_pmc_FETCH_SIZE.json $OUT/${TAG}_pmc_WRITE_SIZE.json cifar8 $OUT/${TAG}_traffic.json 1024000
bash tools/pmc_valu.sh ${TAG} > $OUT/${TAG}_pmc_valu.log 2>&1
python tools/valu_busy.py $OUT/${TAG} $OUT/${TAG}_valu_busy.json
bash tools/config_runs.sh > $OUT/${TAG}_configs.txt 2>&1; cat $OUT/${TAG}_configs.txt | grep -i "pixels\|bits/dim\|==" | head -30
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench.json"))
print("headline", round(d["value"]/1e6,3), d["ms_per_step"], d["lossless"])
r = d["roofline"]; print({k: r[k] for k in ("frac","path_frac","hbm_survey_frac","valu_busy_pmc","avg_launch_ms")}, r["mfma"]["frac"])
for e in d["extra"]:
    print((e.get("config") or str(e))[:110], "|", e.get("value"), e.get("ms_per_step"), e.get("lossless"), e.get("conv_dtype"), e.get("error"))
PY
