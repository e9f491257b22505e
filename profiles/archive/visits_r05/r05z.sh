#!/bin/bash
# round 5, final visit on the round's code: (1) the whole GPU suite; (2) the default fp32 route's forked step, 4 x 250 runs (graph
# replay and eager) -- the control series of the bf16x3 hunt at full length; (3) the default bench line; (4) TIMED-REGION-ONLY profile
# + PMC passes + SQ counters (as tools/r05c.sh)
TAG=${1:-r05z}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/${TAG}_pytest.log
REPRO_FOCUS_REPS=250 timeout 1200 python tools/bf16x3_repro.py --focus > $OUT/${TAG}_fp32_fork_control.txt 2>&1; grep "^codec" $OUT/${TAG}_fp32_fork_control.txt | cut -c1-300
timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?"; tail -2 $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/${TAG}_bench.json") if l.startswith("{")][-1])
print(json.dumps(d["summary"])[:2500])
r = d["roofline"]; print({k: r.get(k) for k in ("frac","valu_issue_frac","avg_launch_ms","avg_launch_ms_in_pipeline","traffic","valu_busy_pmc")})
PY
BCMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
export BITSWAP_BENCH_SENTINEL=1
( cd /tmp && rm -rf prof_stats prof_fetch prof_write
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st --output-format csv -- $BCMD > $OUT/${TAG}_prof_stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o pf --output-format csv -- $BCMD > $OUT/${TAG}_prof_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o pw --output-format csv -- $BCMD > $OUT/${TAG}_prof_write.log 2>&1 )
unset BITSWAP_BENCH_SENTINEL
python tools/prof_summary.py stats /tmp/prof_stats $OUT/${TAG}_kernel_stats_timed.txt timed > /dev/null
python tools/prof_summary.py pmc /tmp/prof_fetch FETCH_SIZE $OUT/${TAG}_pmc_FETCH_SIZE_timed.json timed > /dev/null
python tools/prof_summary.py pmc /tmp/prof_write WRITE_SIZE $OUT/${TAG}_pmc_WRITE_SIZE_timed.json timed > /dev/null
head -12 $OUT/${TAG}_kernel_stats_timed.txt
rm -f $OUT/${TAG}_traffic.json
python tools/prof_summary.py traffic $OUT/${TAG}_pmc_FETCH_SIZE_timed.json $OUT/${TAG}_pmc_WRITE_SIZE_timed.json cifar8 $OUT/${TAG}_traffic.json 1024000 | cut -c1-300
