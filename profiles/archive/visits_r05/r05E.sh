#!/bin/bash
# round 5, full visit on the round's final code (pivot pop diet, net_epilogue.hip without the SLP vectorizer, bf16x3 fork gate removed):
# (1) the whole GPU suite; (2) smoke; (3) regression of the root-caused failure: 400 forked eager bf16x3 runs on the product
# library, record leg; (4) the default bench line; (5) TIMED-REGION-ONLY profile + PMC passes (as tools/r05c.sh / r05z.sh)
TAG=${1:-r05E}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/${TAG}_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
REPRO_RECORD_REPS=400 REPRO_MAX_FAIL=99 timeout 400 python tools/bf16x3_repro.py --record > $OUT/${TAG}_bf16x3_record_product.txt 2>&1
python - $OUT/${TAG}_bf16x3_record_product.txt <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
line = [l for l in txt.splitlines() if l.startswith("record ")]
d = json.loads(line[-1][7:]) if line else {"error": txt[-500:]}
print("bf16x3 forked, product library:", {k: d.get(k) for k in ("runs", "lossless", "baseline_replay_mismatches", "error")})
PY
timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?"; tail -2 $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/${TAG}_bench.json") if l.startswith("{")][-1])
print(json.dumps(d["summary"])[:2600])
r = d["roofline"]; print({k: r.get(k) for k in ("frac","valu_issue_frac","avg_launch_ms","avg_launch_ms_in_pipeline","traffic","valu_busy_pmc")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")})
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
head -14 $OUT/${TAG}_kernel_stats_timed.txt | cut -c1-160
rm -f $OUT/${TAG}_traffic.json
python tools/prof_summary.py traffic $OUT/${TAG}_pmc_FETCH_SIZE_timed.json $OUT/${TAG}_pmc_WRITE_SIZE_timed.json cifar8 $OUT/${TAG}_traffic.json 1024000 | cut -c1-300
