#!/bin/bash
# One full visit of a GPU box on the current code (the one script; earlier rounds' one-offs: profiles/archive/visits_r0*):
#   bash tools/visit.sh <tag> [suite] [bench] [profile] [valu] [configs]      (no selection = all of them)
# suite:   the whole GPU test suite + smoke
# bench:   the driver's own command line (stdout = the compact headline, full record beside it), EXTRA=all for the twenty shapes
# profile: TIMED-REGION-ONLY rocprofv3 passes of the bench (k_where sentinels on either side of its timed region): kernel trace
#          + stats, PMC FETCH_SIZE and WRITE_SIZE in passes of their own -> kernel stats, per-kernel traffic (traffic.json)
# valu:    SQ counters of the table kernels (valu_busy.json, keyed by kernel flavour)
# configs: BASELINE configs through the reference-named scripts (tools/config_runs.sh)
TAG=${1:-visit}; shift
WHAT=${*:-suite bench profile valu configs}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has suite; then
  timeout 1700 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
  echo "pytest exit $?"; tail -3 $OUT/${TAG}_pytest.log; tail -5 $OUT/${TAG}_pytest.log > $OUT/${TAG}_pytest_tail.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
if has bench; then
  T0=$(date +%s)
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --extra ${EXTRA:-core} --full-record $OUT/${TAG}_bench_full.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  echo "bench exit $? in $(( $(date +%s) - T0 )) s; last stdout line: $(tail -1 $OUT/${TAG}_bench.json | wc -c) bytes"; tail -2 $OUT/${TAG}_bench.err
  tail -1 $OUT/${TAG}_bench.json
fi
if has profile; then
  BCMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --full-record /dev/null"
  export BITSWAP_BENCH_SENTINEL=1
  ( cd /tmp && rm -rf prof_stats prof_fetch prof_write
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st --output-format csv -- $BCMD > $OUT/${TAG}_prof_stats.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o pf --output-format csv -- $BCMD > $OUT/${TAG}_prof_fetch.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o pw --output-format csv -- $BCMD > $OUT/${TAG}_prof_write.log 2>&1 )
  unset BITSWAP_BENCH_SENTINEL
  python tools/prof_summary.py stats /tmp/prof_stats $OUT/${TAG}_kernel_stats_timed.txt timed > /dev/null
  python tools/prof_summary.py pmc /tmp/prof_fetch FETCH_SIZE $OUT/${TAG}_pmc_FETCH_SIZE_timed.json timed > /dev/null
  python tools/prof_summary.py pmc /tmp/prof_write WRITE_SIZE $OUT/${TAG}_pmc_WRITE_SIZE_timed.json timed > /dev/null
  head -16 $OUT/${TAG}_kernel_stats_timed.txt | cut -c1-170
  rm -f $OUT/${TAG}_traffic.json
  python tools/prof_summary.py traffic $OUT/${TAG}_pmc_FETCH_SIZE_timed.json $OUT/${TAG}_pmc_WRITE_SIZE_timed.json cifar8 $OUT/${TAG}_traffic.json 1024000 | cut -c1-300
fi
if has valu; then
  bash tools/pmc_valu.sh ${TAG} > $OUT/${TAG}_pmc_valu.log 2>&1
  python tools/valu_busy.py $OUT/${TAG} $OUT/${TAG}_valu_busy.json; head -40 $OUT/${TAG}_valu_busy.json
fi
if has configs; then
  bash tools/config_runs.sh > $OUT/${TAG}_configs.txt 2>&1; grep -i "pixels\|bits/dim\|==" $OUT/${TAG}_configs.txt | head -30
fi
