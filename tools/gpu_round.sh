#!/bin/bash
# One GPU-box visit: parity tests, bench, kernel microbench, rocprofv3 stats + PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a [notest] [noprof] [nobench]'
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
TAG=${1:-rXX}
shift
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { for a in "$@"; do [ "$a" = "$WANT" ] && return 0; done; return 1; }
WANT=notest; if ! has "$@"; then
  timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_pytest.log" 2>&1
  echo "pytest exit $?" | tee -a "$OUT/${TAG}_pytest.log"
  tail -15 "$OUT/${TAG}_pytest.log"
fi
WANT=nobench; if ! has "$@"; then
  timeout 900 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
  echo "bench exit $?"; cat "$OUT/${TAG}_bench.json"; tail -3 "$OUT/${TAG}_bench.err"
fi
timeout 300 python tools/microbench.py > "$OUT/${TAG}_micro.json" 2> "$OUT/${TAG}_micro.err"
echo "micro exit $?"; cat "$OUT/${TAG}_micro.json"; tail -3 "$OUT/${TAG}_micro.err"
WANT=noprof; if ! has "$@"; then
  BCMD="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
  ( cd /tmp && rm -rf prof_stats prof_fetch prof_write
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st --output-format csv -- $BCMD > "$OUT/${TAG}_prof_stats.log" 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o pf --output-format csv -- $BCMD > "$OUT/${TAG}_prof_fetch.log" 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o pw --output-format csv -- $BCMD > "$OUT/${TAG}_prof_write.log" 2>&1 )
  python tools/prof_summary.py stats /tmp/prof_stats "$OUT/${TAG}_kernel_stats.txt" > /dev/null
  python tools/prof_summary.py pmc /tmp/prof_fetch FETCH_SIZE "$OUT/${TAG}_pmc_FETCH_SIZE.json" > /dev/null
  python tools/prof_summary.py pmc /tmp/prof_write WRITE_SIZE "$OUT/${TAG}_pmc_WRITE_SIZE.json" > /dev/null
  head -25 "$OUT/${TAG}_kernel_stats.txt"
  grep -h "BENCH\|metric" "$OUT/${TAG}_prof_stats.log" | tail -2
fi
exit 0
