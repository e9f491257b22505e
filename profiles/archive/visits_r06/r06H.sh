#!/bin/bash
# round 6, visit H: where does the 100-chain step (22.1 ms, the reference's own shape) go?  Kernel trace of the timed region: bulk /
# serial / idle shares of the wall, per-kernel sums.
TAG=${1:-r06H}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
BCMD="python $R/bench.py --scaling strong --total-chains 100 --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-roofline --full-record /dev/null"
export BITSWAP_BENCH_SENTINEL=1
( cd /tmp && rm -rf prof100 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof100 -o st --output-format csv -- $BCMD > $OUT/${TAG}_prof.log 2>&1 )
unset BITSWAP_BENCH_SENTINEL
python tools/prof_summary.py stats /tmp/prof100 $OUT/${TAG}_kernel_stats_timed_100chains.txt timed > /dev/null
head -30 $OUT/${TAG}_kernel_stats_timed_100chains.txt | cut -c1-170
python tools/overlap_stats.py /tmp/prof100 --ms 120 > $OUT/${TAG}_overlap_100chains.txt 2>&1; cat $OUT/${TAG}_overlap_100chains.txt | head -60
