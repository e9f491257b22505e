#!/bin/bash
# round 6, visit X2: how the wall of the 1000-chain step splits -- a bulk kernel running / only serial coder kernels / nothing --
# on the final code (tools/overlap_stats.py over a kernel trace of the timed region)
TAG=${1:-r06X2}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
BCMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-roofline --full-record /dev/null"
export BITSWAP_BENCH_SENTINEL=1
( cd /tmp && rm -rf prof1k && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1k -o st --output-format csv -- $BCMD > $OUT/${TAG}_prof.log 2>&1 )
unset BITSWAP_BENCH_SENTINEL
python tools/overlap_stats.py /tmp/prof1k --ms 400 > $OUT/${TAG}_overlap_1000chains.txt 2>&1; head -30 $OUT/${TAG}_overlap_1000chains.txt
