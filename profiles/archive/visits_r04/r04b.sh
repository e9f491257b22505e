#!/bin/bash
# round 4 visit B: kernel traces of the forked few-chain step (13 and 100 chains), fork on / off
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
tr() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( cd /tmp && rm -rf tr_$tag && env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 6 --warmup 2 "$@" > $OUT/r04b_$tag.log 2>&1 )
  grep -h '^{' $OUT/r04b_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['lossless'])"
  python tools/step_trace.py /tmp/tr_$tag $OUT/r04b_trace_$tag.txt --ms $MS | tail -32
}
MS=14 tr c13_fork1 BITSWAP_FORK=auto -- --chains 13 --groups 1
MS=14 tr c13_fork0 BITSWAP_FORK=0 -- --chains 13 --groups 1
MS=28 tr c100_fork1 BITSWAP_FORK=auto -- --chains 100 --groups 1
MS=28 tr c100_fork0 BITSWAP_FORK=0 -- --chains 100 --groups 1
