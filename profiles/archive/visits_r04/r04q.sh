#!/bin/bash
# round 4 visit Q: 100 chains in two chain groups under the kernel trace (why do two groups not beat one?)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
tr() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( cd /tmp && rm -rf tr_$tag && env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 6 --warmup 2 "$@" > $OUT/r04q_$tag.log 2>&1 )
  grep -h '^{' $OUT/r04q_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['lossless'])"
  python tools/step_trace.py /tmp/tr_$tag $OUT/r04q_trace_$tag.txt --ms $MS | tail -24
}
MS=25 tr c100_g2 BITSWAP_FORK=auto -- --chains 100 --groups 2
MS=25 tr c100_g2_nofork BITSWAP_FORK=0 -- --chains 100 --groups 2
