#!/bin/bash
# round 4 visit P: fresh kernel traces of the forked step (13 and 100 chains) with the final kernels, and the SQ counters of the
# bf16x3 GEMM written to a file this time
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
tr() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( cd /tmp && rm -rf tr_$tag && env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 6 --warmup 2 "$@" > $OUT/r04p_$tag.log 2>&1 )
  grep -h '^{' $OUT/r04p_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['lossless'])"
  python tools/step_trace.py /tmp/tr_$tag $OUT/r04p_trace_$tag.txt --ms $MS | tail -24
}
MS=11 tr c13 BITSWAP_FORK=auto -- --chains 13 --groups 1
MS=24 tr c100 BITSWAP_FORK=auto -- --chains 100 --groups 1
bash tools/visits/r04m.sh 2>&1 | tee $OUT/r04p_bf16x3_sq.txt | tail -8
