#!/bin/bash
# round 4 visit U: the same overlap statistics with the bf16x3 GEMM route (why does a 1.42x kernel buy the step 1.5 %?)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for arith in bf16x3 fp32; do
( cd /tmp && rm -rf tr_u && BITSWAP_GEMM_ARITH=$arith timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_u -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 4 --warmup 1 > $OUT/r04u_$arith.log 2>&1 )
grep -h '^{' $OUT/r04u_$arith.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arith ms_per_step', d['ms_per_step'], d['lossless'])"
python tools/overlap_stats.py /tmp/tr_u --ms 700 | tee $OUT/r04u_overlap_$arith.txt | head -12
done
