#!/bin/bash
# round 4 visit T: is the 1000-chain step bound by the amount of bulk work or by the serial chain?  (kernel trace, overlap statistics)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
( cd /tmp && rm -rf tr_t && timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_t -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 4 --warmup 1 > $OUT/r04t.log 2>&1 )
grep -h '^{' $OUT/r04t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['lossless'])"
python tools/overlap_stats.py /tmp/tr_t --ms 700 | tee $OUT/r04t_overlap.txt
