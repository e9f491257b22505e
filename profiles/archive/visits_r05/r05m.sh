#!/bin/bash
# round 5, visit m: bf16x3 forked-step failure, bisection by swapping kernels (no launch added): without k_conv3_wino
TAG=${1:-r05m}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPRO_BISECT=1 REPRO_FOCUS_REPS=200 timeout 2400 python tools/bf16x3_repro.py --focus > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep "^codec" $OUT/${TAG}_bf16x3_repro.txt | cut -c1-500
