#!/bin/bash
# round 5, visit g: bf16x3 -- every GEMM launch of the eager forked codec run TWICE and compared on the device: is the GEMM the
# one that differs, and where (columns / rows / transform positions)?
TAG=${1:-r05g}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPRO_FOCUS_REPS=200 timeout 2400 python tools/bf16x3_repro.py --focus > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep "^codec" $OUT/${TAG}_bf16x3_repro.txt | cut -c1-3000
