#!/bin/bash
# round 5, visit e: is the eager forked codec itself flaky?  60 runs each: default fp32 GEMM (control), bf16x3 unclaimed / claimed
TAG=${1:-r05e}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python tools/bf16x3_repro.py --focus > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep "^codec" $OUT/${TAG}_bf16x3_repro.txt | cut -c1-600
