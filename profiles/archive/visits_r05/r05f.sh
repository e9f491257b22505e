#!/bin/bash
# round 5, visit f: bf16x3 -- the victim hunt (30,000 GEMM launches per variant beside the small-register kernels, everything
# compared on the device) and the eager forked codec 150 runs per variant: unclaimed / unclaimed with the in-place fragment loads
# issued after the barrier / claimed
TAG=${1:-r05f}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python tools/bf16x3_repro.py --storm 300000 --focus > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep "^codec\|^storm" $OUT/${TAG}_bf16x3_repro.txt | cut -c1-600
