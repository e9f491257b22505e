#!/bin/bash
# round 5, visit j: bf16x3 forked-step failure -- the checksum trail (where does a failing run first leave the one-stream run?)
TAG=${1:-r05j}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python tools/bf16x3_repro.py --trail 150 > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep "^trail" $OUT/${TAG}_bf16x3_repro.txt | cut -c1-6000
