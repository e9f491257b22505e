#!/bin/bash
# round 4 visit G: first run of the bf16x3 GEMM (error against float64, time against the fp32 MFMA kernel)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
BF16X3_OUT=$OUT/r04g_bf16x3_probe.json timeout 600 python tools/bf16x3_probe.py 2>&1 | grep -v Warning | tail -20
