#!/bin/bash
# round 5, visit I: three kernels -- packed k_wino_fused<6,6> + the 32-chain GEMM + a float64 table kernel on three streams (packed build only)
TAG=${1:-r05I}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
BITSWAP_HIP_LIB=$PWD/bitswap_amd/csrc/libbitswap_hip_slp.so timeout 200 python tools/bf16x3_repro.py --trio > $OUT/${TAG}_trio_packed.txt 2>&1; echo "exit $?"; grep "^trio " $OUT/${TAG}_trio_packed.txt | cut -c1-400; tail -1 $OUT/${TAG}_trio_packed.txt | cut -c1-300
