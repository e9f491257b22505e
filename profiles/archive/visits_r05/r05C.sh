#!/bin/bash
# round 5, visit C: two kernels only -- k_wino_fused<6,6> on a side stream beside the GEMM of the 32-chain codec -- with the packed
# build and with the product build of net_epilogue.hip
TAG=${1:-r05C}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SLP=$PWD/bitswap_amd/csrc/libbitswap_hip_slp.so
BITSWAP_HIP_LIB=$SLP timeout 400 python tools/bf16x3_repro.py --pair > $OUT/${TAG}_pair_packed.txt 2>&1; echo "packed exit $?"; grep "^pair " $OUT/${TAG}_pair_packed.txt | cut -c1-400; tail -1 $OUT/${TAG}_pair_packed.txt | cut -c1-300
timeout 400 python tools/bf16x3_repro.py --pair > $OUT/${TAG}_pair_product.txt 2>&1; echo "product exit $?"; grep "^pair " $OUT/${TAG}_pair_product.txt | cut -c1-400
