#!/bin/bash
# round 5, visit F: two conv stacks side by side and nothing else, packed build and product build
TAG=${1:-r05F}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SLP=$PWD/bitswap_amd/csrc/libbitswap_hip_slp.so
BITSWAP_HIP_LIB=$SLP timeout 500 python tools/bf16x3_repro.py --stacks > $OUT/${TAG}_stacks_packed.txt 2>&1; echo "packed exit $?"; grep "^stacks " $OUT/${TAG}_stacks_packed.txt | cut -c1-900; tail -1 $OUT/${TAG}_stacks_packed.txt | cut -c1-300
timeout 500 python tools/bf16x3_repro.py --stacks > $OUT/${TAG}_stacks_product.txt 2>&1; echo "product exit $?"; grep "^stacks " $OUT/${TAG}_stacks_product.txt | cut -c1-500
