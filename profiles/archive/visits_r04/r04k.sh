#!/bin/bash
export TMPDIR=/tmp
export BITSWAP_HIP_LIB=/tmp/libbitswap_full.so BITSWAP_HIPCC_EXTRA=-DBS_X3_FULL_REGS
python -m bitswap_amd.build > /dev/null 2>&1
ARITH=bf16x3 MODE=none timeout 600 python tools/visits/dbg3_bf16.py 2>&1 | grep "^mode"
ARITH=bf16x3 MODE=none timeout 600 python tools/visits/dbg3_bf16.py 2>&1 | grep "^mode"
