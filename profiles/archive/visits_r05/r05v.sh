#!/bin/bash
# round 5, visit v: bf16x3 hunt, record-and-replay leg.  The forked eager bf16x3 codec (shape 2, claimed kernel, 32 chains) with every
# stack kernel call KEPT (arguments and results by reference, no launch added, no buffer reused inside a run); after a failing run
# every call is repeated alone and compared bit by bit: which call does not repeat, on which stream, which elements.
TAG=${1:-r05v}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPRO_RECORD_REPS=${REPS:-160} REPRO_MAX_FAIL=3 timeout 560 python tools/bf16x3_repro.py --record > $OUT/${TAG}_bf16x3_record.txt 2>&1
echo "exit $?"; grep -E "^record|^PARTIAL" $OUT/${TAG}_bf16x3_record.txt | cut -c1-6000 | head -5
tail -3 $OUT/${TAG}_bf16x3_record.txt | cut -c1-600
