#!/bin/bash
# round 5, visit d: bf16x3 -- statistics on the scenario that failed in visit c (unclaimed shape-2 kernel, eager forked codec, 32
# chains) with counted waits and with vmcnt(0) everywhere, the small-register neighbours in the micro loop; then the whole GPU suite
TAG=${1:-r05d}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/bf16x3_repro.py --focus --small --reps 20 > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep "^shape\|^codec" $OUT/${TAG}_bf16x3_repro.txt | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/${TAG}_pytest.log
