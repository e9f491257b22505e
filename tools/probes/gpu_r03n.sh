#!/bin/bash
TAG=${1:-r03n}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 300 -k "pivot" > $OUT/${TAG}_pytest1.log 2>&1; echo "pivot test exit $?"; tail -15 $OUT/${TAG}_pytest1.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/${TAG}_pytest.log
timeout 300 python tools/microbench.py --B 400 > $OUT/${TAG}_micro.json 2>$OUT/${TAG}_micro.err; tail -2 $OUT/${TAG}_micro.err
timeout 600 python bench.py --no-extra --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; cut -c1-300 $OUT/${TAG}_bench.json
BITSWAP_PIVOT=0 timeout 600 python bench.py --no-extra --no-cpu-baseline > $OUT/${TAG}_bench_nopivot.json 2>> $OUT/${TAG}_bench.err; cut -c1-200 $OUT/${TAG}_bench_nopivot.json
exit 0
