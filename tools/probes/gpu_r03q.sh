#!/bin/bash
TAG=${1:-r03q}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/${TAG}_pytest.log
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; cut -c1-200 $OUT/${TAG}_bench.json
exit 0
