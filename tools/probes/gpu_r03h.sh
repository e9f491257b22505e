#!/bin/bash
TAG=${1:-r03h}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python tools/gemm_probe.py --knobs --only-own > $OUT/${TAG}_gemm_probe.txt 2>&1; echo "probe exit $?"; grep -v Warning $OUT/${TAG}_gemm_probe.txt | head -4
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/${TAG}_pytest.log
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; cut -c1-400 $OUT/${TAG}_bench.json; tail -2 $OUT/${TAG}_bench.err
exit 0
