#!/bin/bash
# visit C: the in-wave pipelined GEMM against the visit-A/B kernel (variant 0): correctness, then long loops
TAG=${1:-r03C}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_codec_gpu.py -q -x -k "wino_gemm or own_gemm or winograd_convs" 2>&1 | tail -3
timeout 900 python tools/gemm_probe.py --knobs > $OUT/${TAG}_gemm_probe.txt 2>&1; echo "probe exit $?"; cat $OUT/${TAG}_gemm_probe.txt
exit 0
