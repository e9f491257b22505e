#!/bin/bash
# round 6, visit i: the persistent GEMM after its unit-boundary fix (bitwise tests), SQ counters of the GEMM shapes alone, and the
# pipeline-shape experiments of tools/r06g.sh
TAG=${1:-r06i}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_codec_gpu.py -m gpu -q -k "bf16x3" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/${TAG}_pytest.log
bash tools/pmc_gemm.sh $TAG
bash tools/r06g.sh $TAG
