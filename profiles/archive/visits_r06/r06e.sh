#!/bin/bash
# round 6, visit e: bf16x3 (wave-specialised shape) as the DEFAULT conv arithmetic: (1) error tables of every conv stack of all four
# workloads against float64 torch modules, fp32 route beside it, and the rate difference; (2) the whole GPU suite; (3) the bench
TAG=${1:-r06e}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python tools/bf16x3_error.py $OUT/${TAG}_bf16x3_error.json --workload all > $OUT/${TAG}_bf16x3_error.log 2>&1; echo "error tool exit $?"; tail -5 $OUT/${TAG}_bf16x3_error.log | cut -c1-400
bash tools/visit.sh $TAG suite bench
