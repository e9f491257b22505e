#!/bin/bash
# round 5, visit r: the CLIs end to end (tools/cli_smoke.sh) and BASELINE configs 1, 2, 3, 5 at the reference's own shape through
# the reference-named scripts (tools/config_runs.sh), on the round's code (CDF spec 3, reference draw order)
TAG=${1:-r05r}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/cli_smoke.sh > $OUT/${TAG}_cli_smoke.txt 2>&1; echo "cli_smoke exit $?"; tail -12 $OUT/${TAG}_cli_smoke.txt | cut -c1-300
bash tools/config_runs.sh > $OUT/${TAG}_configs.txt 2>&1; cat $OUT/${TAG}_configs.txt | cut -c1-300
