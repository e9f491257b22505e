#!/bin/bash
# round 6, the full visit on the round's code: GPU suite + smoke, the driver's bench command with all twenty shapes, timed-region
# profile + PMC passes, SQ counters of the table kernels, BASELINE configs through the reference-named scripts, the CLI smoke
TAG=${1:-r06z}
OUT=$PWD/gpurun_out; mkdir -p $OUT
EXTRA=all bash tools/visit.sh $TAG suite bench profile valu configs
timeout 900 bash tools/cli_smoke.sh > $OUT/${TAG}_cli_smoke.txt 2>&1; echo "cli smoke exit $?"; tail -4 $OUT/${TAG}_cli_smoke.txt | cut -c1-300
