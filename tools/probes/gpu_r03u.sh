#!/bin/bash
TAG=${1:-r03u}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_driver_cmd.json 2> $OUT/${TAG}_driver_cmd.err; echo "driver cmd exit $?"; cut -c1-260 $OUT/${TAG}_driver_cmd.json
BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --chains 200 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/${TAG}_2rank_gloo.json 2> $OUT/${TAG}_2rank.err; echo "2-rank exit $?"; cut -c1-400 $OUT/${TAG}_2rank_gloo.json; tail -2 $OUT/${TAG}_2rank.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/cli_smoke.sh > $OUT/${TAG}_cli_smoke.txt 2>&1; echo "cli smoke exit $?"; tail -12 $OUT/${TAG}_cli_smoke.txt
exit 0
