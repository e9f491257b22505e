#!/bin/bash
# visit F: what the co-resident serial coder kernels cost the persistent GEMM: workgroups per CU A/B on one box + timeline
TAG=${1:-r03F}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for rep in 1 2; do for w in 2 1; do
  echo -n "GEMM workgroups per CU $w: "; BITSWAP_GEMM_WGS_PER_CU=$w $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
export BITSWAP_GEMM_WGS_PER_CU=1
BCMD="python $PWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --no-roofline"
( cd /tmp && rm -rf prof_tl && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl --output-format csv -- $BCMD > $OUT/${TAG}_tl.log 2>&1 )
python tools/prof_summary.py timeline /tmp/prof_tl $OUT/${TAG}_timeline_1wg.json 0 0 | grep "^void\|bulk\|serial"
exit 0
