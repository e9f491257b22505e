#!/bin/bash
# round 6, visit U: the big GEMM launches on a companion stream of another HIP priority (BITSWAP_GEMM_STREAM_PRIO), fenced on both sides
# with events: high (-1) beside normal group streams, and normal (0) beside HIGH group streams (BITSWAP_GROUP_STREAM_PRIO=-1) -- does the
# dispatcher place the 438-register workgroups sooner / later, and does the step care?
TAG=${1:-r06U}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:], open('/tmp/o.err').read()[-900:])
PY
}
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
{
for rep in 1 2 3; do
  echo "default"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "GEMMs on a companion stream, same priority (the fences alone)"; BITSWAP_GEMM_STREAM_PRIO=0 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "GEMMs on a HIGH-priority companion stream"; BITSWAP_GEMM_STREAM_PRIO=-1 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "group streams HIGH, GEMMs on a normal companion stream"; BITSWAP_GROUP_STREAM_PRIO=-1 BITSWAP_GEMM_STREAM_PRIO=0 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_gemm_stream_prio.txt 2>&1
cat $OUT/${TAG}_gemm_stream_prio.txt
