#!/bin/bash
# round 6, visit W2: k_conv3_wino with whole 16-byte window reads (-DBS_CONV3_WHOLE: 91 registers instead of 74; round 4 visit Y said
# 8 us faster alone, 7 ms slower in the step) in today's pipeline
TAG=${1:-r06W2}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
FL="-DBS_CONV3_WHOLE"
BITSWAP_HIPCC_EXTRA="$FL" python -c "from bitswap_amd import build; print(build.build_hip())"
B="python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:], open('/tmp/o.err').read()[-600:])
PY
}
{
python tools/conv3_time.py 2>/dev/null | tail -1; BITSWAP_HIPCC_EXTRA="$FL" python tools/conv3_time.py 2>/dev/null | tail -1
for rep in 1 2 3; do
  echo "dword window reads (default)"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "whole 16-byte window reads"; BITSWAP_HIPCC_EXTRA="$FL" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for rep in 1 2; do
  echo "100 chains, default"; timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "100 chains, whole reads"; BITSWAP_HIPCC_EXTRA="$FL" timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_conv3_whole.txt 2>&1
cat $OUT/${TAG}_conv3_whole.txt
