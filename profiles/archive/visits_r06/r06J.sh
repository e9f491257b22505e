#!/bin/bash
# round 6, visit J: channels per block of the input convolution (BITSWAP_CONV3_CPB; the rule picks 16 at 500 blocks) in today's pipeline
TAG=${1:-r06J}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
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
for rep in 1 2 3; do
  echo "1000 chains, rule (16)"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  for cpb in 8 32 64; do echo "1000 chains, cpb $cpb"; BITSWAP_CONV3_CPB=$cpb timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for rep in 1 2; do
  echo "100 chains, rule"; timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  for cpb in 8 16 32; do echo "100 chains, cpb $cpb"; BITSWAP_CONV3_CPB=$cpb timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
} > $OUT/${TAG}_conv3_cpb.txt 2>&1
cat $OUT/${TAG}_conv3_cpb.txt
