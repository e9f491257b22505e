#!/bin/bash
# round 6, visit s: which transform flavour should claim LDS?  <6,6> (the 3x3 layers) only, <8,8> (the 5x5 layers) only, all, none
TAG=${1:-r06s}
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
for rep in 1 2; do
  for wl in "cifar8 1000" "imagenet4 1000" "cifar8 1500"; do
    set -- $wl
    echo "$1 $2 chains: no claim";       BITSWAP_FUSED_LDS_MIN=0 timeout 400 $B --workload $1 --chains $2 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
    echo "$1 $2 chains: <6,6> claims 64 KB"; BITSWAP_FUSED_LDS_MIN=65536 BITSWAP_FUSED_LDS_WHICH=66 timeout 400 $B --workload $1 --chains $2 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
    echo "$1 $2 chains: <8,8> claims 64 KB"; BITSWAP_FUSED_LDS_MIN=65536 BITSWAP_FUSED_LDS_WHICH=88 timeout 400 $B --workload $1 --chains $2 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
    echo "$1 $2 chains: all claim 64 KB";    BITSWAP_FUSED_LDS_MIN=65536 timeout 400 $B --workload $1 --chains $2 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  done
done
} > $OUT/${TAG}_claim_by_flavour.txt 2>&1
cat $OUT/${TAG}_claim_by_flavour.txt
