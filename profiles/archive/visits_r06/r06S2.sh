#!/bin/bash
# round 6, visit S2: the size rule for the cache policy of M / V (ordinary stores and loads up to 128 MB, nontemporal above) against
# always-nontemporal (BITSWAP_BF16X3_PLAIN_STORE=0 BITSWAP_FUSED_PLAIN=0: the code before), on the few-chain shapes and the headline
TAG=${1:-r06S2}
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
timeout 900 python -m pytest tests/test_codec_gpu.py -m gpu -q -x -k "bf16x3 or wino" 2>&1 | tail -3
{
for shape in "--scaling strong --total-chains 100" "--chains 50 --groups 1" "--chains 25 --groups 1" "--chains 13" "--workload imagenet4 --scaling strong --total-chains 100" "--workload imagenetcrop4 --scaling strong --total-chains 100 --steps 16" "--chains 200" "--chains 400"; do
  for rep in 1 2; do
    echo "$shape: size rule"; timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
    echo "$shape: always nontemporal"; BITSWAP_BF16X3_PLAIN_STORE=0 BITSWAP_FUSED_PLAIN=0 timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  done
done
for rep in 1 2; do
  echo "1000 chains: size rule"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "1000 chains: always nontemporal"; BITSWAP_BF16X3_PLAIN_STORE=0 BITSWAP_FUSED_PLAIN=0 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_cache_rule.txt 2>&1
cat $OUT/${TAG}_cache_rule.txt
