#!/bin/bash
# round 6, visit L: soak of the final build -- long lossless round trips (every chain's decoded blocks compared with what was coded,
# the stream returned to its initial words) on the shapes of the bench, several hundred block steps each
TAG=${1:-r06L}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['steps']} steps  {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}  crc={(d.get('stream_gather') or {}).get('crc32_of_streams_in_chain_order')}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:], open('/tmp/o.err').read()[-600:])
PY
}
{
for rep in 1 2; do
  echo "cifar8 1000 chains x 120 blocks"; timeout 900 $B --steps 120 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "imagenet4 1000 chains x 120 blocks"; timeout 900 $B --steps 120 --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "cifar8 100 chains (forked step) x 400 blocks"; timeout 900 $B --steps 400 --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "cifar8 13 chains x 400 blocks"; timeout 900 $B --steps 400 --chains 13 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "imagenetcrop4 100 ragged chains"; timeout 900 $B --steps 16 --workload imagenetcrop4 --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_soak.txt 2>&1
cat $OUT/${TAG}_soak.txt
