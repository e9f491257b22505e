#!/bin/bash
# round 6, visit r: the 64 KB LDS claim of the transform passes (default since r06q) against no claim, both headline workloads,
# alternating, three repetitions
TAG=${1:-r06r}
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
  for wl in cifar8 imagenet4; do
    for lds in 0 65536; do
      echo "$wl 1000 chains, transform-pass LDS claim $lds"; BITSWAP_FUSED_LDS_MIN=$lds timeout 400 $B --workload $wl > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
    done
  done
done
for lds in 0 65536; do echo "cifar8 1500 chains, claim $lds"; BITSWAP_FUSED_LDS_MIN=$lds timeout 400 $B --chains 1500 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_fused_claim_ab.txt 2>&1
cat $OUT/${TAG}_fused_claim_ab.txt
