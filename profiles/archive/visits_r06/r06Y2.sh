#!/bin/bash
# round 6, visit Y2: table kernels at 4 / 3 workgroups per CU instead of 5 (an LDS claim they do not use: BITSWAP_TABLE_LDS_CLAIM) --
# the VALU-bound table kernel and the HBM-bound transform passes of the other chain group are complementary; at 5 x 96 registers per
# SIMD nothing fits beside the tables, at 4 x 96 a 74-register transform wavefront does
TAG=${1:-r06Y2}
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
  for c in 0 40000 53000; do echo "1000 chains, table LDS claim $c"; BITSWAP_TABLE_LDS_CLAIM=$c timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for c in 0 40000; do echo "imagenet4, claim $c"; BITSWAP_TABLE_LDS_CLAIM=$c timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for c in 0 40000; do echo "100 chains, claim $c"; BITSWAP_TABLE_LDS_CLAIM=$c timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_table_claim.txt 2>&1
cat $OUT/${TAG}_table_claim.txt
