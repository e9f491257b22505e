#!/bin/bash
# round 6, visit I: chains per wavefront of the table kernels (BITSWAP_TABLE_NB; default 8) again, now that the spec 4 kernel runs five
# wavefronts per SIMD: 1000 chains (2 groups of 500) and 100 chains
TAG=${1:-r06I}
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
  for nb in 8 12 16 32; do echo "1000 chains, nb $nb"; BITSWAP_TABLE_NB=$nb timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for rep in 1 2 3; do
  for nb in 2 4 8 16; do echo "100 chains, nb $nb"; BITSWAP_TABLE_NB=$nb timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for nb in 8 16; do echo "imagenet4 1000 chains, nb $nb"; BITSWAP_TABLE_NB=$nb timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for nb in 1 2 4 8; do echo "13 chains, nb $nb"; BITSWAP_TABLE_NB=$nb timeout 400 $B --chains 13 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_table_nb.txt 2>&1
cat $OUT/${TAG}_table_nb.txt
