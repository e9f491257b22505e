#!/bin/bash
# round 6, visit g: with the new GEMM and table kernels, is the launch shape of the pipeline still the right one?  chains per GPU x
# chain groups; chains per wavefront of the table kernel (BITSWAP_TABLE_NB)
TAG=${1:-r06g}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
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
for cfg in "1000 2" "1000 3" "1500 2" "1500 3" "2000 2" "2000 4" "750 2" "1000 2"; do
  set -- $cfg
  echo "cifar8, $1 chains / $2 groups"; timeout 500 $B --chains $1 --groups $2 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for nb in 4 8 16; do
  echo "cifar8 1000 / 2, BITSWAP_TABLE_NB=$nb"; BITSWAP_TABLE_NB=$nb timeout 500 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for cfg in "1000 2" "1500 3" "2000 2"; do
  set -- $cfg
  echo "imagenet4, $1 chains / $2 groups"; timeout 500 $B --workload imagenet4 --chains $1 --groups $2 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_shape_ab.txt 2>&1
cat $OUT/${TAG}_shape_ab.txt
