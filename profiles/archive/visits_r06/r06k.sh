#!/bin/bash
# round 6, visit k: the cache-resident chunk experiment again, now that the GEMM is the bf16x3 kernel (HBM-bound alone: profiles/
# r06j_gemm_ws_lab.txt) instead of the MFMA-bound fp32 one
TAG=${1:-r06k}
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
for nb in 0 250 167 125 0; do echo "cifar8 1000 chains / 2 groups, nn_batch $nb"; timeout 500 $B --nn-batch $nb > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for nb in 0 250; do echo "imagenet4 1000 chains / 2 groups, nn_batch $nb"; timeout 500 $B --workload imagenet4 --nn-batch $nb > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_chunks.txt 2>&1
cat $OUT/${TAG}_chunks.txt
