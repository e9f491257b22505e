#!/bin/bash
# round 6, visit p: right-sizing the HBM-bound transform passes -- fewer of their workgroups per CU (an LDS claim), so that they
# take a smaller part of every CU while the other chain group's GEMM / table kernels run: does the pipeline overlap better?
TAG=${1:-r06p}
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
  for lds in 0 65536 82000 112000 163840; do
    echo "k_wino_fused LDS claim >= $lds B (at most $(( lds > 0 ? 163840 / lds : 6 )) workgroups per CU), 1000 chains"
    BITSWAP_FUSED_LDS_MIN=$lds timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  done
done
} > $OUT/${TAG}_fused_footprint_ab.txt 2>&1
cat $OUT/${TAG}_fused_footprint_ab.txt
