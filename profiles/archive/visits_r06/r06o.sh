#!/bin/bash
# round 6, visit o: does co-residency with the GEMM matter?  k_wino_fused<6,6> compiled for 64 registers (fits beside a GEMM
# workgroup: 2 x 224 + 64 = 512 per SIMD) against its natural 66 (does not), in the 1000-chain pipeline, alternating
TAG=${1:-r06o}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}  crc={(d.get('stream_gather') or {}).get('crc32_of_streams_in_chain_order')}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:], open('/tmp/o.err').read()[-600:])
PY
}
echo "(this visit needed the BS_FUSED66_WAVES macro, removed after it; kept for the record)"; exit 0
{
for rep in 1 2 3; do
  echo "k_wino_fused<6,6> 66 registers, 1000 chains"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "k_wino_fused<6,6> 64 registers, 1000 chains"; BITSWAP_HIPCC_EXTRA=-DBS_FUSED66_WAVES=8 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
echo "66 registers, imagenet4"; timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
echo "64 registers, imagenet4"; BITSWAP_HIPCC_EXTRA=-DBS_FUSED66_WAVES=8 timeout 400 $B --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
} > $OUT/${TAG}_fused64_ab.txt 2>&1
cat $OUT/${TAG}_fused64_ab.txt
