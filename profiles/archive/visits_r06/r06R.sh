#!/bin/bash
# round 6, visit R: cache policy of the tensors between the GEMM and the transform passes -- M through ordinary instead of nontemporal
# stores (-DBS_GEMM_PLAIN_STORE), the transform passes' M loads / V stores ordinary (BITSWAP_FUSED_PLAIN=1), both
TAG=${1:-r06R}
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
X="-DBS_GEMM_PLAIN_STORE"
BITSWAP_HIPCC_EXTRA="$X" python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1
{
for rep in 1 2 3; do
  echo "default"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "GEMM: ordinary stores of M"; BITSWAP_HIPCC_EXTRA="$X" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "transform passes: ordinary loads of M / stores of V"; BITSWAP_FUSED_PLAIN=1 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "both"; BITSWAP_FUSED_PLAIN=1 BITSWAP_HIPCC_EXTRA="$X" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for v in "" "$X"; do echo "100 chains, flags: ${v:-none}"; BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "100 chains, flags: ${v:-none} + FUSED_PLAIN"; BITSWAP_FUSED_PLAIN=1 BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_cache_policy.txt 2>&1
cat $OUT/${TAG}_cache_policy.txt
