#!/bin/bash
# round 6, visit t: how sensitive is the 1000-chain step to the VALU work of the serial pop kernels?  k_rans_pop_pivot with
# BS_POP_PAD extra (dead) float64 instructions per symbol (51 real ones): if +50 % of float64 work costs x %, a pop that needs half
# the VALU is worth about that much
TAG=${1:-r06t}
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
for pad in 25 50 100; do BITSWAP_HIPCC_EXTRA=-DBS_POP_PAD=$pad python -c "from bitswap_amd import build; print(build.build_hip())"; done
{
for rep in 1 2; do
  echo "pad 0";  timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  for pad in 25 50 100; do echo "pad $pad float64 instructions per symbol"; BITSWAP_HIPCC_EXTRA=-DBS_POP_PAD=$pad timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
} > $OUT/${TAG}_pop_pad.txt 2>&1
cat $OUT/${TAG}_pop_pad.txt
