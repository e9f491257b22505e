#!/bin/bash
# round 6, visit G: the one shape that got slower in the final visit (imagenetcrop4, 25 ragged chains: 6.2 -> 8.2 ms) -- the priorities
# of r06v / r06y or a hiccup?  Default build against one without the priorities, three repetitions.
TAG=${1:-r06G}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null --workload imagenetcrop4 --scaling strong --steps 16 --warmup 5"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:], open('/tmp/o.err').read()[-600:])
PY
}
NOPRIO="-DBS_XFORM_PRIO=0 -DBS_GEMM32_PRIO=0"
BITSWAP_HIPCC_EXTRA="$NOPRIO" python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1
{
for n in 25 50 13; do
  for rep in 1 2 3; do
    echo "$n chains, default"; timeout 400 $B --total-chains $n > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
    echo "$n chains, no priorities"; BITSWAP_HIPCC_EXTRA="$NOPRIO" timeout 400 $B --total-chains $n > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  done
done
} > $OUT/${TAG}_crop_ab.txt 2>&1
cat $OUT/${TAG}_crop_ab.txt
