#!/bin/bash
# round 6, visit Z2: r06Y2 showed how much the spec 4 table kernel loses from 5 to 4 wavefronts per SIMD (+7.5 % on the step) -- then
# what does a sixth buy?  -DBS_SPEC4_WAVES=6: 80 registers, 12-18 of them spilled to scratch
TAG=${1:-r06Z2}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
FL="-DBS_SPEC4_WAVES=6"
BITSWAP_HIPCC_EXTRA="$FL" python -c "from bitswap_amd import build; print(build.build_hip())"
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
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "logistic or table" 2>&1 | tail -2
BITSWAP_HIPCC_EXTRA="$FL" timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "logistic or table" 2>&1 | tail -2
for rep in 1 2 3; do
  echo "5 wavefronts per SIMD (96 registers)"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "6 wavefronts per SIMD (80 registers + spills)"; BITSWAP_HIPCC_EXTRA="$FL" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_table_waves.txt 2>&1
cat $OUT/${TAG}_table_waves.txt
