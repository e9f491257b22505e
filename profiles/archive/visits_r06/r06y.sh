#!/bin/bash
# round 6, visit y: priority knobs on the remaining bulk kernels, on top of the transform kernels' 1: the table kernels (k_logistic) at
# 1, the float32 GEMM of the heads at 1, the transforms at 2 with the tables at 1
TAG=${1:-r06y}
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
VARIANTS=("" "-DBS_TABLE_PRIO=1" "-DBS_GEMM32_PRIO=1" "-DBS_TABLE_PRIO=1 -DBS_XFORM_PRIO=2" "-DBS_TABLE_PRIO=1 -DBS_XFORM_PRIO=2 -DBS_GEMM32_PRIO=2"
          "-DBS_TABLE_PRIO=2 -DBS_XFORM_PRIO=1")
for v in "${VARIANTS[@]}"; do BITSWAP_HIPCC_EXTRA="$v" python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1; done
{
for rep in 1 2 3; do
  for v in "${VARIANTS[@]}"; do echo "flags: ${v:-none}"; BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for shape in "--chains 100" "--workload imagenet4 --chains 1000"; do
  for v in "" "-DBS_TABLE_PRIO=1" "-DBS_TABLE_PRIO=1 -DBS_XFORM_PRIO=2"; do
    echo "$shape  flags: ${v:-none}"; BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  done
done
} > $OUT/${TAG}_prio.txt 2>&1
cat $OUT/${TAG}_prio.txt
