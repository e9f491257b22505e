#!/bin/bash
# round 6, visit v: follow-up of r06u -- the transform kernels (k_wino_fused, k_conv3_wino) at s_setprio 1 / 2 / 3, with and without the
# GEMM at 2; three repetitions at 1000 chains, then the other shapes for the candidates
TAG=${1:-r06v}
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
VARIANTS=("" "-DBS_XFORM_PRIO=1" "-DBS_XFORM_PRIO=2" "-DBS_XFORM_PRIO=3" "-DBS_GEMM_PRIO_CONS=2 -DBS_GEMM_PRIO_PROD=2 -DBS_XFORM_PRIO=1"
          "-DBS_GEMM_PRIO_CONS=1 -DBS_GEMM_PRIO_PROD=1 -DBS_XFORM_PRIO=2")
for v in "${VARIANTS[@]}"; do BITSWAP_HIPCC_EXTRA="$v" python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1; done
{
for rep in 1 2 3; do
  for v in "${VARIANTS[@]}"; do echo "flags: ${v:-none}"; BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for shape in "--chains 100" "--chains 13" "--workload imagenet4 --chains 1000" "--chains 1500"; do
  for v in "" "-DBS_XFORM_PRIO=1" "-DBS_XFORM_PRIO=2" "-DBS_GEMM_PRIO_CONS=2 -DBS_GEMM_PRIO_PROD=2 -DBS_XFORM_PRIO=1"; do
    echo "$shape  flags: ${v:-none}"; BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  done
done
} > $OUT/${TAG}_prio.txt 2>&1
cat $OUT/${TAG}_prio.txt
