#!/bin/bash
# round 6, visit u: issue priority of the bulk kernels.  A GEMM of one chain group starts beside an OLDER table / transform kernel
# of the other group; the SIMD arbitrates by priority, then age.  Does s_setprio 1/2 in the GEMM's wavefronts (consumers, producers
# or both), or in the transform kernels, shorten the step?  (the serial coder kernels run at 3)
TAG=${1:-r06u}
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
VARIANTS=("" "-DBS_GEMM_PRIO_CONS=1 -DBS_GEMM_PRIO_PROD=1" "-DBS_GEMM_PRIO_CONS=2 -DBS_GEMM_PRIO_PROD=2" "-DBS_GEMM_PRIO_CONS=2 -DBS_GEMM_PRIO_PROD=1"
          "-DBS_GEMM_PRIO_CONS=1 -DBS_GEMM_PRIO_PROD=2" "-DBS_GEMM_PRIO_CONS=3 -DBS_GEMM_PRIO_PROD=3" "-DBS_XFORM_PRIO=1"
          "-DBS_GEMM_PRIO_CONS=2 -DBS_GEMM_PRIO_PROD=2 -DBS_XFORM_PRIO=1")
for v in "${VARIANTS[@]}"; do BITSWAP_HIPCC_EXTRA="$v" python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1; done
{
for rep in 1 2; do
  for v in "${VARIANTS[@]}"; do echo "flags: ${v:-none}"; BITSWAP_HIPCC_EXTRA="$v" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
} > $OUT/${TAG}_prio.txt 2>&1
cat $OUT/${TAG}_prio.txt
