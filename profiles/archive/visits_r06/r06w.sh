#!/bin/bash
# round 6, visit w: VERDICT r5 #5 re-measured -- round 3's matrix-core input convolution (k_conv3_mfma, 240 registers, fp32 MFMA)
# spliced back as a lab build (-DBS_CONV3_MFMA) and run in today's pipeline (bf16x3 GEMM, transform kernels at priority 1)
TAG=${1:-r06w}
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
LAB="-DBS_CONV3_MFMA"
python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1
BITSWAP_HIPCC_EXTRA="$LAB" python -c "from bitswap_amd import build; print(build.build_hip())" || exit 1
{
echo "== alone"; python tools/conv3_time.py; BITSWAP_HIPCC_EXTRA="$LAB" python tools/conv3_time.py
echo "== the kernel's own test with the lab build (tolerances against torch; the bitwise parts compare the kernel with itself)"
BITSWAP_HIPCC_EXTRA="$LAB" timeout 600 python -m pytest tests/test_codec_gpu.py -q -m gpu -k "conv3_wino" 2>&1 | tail -3
for rep in 1 2 3; do
  echo "product"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "lab: k_conv3_mfma"; BITSWAP_HIPCC_EXTRA="$LAB" timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for shape in "--chains 100" "--workload imagenet4 --chains 1000"; do
  echo "$shape product"; timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "$shape lab: k_conv3_mfma"; BITSWAP_HIPCC_EXTRA="$LAB" timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_conv3_mfma.txt 2>&1
cat $OUT/${TAG}_conv3_mfma.txt
