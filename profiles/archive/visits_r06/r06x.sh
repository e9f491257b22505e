#!/bin/bash
# round 6, visit x: the persistent wave-specialised GEMM with the U fragments through LDS as well (BITSWAP_BF16X3_ULDS=1): the
# multiplying wavefronts issue no global load, so nothing of theirs queues behind their own stores.  Bits, time alone, pipeline.
TAG=${1:-r06x}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_codec_gpu.py -m gpu -q -x -k "bf16x3_gemm" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/${TAG}_pytest.log
timeout 600 python tools/gemm_shapes_time.py > $OUT/${TAG}_gemm_times.txt 2>&1; cat $OUT/${TAG}_gemm_times.txt
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
  echo "default (one unit per workgroup at 1000 chains)"; timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "persistent"; BITSWAP_BF16X3_PERSISTENT=1 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "persistent, U through LDS"; BITSWAP_BF16X3_PERSISTENT=1 BITSWAP_BF16X3_ULDS=1 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for shape in "--chains 100" "--workload imagenet4 --chains 1000"; do
  echo "$shape default"; timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "$shape persistent, U through LDS"; BITSWAP_BF16X3_PERSISTENT=1 BITSWAP_BF16X3_ULDS=1 timeout 400 $B $shape > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_pipeline_ab.txt 2>&1
cat $OUT/${TAG}_pipeline_ab.txt
