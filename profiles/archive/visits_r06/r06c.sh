#!/bin/bash
# round 6, visit c: CDF spec 4 (blocks of 8 bins + one Newton correction per quotient; table kernel at 5 wavefronts per SIMD):
# (1) parity: the hip-parity suite (every spec-parametrised test now includes 4) + the full-width word-parity cases; (2) the
# kernels alone (microbench, spec 1..4 side by side) at 500 and 13 chains; (3) headline step, spec 3 against spec 4, alternating
TAG=${1:-r06c}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_abi.py -m gpu -q -x > $OUT/${TAG}_pytest_parity.log 2>&1; echo "parity exit $?"; tail -3 $OUT/${TAG}_pytest_parity.log
timeout 900 python -m pytest tests/test_codec_gpu.py -m gpu -q -x -k "full_width_oracle_word_parity or bits_per_dim or round_trip_and_oracle" > $OUT/${TAG}_pytest_codec.log 2>&1; echo "codec exit $?"; tail -3 $OUT/${TAG}_pytest_codec.log
timeout 300 python tools/microbench.py --B 500 > $OUT/${TAG}_micro.json 2> $OUT/${TAG}_micro.err; echo "micro exit $?"
timeout 300 python tools/microbench.py --B 13 > $OUT/${TAG}_micro13.json 2>> $OUT/${TAG}_micro.err
python - <<PY
import json
for f in ("$OUT/${TAG}_micro.json", "$OUT/${TAG}_micro13.json"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "failed", e); continue
    for k, v in d.items():
        if isinstance(v, dict) and "spec3" in v:
            for sp in ("spec1", "spec2", "spec3", "spec4"):
                if sp in v:
                    print(f.split("_")[-1], k, sp, {a: round(b * 1e6, 1) for a, b in v[sp].items() if isinstance(b, float)})
PY
B="python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:])
PY
}
{
for rep in 1 2 3; do
  for sp in 3 4; do echo "cdf spec $sp, 1000 chains"; timeout 400 $B --cdf-spec $sp > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
done
for sp in 3 4; do echo "cdf spec $sp, 100 chains"; timeout 400 $B --cdf-spec $sp --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for sp in 3 4; do echo "cdf spec $sp, 13 chains"; timeout 400 $B --cdf-spec $sp --chains 13 --groups 1 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
for sp in 3 4; do echo "cdf spec $sp, imagenet4 1000 chains"; timeout 400 $B --cdf-spec $sp --workload imagenet4 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json; done
} > $OUT/${TAG}_spec_ab.txt 2>&1
cat $OUT/${TAG}_spec_ab.txt
