#!/bin/bash
# round 5, visit b: new GPU tests (masked streams, divergence horizon), bf16x3 co-residency repro hunt (micro + codec), and the
# CU-mask A/B of the headline (bulk + serial arrangement, serial streams on 0 / 24 / 32 / 48 CUs) in both conv arithmetics
TAG=${1:-r05b}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "masked or horizon or bf16x3 or abi" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/${TAG}_pytest.log
timeout 900 python tools/bf16x3_repro.py --reps 30 --codec > $OUT/${TAG}_bf16x3_repro.txt 2>&1; tail -40 $OUT/${TAG}_bf16x3_repro.txt
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $OUT/${TAG}_mask_${name}.json 2> $OUT/${TAG}_mask_${name}.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${TAG}_mask_${name}.json") if l.startswith("{")][-1])
    print("$name", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms lossless", d["lossless"], d["config"]["conv_dtype"])
except Exception as e:
    print("$name FAILED", e, open("$OUT/${TAG}_mask_${name}.err").read()[-400:])
PY
}
run default_a X=1
run split_unmasked BITSWAP_GROUP_STREAMS=0
run mask32 BITSWAP_SERIAL_CUS=32 BITSWAP_GEMM_CUS=224
run mask24 BITSWAP_SERIAL_CUS=24 BITSWAP_GEMM_CUS=232
run mask48 BITSWAP_SERIAL_CUS=48 BITSWAP_GEMM_CUS=208
run default_b X=1
run bf16x3_default BITSWAP_GEMM_ARITH=bf16x3
run bf16x3_mask32 BITSWAP_GEMM_ARITH=bf16x3 BITSWAP_SERIAL_CUS=32 BITSWAP_GEMM_CUS=224
run bf16x3x9_default BITSWAP_GEMM_ARITH=bf16x3x9
