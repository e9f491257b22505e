#!/bin/bash
# round 5, visit B: (1) the packed-addition pattern alone (bs_debug_pk_probe) beside the bf16x3 GEMM / the fp32 GEMM / nothing;
# (2) the new GPU test of the forked bf16x3 codec; (3) headline with the opt-in arithmetics forked again: fp32 / bf16x3 / bf16x3x9
TAG=${1:-r05B}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/bf16x3_repro.py --pk > $OUT/${TAG}_pk_probe.txt 2>&1; echo "pk exit $?"; grep "^pk " $OUT/${TAG}_pk_probe.txt | cut -c1-300; tail -2 $OUT/${TAG}_pk_probe.txt | cut -c1-300
timeout 900 python -m pytest tests/test_codec_gpu.py tests/test_abi.py -m gpu -q -x -k "bf16x3 or abi or c_abi" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/${TAG}_pytest.log
run() { local name=$1; local chains=$2; shift; shift
  env "$@" timeout 600 python bench.py --chains $chains --no-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $OUT/${TAG}_${name}.json 2> $OUT/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${TAG}_${name}.json") if l.startswith("{")][-1])
    print("$name", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms lossless", d["lossless"])
except Exception as e:
    print("$name FAILED", e)
PY
}
run fp32 1000 X=1
run bf16x3 1000 BITSWAP_GEMM_ARITH=bf16x3
run bf16x3x9 1000 BITSWAP_GEMM_ARITH=bf16x3x9
run c100_fp32 100 X=1
run c100_bf16x3 100 BITSWAP_GEMM_ARITH=bf16x3
run c13_fp32 13 X=1
run c13_bf16x3 13 BITSWAP_GEMM_ARITH=bf16x3
