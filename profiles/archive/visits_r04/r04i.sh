#!/bin/bash
# round 4 visit I: bf16x3 route end to end: tests, error table, bench with the route on / off
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16x3 or abi" > $OUT/r04i_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/r04i_pytest.log
timeout 600 python tools/bf16x3_error.py $OUT/r04i_bf16x3_error.json 2>&1 | tail -1
B="python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2"
for a in fp32 bf16x3 bf16x3x9; do
  BITSWAP_GEMM_ARITH=$a timeout 300 $B 2> $OUT/r04i_$a.err | tail -1 > $OUT/r04i_bench_$a.json
  python -c "
import json; d=json.load(open('$OUT/r04i_bench_$a.json')); print('$a', round(d['value']/1e6,3), d['ms_per_step'], d['lossless'], d['config']['conv_dtype'], d['bits_per_dim'])" || tail -5 $OUT/r04i_$a.err
done
BITSWAP_GEMM_ARITH=bf16x3 timeout 300 $B --chains 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bf16x3 100 chains', round(d['value']/1e6,3), d['ms_per_step'], d['lossless'])"
BF16X3_OUT=$OUT/r04i_bf16x3_probe.json timeout 300 python tools/bf16x3_probe.py --quick 2>&1 | grep "^T" | cut -c1-400
