#!/bin/bash
# round 4 visit G: the bf16x3 GEMM alone (error against float64, time against the fp32 MFMA kernel) + its tests, per kernel shape
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for shape in 2 1; do
echo "== BITSWAP_BF16X3_SHAPE=$shape"
BITSWAP_BF16X3_SHAPE=$shape BF16X3_OUT=$OUT/r04g_bf16x3_probe_shape$shape.json timeout 600 python tools/bf16x3_probe.py 2>&1 | grep "^T" | python -c "
import sys,json,re
for l in sys.stdin:
    m=re.match(r'(T\d+ \S+) (\{.*\})', l)
    d=json.loads(m.group(2)); print(m.group(1), {k:(round(v['rms_err_over_rms']*1e7,3), round(v['max_err_over_range']*1e7,2), v['ms'], v['TFLOPs_equiv'], v.get('prefix_columns_same_bits')) for k,v in d.items()})
"
done
for shape in 2 1; do BITSWAP_BF16X3_SHAPE=$shape timeout 900 python -m pytest tests -m gpu -x -q -k "bf16x3" 2>&1 | tail -2; done
