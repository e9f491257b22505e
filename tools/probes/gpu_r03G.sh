#!/bin/bash
# visit G3: work-stealing GEMM with the cheap end-of-range check: correctness, exclusive speed, bench A/B
TAG=${1:-r03G3}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_codec_gpu.py -q -x -k "wino_gemm or own_gemm or winograd_convs" 2>&1 | tail -3
timeout 600 python tools/gemm_probe.py --only-own > $OUT/${TAG}_gemm_probe.txt 2>&1; cut -c1-200 $OUT/${TAG}_gemm_probe.txt | tail -11
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for rep in 1 2; do for w in 1 0; do
  echo -n "GEMM steal $w: "; BITSWAP_GEMM_STEAL=$w $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
exit 0
