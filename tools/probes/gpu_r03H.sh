#!/bin/bash
# visit H3: pop kernel with 4 rows of prefetch (library built with -DBS_POP_PF=4) against 8: parity tests with it, bench A/B
export TMPDIR=/tmp
BITSWAP_HIP_LIB=$PWD/tools/probes/_build/libbitswap_pf4.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_codec_gpu.py -q -x -k "pivot or full_width_oracle or lowrate" 2>&1 | tail -2
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for p in 8 4 8 4; do
  if [ $p = 8 ]; then unset BITSWAP_HIP_LIB; else export BITSWAP_HIP_LIB=$PWD/tools/probes/_build/libbitswap_pf$p.so; fi
  echo -n "pop prefetch $p rows: "; $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
exit 0
