#!/bin/bash
# visit H2: pop kernel with 4 rows of prefetch (library built with -DBS_POP_PF=4) against 8: bench A/B on one box
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for rep in 1 2; do for p in 8 4; do
  if [ $p = 8 ]; then unset BITSWAP_HIP_LIB; else export BITSWAP_HIP_LIB=$PWD/tools/probes/_build/libbitswap_pf$p.so; fi
  echo -n "pop prefetch $p rows: "; $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
exit 0
