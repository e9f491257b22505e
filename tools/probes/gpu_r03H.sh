#!/bin/bash
# visit H: issue priority of the serial coder kernels (library built with -DBS_SERIAL_PRIO=p): bench A/B on one box
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for rep in 1 2; do for p in 3 0 1; do
  if [ $p = 3 ]; then unset BITSWAP_HIP_LIB; else export BITSWAP_HIP_LIB=$PWD/tools/probes/_build/libbitswap_prio$p.so; fi
  echo -n "serial prio $p: "; $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
exit 0
