#!/bin/bash
# visit K: the reference's own shape (100 chains) with 1 / 2 / 4 chain groups
B="python bench.py --chains 100 --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-roofline"
for g in 1 2 4 8 2 4; do
  echo -n "100 chains, $g group(s): "; $B --groups $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('graphs'))"
done
echo -n "200 chains, 4 groups: "; python bench.py --chains 200 --groups 4 --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
exit 0
