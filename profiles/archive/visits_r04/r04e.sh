#!/bin/bash
# round 4 visit E: hardware queues vs streams of the forked groups (HIP multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues by default)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 12 --warmup 3"
one() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 $B "$@" 2> $OUT/r04e_$tag.err | tail -1 > $OUT/r04e_$tag.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/r04e_$tag.json")); print("$tag", round(d["value"]/1e6, 3), "Mpx/s", d["ms_per_step"], "ms/step lossless", d["lossless"], "groups", d["config"]["chain_groups"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/r04e_$tag.err").read()[-800:])
PY
}
for q in 4 8 16; do
one c100_g2_q$q GPU_MAX_HW_QUEUES=$q -- --chains 100 --groups 2
one c100_g3_q$q GPU_MAX_HW_QUEUES=$q -- --chains 100 --groups 3
one c100_g4_q$q GPU_MAX_HW_QUEUES=$q -- --chains 100 --groups 4
done
one c100_g1_q8 GPU_MAX_HW_QUEUES=8 -- --chains 100 --groups 1
one c13_q8 GPU_MAX_HW_QUEUES=8 -- --chains 13 --groups 1
one c13_g2_q8 GPU_MAX_HW_QUEUES=8 -- --chains 13 --groups 2
one c1000_q8 GPU_MAX_HW_QUEUES=8 -- --chains 1000 --groups 2 --steps 6 --warmup 2
one c100_g2_nograph BITSWAP_GROUP_GRAPHS=0 -- --chains 100 --groups 2
