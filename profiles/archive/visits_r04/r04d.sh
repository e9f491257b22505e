#!/bin/bash
# round 4 visit D: GEMM decomposition at small column counts (units per workgroup), alone
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for mu in 1 2 4; do for w in 1 2; do
  BITSWAP_GEMM_MIN_UNITS=$mu BITSWAP_GEMM_WGS_PER_CU=$w python tools/gemm_small.py > $OUT/r04d_gemm_mu${mu}_w${w}.json 2>/dev/null
done; done
python - <<PY
import json
tags = [(m, w) for m in (1, 2, 4) for w in (1, 2)]
d = {t: json.load(open("$OUT/r04d_gemm_mu%d_w%d.json" % t)) for t in tags}
print("shape".ljust(28), *[f"mu{m}w{w}".rjust(9) for m, w in tags])
for k in d[tags[0]]:
    if k == "ns3_units": continue
    print(k.ljust(28), *[f"{d[t][k]['us']:9.1f}" for t in tags], "same bits", len({d[t][k]["checksum"] for t in tags}) == 1)
PY
