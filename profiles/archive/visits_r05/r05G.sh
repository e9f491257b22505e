#!/bin/bash
# round 5, visit G: is the coder wavefronts' s_setprio 3 an ingredient?  Packed build (fails 2-5 % of forked bf16x3 runs) against the
# same build with BS_SERIAL_PRIO=0 in pop.hip / push.hip, record leg, 500 runs each, alternating
TAG=${1:-r05G}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
line = [l for l in txt.splitlines() if l.startswith("record ")]
d = json.loads(line[-1][7:]) if line else {"error": txt[-400:]}
print(sys.argv[2], {k: d.get(k) for k in ("runs", "lossless", "error")}, [(f["run"], f["bad_chains"]) for f in d.get("failures", [])][:20])
PY
}
for leg in a b; do
  BITSWAP_HIP_LIB=$PWD/bitswap_amd/csrc/libbitswap_hip_slp.so REPRO_RECORD_REPS=500 REPRO_MAX_FAIL=99 timeout 300 python tools/bf16x3_repro.py --record > $OUT/${TAG}_packed_prio3_$leg.txt 2>&1; summ $OUT/${TAG}_packed_prio3_$leg.txt "packed, s_setprio 3 ($leg)"
  BITSWAP_HIP_LIB=$PWD/bitswap_amd/csrc/libbitswap_hip_slp_prio0.so REPRO_RECORD_REPS=500 REPRO_MAX_FAIL=99 timeout 300 python tools/bf16x3_repro.py --record > $OUT/${TAG}_packed_prio0_$leg.txt 2>&1; summ $OUT/${TAG}_packed_prio0_$leg.txt "packed, s_setprio 0 ($leg)"
done
