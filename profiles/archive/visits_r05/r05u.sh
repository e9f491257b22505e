#!/bin/bash
# round 5, visit u: the pivot pop on a diet (bin 0 of the group from the pivots: the group below is not evaluated; two-instruction
# DPP exchanges; parameters by one LDS broadcast; half-wave exchange by v_permlane32_swap; scalar renormalisation test).
# (1) the whole GPU suite on the new library; (2) the pop alone, old library vs new on the SAME box (tools/microbench.py);
# (3) the headline step, old / new / old / new.
TAG=${1:-r05u}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$PWD/bitswap_amd/csrc/libbitswap_hip_oldpop.so
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/${TAG}_pytest.log
for v in old new; do
  if [ $v = old ]; then export BITSWAP_HIP_LIB=$OLD; else unset BITSWAP_HIP_LIB; fi
  timeout 600 python tools/microbench.py --B 1000 > $OUT/${TAG}_micro_$v.json 2> $OUT/${TAG}_micro_$v.err
  python - <<PY
import json
d = json.load(open("$OUT/${TAG}_micro_$v.json"))
for k, r in d.items():
    if isinstance(r, dict):
        print("$v", k, {s: {q: round(x * 1e3, 4) for q, x in r[s].items() if q in ("pop_pivot_s", "tables_pivot_s", "pop_wave_s")} for s in r if s.startswith("spec")})
PY
done
run() { local name=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $OUT/${TAG}_${name}.json 2> $OUT/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/${TAG}_${name}.json") if l.startswith("{")][-1])
    print("$name", round(d["value"]/1e6,3), "Mpx/s", d["ms_per_step"], "ms lossless", d["lossless"])
except Exception as e:
    print("$name FAILED", e)
PY
}
unset BITSWAP_HIP_LIB
run old_a BITSWAP_HIP_LIB=$OLD
run new_a X=1
run old_b BITSWAP_HIP_LIB=$OLD
run new_b X=1
