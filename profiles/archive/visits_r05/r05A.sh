#!/bin/bash
# round 5, visit A: does the forked bf16x3 codec still fail once net_epilogue.hip is built WITHOUT the SLP vectorizer (no v_pk_add_f32
# in k_wino_fused / k_conv3_wino)?  Same box: (1) control, the library with the packed operations (BITSWAP_HIP_LIB=..._slp.so), 400
# runs; (2) the new library, 1500 runs; (3) control again, 400 runs.  Then (4) the GPU suite on the new library and (5) the headline
# step, slp / new / slp / new.
TAG=${1:-r05A}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SLP=$PWD/bitswap_amd/csrc/libbitswap_hip_slp.so
summ() { python - "$1" "$2" <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
line = [l for l in txt.splitlines() if l.startswith("record ")]
if not line:
    print(sys.argv[2], "NO RESULT", txt[-600:]); sys.exit(0)
d = json.loads(line[-1][7:])
print(sys.argv[2], {k: d.get(k) for k in ("runs", "lossless", "baseline_replay_mismatches", "error")},
      [(f["run"], f["bad_chains"], f["first_calls_that_do_not_repeat"][0]["kernel"] if f["first_calls_that_do_not_repeat"] else None,
        (f["first_calls_that_do_not_repeat"][0].get("autopsy") or {}).get("tile_elements") if f["first_calls_that_do_not_repeat"] else None) for f in d.get("failures", [])][:12])
PY
}
BITSWAP_HIP_LIB=$SLP REPRO_RECORD_REPS=400 REPRO_MAX_FAIL=99 timeout 300 python tools/bf16x3_repro.py --record > $OUT/${TAG}_record_slp_a.txt 2>&1; summ $OUT/${TAG}_record_slp_a.txt "packed(control a)"
REPRO_RECORD_REPS=1500 REPRO_MAX_FAIL=99 timeout 600 python tools/bf16x3_repro.py --record > $OUT/${TAG}_record_noslp.txt 2>&1; summ $OUT/${TAG}_record_noslp.txt "no-slp(new)"
BITSWAP_HIP_LIB=$SLP REPRO_RECORD_REPS=400 REPRO_MAX_FAIL=99 timeout 300 python tools/bf16x3_repro.py --record > $OUT/${TAG}_record_slp_b.txt 2>&1; summ $OUT/${TAG}_record_slp_b.txt "packed(control b)"
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -2 $OUT/${TAG}_pytest.log
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
run slp_a BITSWAP_HIP_LIB=$SLP
run new_a X=1
run slp_b BITSWAP_HIP_LIB=$SLP
run new_b X=1
