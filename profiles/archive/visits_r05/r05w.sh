#!/bin/bash
# round 5, visit w: record-and-replay leg with POISONED memory: after every run each float32 result of a stack kernel is overwritten
# with NaN before its memory returns to the allocator.  Does the failing k_wino_fused call show the poison (it read memory its
# producer had not visibly written) or a finite wrong value?  Second series: the same + a 2 GB read-modify-write sweep after the
# poisoning (no poison line survives in any L2).
TAG=${1:-r05w}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
line = [l for l in txt.splitlines() if l.startswith("record ")]
if not line:
    print("NO RESULT", txt[-800:]); sys.exit(0)
d = json.loads(line[-1][7:])
print({k: d.get(k) for k in ("runs", "lossless", "poison", "sweep_mb", "baseline_replay_mismatches", "error")})
for f in d.get("failures", []):
    for c in f["first_calls_that_do_not_repeat"][:1]:
        o = list(c["outputs"].values())[0]
        print(" run", f["run"], "chains", f["bad_chains"], c["kernel"], "call", c["call"], "aux" if c["aux_stream"] else "main", "before", c["before"],
              "differing", o["differing"], "t", o["dim0"]["values"][:6], "...", "ch", o["dim1"]["values"], "chains", o.get("chains"),
              "nan", o.get("run_values_nan"), "zero", o.get("run_values_zero"), "first", [(e["run"], e["alone"]) for e in o["first"][:3]])
PY
}
REPRO_POISON=nan REPRO_RECORD_REPS=${REPS:-200} REPRO_MAX_FAIL=4 timeout 500 python tools/bf16x3_repro.py --record > $OUT/${TAG}_record_poison.txt 2>&1
echo "poison exit $?"; summ $OUT/${TAG}_record_poison.txt
REPRO_POISON=nan REPRO_SWEEP_MB=2048 REPRO_RECORD_REPS=${REPS:-200} REPRO_MAX_FAIL=4 timeout 500 python tools/bf16x3_repro.py --record > $OUT/${TAG}_record_poison_sweep.txt 2>&1
echo "poison+sweep exit $?"; summ $OUT/${TAG}_record_poison_sweep.txt
