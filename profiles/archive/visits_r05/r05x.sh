#!/bin/bash
# round 5, visit x: record-and-replay leg with the AUTOPSY of the failing k_wino_fused call: which element of the activated plane
# changed (least squares through the forward transform), which value the kernel must have held for it, and whether that value
# exists anywhere in the run's tensors.
TAG=${1:-r05x}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
REPRO_RECORD_REPS=${REPS:-300} REPRO_MAX_FAIL=5 timeout 700 python tools/bf16x3_repro.py --record > $OUT/${TAG}_record_autopsy.txt 2>&1
echo "exit $?"
python - $OUT/${TAG}_record_autopsy.txt <<'PY'
import json, sys
txt = open(sys.argv[1]).read()
line = [l for l in txt.splitlines() if l.startswith("record ")]
if not line:
    print("NO RESULT", txt[-1500:]); sys.exit(0)
d = json.loads(line[-1][7:])
print({k: d.get(k) for k in ("runs", "lossless", "baseline_replay_mismatches", "error")})
for f in d.get("failures", []):
    c = f["first_calls_that_do_not_repeat"][0]
    print(" run", f["run"], "chains", f["bad_chains"], c["kernel"], "call", c["call"], "aux" if c["aux_stream"] else "main")
    au = c.get("autopsy") or {}; print("   autopsy", {k: au.get(k) for k in ("channel", "chain", "tile_elements", "M_position", "lstsq_residual")}, c.get("autopsy_error"))
    for h in (au.get("lane_search") or [])[:12]:
        print("     hit", h)
PY
