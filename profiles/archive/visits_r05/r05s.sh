#!/bin/bash
# round 5, visit s: final validation on the round's code -- GPU suite, smoke, default bench line (with the CU-mask extra), and fresh
# kernel traces of the forked few-chain step (13 and 100 chains) with the spec 3 kernels
TAG=${1:-r05s}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/${TAG}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads([l for l in open("$OUT/${TAG}_bench.json") if l.startswith("{")][-1])
print(json.dumps(d["summary"])[:3000])
PY
tr() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( cd /tmp && rm -rf tr_$tag && env "${envs[@]}" timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 6 --warmup 2 "$@" > $OUT/${TAG}_$tag.log 2>&1 )
  grep -h '^{' $OUT/${TAG}_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['lossless'])"
  python tools/step_trace.py /tmp/tr_$tag $OUT/${TAG}_trace_$tag.txt --ms $MS | tail -16
}
MS=11 tr c13 X=1 -- --chains 13 --groups 1
MS=24 tr c100 X=1 -- --chains 100 --groups 1
