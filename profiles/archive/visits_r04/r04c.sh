#!/bin/bash
# round 4 visit C: pop kernel diet + three-stage GEMM for small launches: parity, kernel timings, few-chain bench
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "wino_gemm or own_gemm or rans or pop or round_trip or pivot or forked or chain_replay or full_width_oracle" > $OUT/r04c_pytest.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/r04c_pytest.log
for v in 0 default 1000000; do
  if [ $v = default ]; then python tools/gemm_small.py > $OUT/r04c_gemm_$v.json 2>/dev/null; else BITSWAP_GEMM_NS3_UNITS=$v python tools/gemm_small.py > $OUT/r04c_gemm_$v.json 2>/dev/null; fi
done
python - <<PY
import json
d = {v: json.load(open("$OUT/r04c_gemm_%s.json" % v)) for v in ("0", "default", "1000000")}
for k in d["0"]:
    if k == "ns3_units": continue
    same = len({d[v][k]["checksum"] for v in d}) == 1
    print(f"{k:28s} ns2 {d['0'][k]['us']:8.1f} us  default {d['default'][k]['us']:8.1f}  ns3 {d['1000000'][k]['us']:8.1f} us  ({d['1000000'][k]['TFLOPs']} TF)  same bits {same}")
PY
python tools/microbench.py --B 13 > $OUT/r04c_micro13.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/r04c_micro13.json')); print({k:v for k,v in d.items() if 'pop' in k or 'push' in k})"
python tools/microbench.py --B 100 > $OUT/r04c_micro100.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/r04c_micro100.json')); print({k:v for k,v in d.items() if 'pop' in k or 'push' in k})"
B="python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 12 --warmup 3"
one() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 $B "$@" 2> $OUT/r04c_$tag.err | tail -1 > $OUT/r04c_$tag.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/r04c_$tag.json")); print("$tag", round(d["value"]/1e6, 3), "Mpx/s", d["ms_per_step"], "ms/step lossless", d["lossless"], "groups", d["config"]["chain_groups"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/r04c_$tag.err").read()[-1500:])
PY
}
one c13 X=1 -- --chains 13 --groups 1
one c13_ns2 BITSWAP_GEMM_NS3_UNITS=0 -- --chains 13 --groups 1
one c100_g2 X=1 -- --chains 100 --groups 2
one c100_g2_ns2 BITSWAP_GEMM_NS3_UNITS=0 -- --chains 100 --groups 2
one c100_g2_ns3all BITSWAP_GEMM_NS3_UNITS=1000000 -- --chains 100 --groups 2
one c100_g1 X=1 -- --chains 100 --groups 1
one c100_g4 X=1 -- --chains 100 --groups 4
one c1000 X=1 -- --chains 1000 --groups 2 --steps 6 --warmup 2
one c1000_ns3all BITSWAP_GEMM_NS3_UNITS=1000000 -- --chains 1000 --groups 2 --steps 6 --warmup 2
