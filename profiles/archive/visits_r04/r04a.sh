#!/bin/bash
# round 4 visit A: forked block step -- parity tests, A/B at 13 / 100 chains, kernel trace of a 13-chain step
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "forked or graph_equals or grouped or strong or launch_size or full_width_oracle or ragged" > $OUT/r04a_pytest.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/r04a_pytest.log
B="python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 12 --warmup 3"
run() { tag=$1; shift; env "$@" > /dev/null 2>&1; }
one() { # tag env... -- args
  tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 $B "$@" 2> $OUT/r04a_$tag.err | tail -1 > $OUT/r04a_$tag.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/r04a_$tag.json")); print("$tag", round(d["value"]/1e6, 3), "Mpx/s", d["ms_per_step"], "ms/step lossless", d["lossless"], "forked", d["config"].get("forked_block_step"), "groups", d["config"]["chain_groups"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/r04a_$tag.err").read()[-1500:])
PY
}
one c13_fork0 BITSWAP_FORK=0 -- --chains 13 --groups 1
one c13_fork1 BITSWAP_FORK=auto -- --chains 13 --groups 1
one c100_g2_fork0 BITSWAP_FORK=0 -- --chains 100 --groups 2
one c100_g2_fork1 BITSWAP_FORK=auto -- --chains 100 --groups 2
one c100_g1_fork1 BITSWAP_FORK=auto -- --chains 100 --groups 1
one c100_g1_fork0 BITSWAP_FORK=0 -- --chains 100 --groups 1
one i13_bbans_fork0 BITSWAP_FORK=0 -- --chains 13 --groups 1 --workload imagenet4 --bitswap 0
one i13_bbans_fork1 BITSWAP_FORK=auto -- --chains 13 --groups 1 --workload imagenet4 --bitswap 0
one c1_fork1 BITSWAP_FORK=auto -- --chains 1 --groups 1
one c13_w64_fork1 BITSWAP_FORK=auto -- --chains 13 --groups 1 --format wave64
one c13_w64_fork0 BITSWAP_FORK=0 -- --chains 13 --groups 1 --format wave64
( cd /tmp && rm -rf tr13 && timeout 300 rocprofv3 --kernel-trace -d /tmp/tr13 -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 6 --warmup 2 --chains 13 --groups 1 > $OUT/r04a_trace13.log 2>&1 )
python tools/step_trace.py /tmp/tr13 $OUT/r04a_trace13.txt --ms 25 | tail -40
( cd /tmp && rm -rf tr100 && timeout 300 rocprofv3 --kernel-trace -d /tmp/tr100 -o t --output-format csv -- python $R/bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 6 --warmup 2 --chains 100 --groups 1 > $OUT/r04a_trace100.log 2>&1 )
python tools/step_trace.py /tmp/tr100 $OUT/r04a_trace100.txt --ms 50 | tail -40
