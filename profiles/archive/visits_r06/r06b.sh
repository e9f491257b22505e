#!/bin/bash
# round 6, visit b: (1) what the bf16 matrix pipe sustains under the power budget (tools/probes/mfma_rate.hip);
# (2) VERDICT r5 #6: conv stacks over cache-resident column chunks (bench --nn-batch) with FETCH / WRITE counters;
# (3) VERDICT r5 #2b: the bf16x3 GEMM with and without its whole-register-file claim IN the 1000-chain pipeline and at 100 chains
TAG=${1:-r06b}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_rate tools/probes/mfma_rate.hip && timeout 120 /tmp/mfma_rate > $OUT/${TAG}_mfma_rate.txt 2>&1; cat $OUT/${TAG}_mfma_rate.txt
B="python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-roofline --full-record /dev/null"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"   {d['ms_per_step']:8.2f} ms/step  {d['value']/1e6:6.3f} Mpixel/s  lossless={d['lossless']}  crc={(d.get('stream_gather') or {}).get('crc32_of_streams_in_chain_order')}")
except Exception as e:
    print("   failed:", e, open(sys.argv[1]).read()[-300:])
PY
}
{
echo "== conv stacks over column chunks (bench --nn-batch N: chains per chunk), cifar8 1000 chains / 2 groups, ms per step"
for nb in 0 250 125 100 63 50 0; do
  echo "nn_batch $nb"; timeout 400 $B --nn-batch $nb > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
echo "== bf16x3 GEMM: whole-register-file claim (default) vs unclaimed (BITSWAP_BF16X3_DIAG=noclaim), same box, alternating"
for rep in 1 2; do
  echo "fp32 1000";            timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "bf16x3 claimed 1000";  BITSWAP_GEMM_ARITH=bf16x3 timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "bf16x3 unclaimed 1000"; BITSWAP_GEMM_ARITH=bf16x3 BITSWAP_BF16X3_DIAG=noclaim timeout 400 $B > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
for rep in 1 2; do
  echo "fp32 100";             timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "bf16x3 claimed 100";   BITSWAP_GEMM_ARITH=bf16x3 timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
  echo "bf16x3 unclaimed 100"; BITSWAP_GEMM_ARITH=bf16x3 BITSWAP_BF16X3_DIAG=noclaim timeout 400 $B --scaling strong --total-chains 100 > /tmp/o.json 2>/tmp/o.err; line /tmp/o.json
done
} > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
# counters of the chunked run against the default (timed region only)
for nb in 0 63; do
  BCMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-roofline --full-record /dev/null --nn-batch $nb"
  export BITSWAP_BENCH_SENTINEL=1
  ( cd /tmp && rm -rf pf pw ps
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ps -o st --output-format csv -- $BCMD > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- $BCMD > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- $BCMD > /dev/null 2>&1 )
  unset BITSWAP_BENCH_SENTINEL
  python tools/prof_summary.py stats /tmp/ps $OUT/${TAG}_kernel_stats_nb$nb.txt timed > /dev/null
  python tools/prof_summary.py pmc /tmp/pf FETCH_SIZE $OUT/${TAG}_pmc_FETCH_SIZE_nb$nb.json timed > /dev/null
  python tools/prof_summary.py pmc /tmp/pw WRITE_SIZE $OUT/${TAG}_pmc_WRITE_SIZE_nb$nb.json timed > /dev/null
  echo "== nn_batch $nb"; head -12 $OUT/${TAG}_kernel_stats_nb$nb.txt | cut -c1-150
done
