#!/bin/bash
# round 5, visit c: (1) bf16x3 co-residency: round 4's failing scenario on today's kernels, and the defect re-introduced on purpose
# (stray LDS-DMA at workgroup exit); (2) the default bench line; (3) TIMED-REGION-ONLY profile: kernel trace + PMC FETCH / WRITE passes
# of the bench with the k_where sentinels on either side of its timed region; (4) SQ counters of the table kernels (valu_busy)
TAG=${1:-r05c}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python tools/bf16x3_repro.py --reps 25 --codec > $OUT/${TAG}_bf16x3_repro.txt 2>&1; grep -v "^ \|^{\|^}" $OUT/${TAG}_bf16x3_repro.txt | tail -30
timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?"; tail -2 $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/${TAG}_bench.json") if l.startswith("{")][-1])
print(json.dumps(d["summary"], indent=0)[:3000])
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","kind")})
PY
BCMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
export BITSWAP_BENCH_SENTINEL=1
( cd /tmp && rm -rf prof_stats prof_fetch prof_write
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st --output-format csv -- $BCMD > $OUT/${TAG}_prof_stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -o pf --output-format csv -- $BCMD > $OUT/${TAG}_prof_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -o pw --output-format csv -- $BCMD > $OUT/${TAG}_prof_write.log 2>&1 )
unset BITSWAP_BENCH_SENTINEL
python tools/prof_summary.py stats /tmp/prof_stats $OUT/${TAG}_kernel_stats_timed.txt timed > /dev/null
python tools/prof_summary.py stats /tmp/prof_stats $OUT/${TAG}_kernel_stats_whole_process.txt > /dev/null
python tools/prof_summary.py pmc /tmp/prof_fetch FETCH_SIZE $OUT/${TAG}_pmc_FETCH_SIZE_timed.json timed > /dev/null
python tools/prof_summary.py pmc /tmp/prof_write WRITE_SIZE $OUT/${TAG}_pmc_WRITE_SIZE_timed.json timed > /dev/null
head -22 $OUT/${TAG}_kernel_stats_timed.txt
rm -f $OUT/${TAG}_traffic.json
python tools/prof_summary.py traffic $OUT/${TAG}_pmc_FETCH_SIZE_timed.json $OUT/${TAG}_pmc_WRITE_SIZE_timed.json cifar8 $OUT/${TAG}_traffic.json 1024000
bash tools/pmc_valu.sh ${TAG} > $OUT/${TAG}_pmc_valu.log 2>&1
python tools/valu_busy.py $OUT/${TAG} $OUT/${TAG}_valu_busy.json; cat $OUT/${TAG}_valu_busy.json | head -60
