OUT=$PWD/gpurun_out; export TMPDIR=/tmp
for fmt in reference wave64; do
  timeout 300 python bench.py --chains 13 --groups 1 --steps 12 --warmup 2 --no-extra --no-cpu-baseline --format $fmt > $OUT/r02e_bench13_$fmt.json 2> $OUT/r02e_bench13_$fmt.err; echo "13-chain $fmt exit $?"; cut -c1-200 $OUT/r02e_bench13_$fmt.json
  timeout 300 python bench.py --chains 100 --groups 1 --steps 6 --warmup 1 --no-extra --no-cpu-baseline --format $fmt > $OUT/r02e_bench100_$fmt.json 2> /dev/null; echo "100-chain $fmt exit $?"; cut -c1-200 $OUT/r02e_bench100_$fmt.json
done
timeout 300 python bench.py --workload imagenet4 --steps 6 --warmup 1 --no-extra --no-cpu-baseline --format wave64 > $OUT/r02e_bench_imagenet4_wave64.json 2>/dev/null; cut -c1-200 $OUT/r02e_bench_imagenet4_wave64.json
BCMD="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --format wave64"
( cd /tmp && rm -rf p64s p64f p64w
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p64s -o st --output-format csv -- $BCMD > $OUT/r02e_w64_prof_stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/p64f -o pf --output-format csv -- $BCMD > $OUT/r02e_w64_prof_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/p64w -o pw --output-format csv -- $BCMD > $OUT/r02e_w64_prof_write.log 2>&1 )
python tools/prof_summary.py stats /tmp/p64s $OUT/r02e_wave64_kernel_stats.txt > /dev/null
python tools/prof_summary.py pmc /tmp/p64f FETCH_SIZE $OUT/r02e_wave64_pmc_FETCH_SIZE.json > /dev/null
python tools/prof_summary.py pmc /tmp/p64w WRITE_SIZE $OUT/r02e_wave64_pmc_WRITE_SIZE.json > /dev/null
head -12 $OUT/r02e_wave64_kernel_stats.txt | cut -c1-150
BENCH_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --chains 100 --groups 2 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/r02e_bench_2rank_gloo.json 2> $OUT/r02e_bench_2rank_gloo.err; echo "2-rank exit $?"; cut -c1-300 $OUT/r02e_bench_2rank_gloo.json; tail -3 $OUT/r02e_bench_2rank_gloo.err
