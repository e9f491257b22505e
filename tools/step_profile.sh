#!/bin/bash
# Per-kernel time of ONE bench step: kernel-trace totals of a 5-step run minus those of a 1-step run, divided by 4
# (setup, bins sampling and warm-up cancel).   gpurun --timeout 900 -- '[BENCH_ARGS="--chains 100 --groups 1"] bash tools/step_profile.sh r02t'
TAG=${1:-rXX}; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for k in 1 5; do
  ( cd /tmp && rm -rf sp$k && timeout 400 rocprofv3 --kernel-trace -d /tmp/sp$k -o st --output-format csv -- python $R/bench.py --steps $k --warmup 1 --no-cpu-baseline --no-extra $BENCH_ARGS > /dev/null 2>&1 )
done
python - <<PY
import csv, glob, collections
def tot(d):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]
            a = acc[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc
a, b = tot("/tmp/sp1"), tot("/tmp/sp5")
rows = []
for n in b:
    dc, dt = b[n][0] - a.get(n, [0, 0])[0], b[n][1] - a.get(n, [0, 0.0])[1]
    if dc > 0:
        rows.append((dt / 4 / 1e3, dc / 4, n))
rows.sort(reverse=True)
with open("$OUT/${TAG}_step_kernels.txt", "w") as fo:
    fo.write("# ms per bench step (sender + receiver of one block for 800 chains), launches per step, kernel -- (5-step run minus 1-step run) / 4\\n")
    fo.write(f"# total {sum(r[0] for r in rows):.1f} ms of kernel time per step (streams overlap: wall time per step is lower)\\n")
    for ms, cnt, n in rows:
        fo.write(f"{ms:9.3f} {cnt:8.1f}  {n}\\n")
print(open("$OUT/${TAG}_step_kernels.txt").read()[:3500])
PY
