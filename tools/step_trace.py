#!/usr/bin/env python3
"""Few-chain block step under the microscope: from a `rocprofv3 --kernel-trace` directory, the kernels of the LAST `--ms`
milliseconds before the final gap-free stretch ends (the timed steps of `bench.py --no-extra --no-roofline`), as
(start us, duration us, gap to the previous kernel on the same queue us, queue, kernel) rows plus per-kernel totals.
    python tools/step_trace.py /tmp/trace_dir out.txt [--ms 30]
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    ms = float(sys.argv[sys.argv.index("--ms") + 1]) if "--ms" in sys.argv else 30.0
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            n = n.split("(")[0][-60:]
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n))
    ev.sort()
    # the window ends with the last coding kernel of the run (the receiver's last push: verification copies follow it)
    hi = max(e[1] for e in ev if "k_rans_" in e[3] or "k_layer" in e[3])
    lo = hi - int(ms * 1e6)
    win = [e for e in ev if e[0] >= lo and e[1] <= hi]
    t0 = win[0][0]
    lastq = {}
    per = defaultdict(lambda: [0, 0.0, 0.0])
    busy = []
    with open(out, "w") as fo:
        fo.write(f"# last {ms} ms of the trace: {len(win)} kernels\n# start_us dur_us gap_same_queue_us queue kernel\n")
        for a, b, q, n in win:
            gap = (a - lastq[q]) * 1e-3 if q in lastq else 0.0
            lastq[q] = b
            p = per[n]
            p[0] += 1
            p[1] += (b - a) * 1e-3
            p[2] += max(gap, 0.0)
            fo.write(f"{(a - t0) * 1e-3:10.1f} {(b - a) * 1e-3:8.1f} {gap:8.1f} {q:>3} {n}\n")
        fo.write("# ---- per kernel: calls, total us, mean us, mean gap before it on its queue us\n")
        for n, (c, t, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"# {c:6d} {t:10.1f} {t / c:8.1f} {g / c:8.1f}  {n}\n")
        # union of busy intervals
        iv = sorted((a, b) for a, b, _, _ in win)
        tot, cs, ce = 0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > ce:
                tot += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        tot += ce - cs
        fo.write(f"# window {(hi - t0) * 1e-3:.1f} us, some kernel running {tot * 1e-3:.1f} us, sum of kernel time {sum(p[1] for p in per.values()):.1f} us\n")
    print(open(out).read()[-3000:])


if __name__ == "__main__":
    main()
