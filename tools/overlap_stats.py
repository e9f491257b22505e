#!/usr/bin/env python3
"""From a `rocprofv3 --kernel-trace` directory: over the last `--ms` milliseconds of coding, how much of the wall time has a bulk
kernel (GEMM, transforms, tables) running, how much only serial coder kernels (k_rans_*), how much nothing -- and the per-kernel
sums.  Says whether the many-chain step is bound by the amount of bulk work or by the serial chain.
    python tools/overlap_stats.py /tmp/trace_dir [--ms 400]
"""
import csv
import glob
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def main():
    d = sys.argv[1]
    ms = float(sys.argv[sys.argv.index("--ms") + 1]) if "--ms" in sys.argv else 400.0
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-48:]
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    hi = max(e[1] for e in ev if "k_rans_" in e[2] or "k_layer" in e[2])
    lo = hi - int(ms * 1e6)
    win = [(max(a, lo), b, n) for a, b, n in ev if b > lo and a < hi]
    serial = [(a, b) for a, b, n in win if "k_rans_" in n]
    bulk = [(a, b) for a, b, n in win if "k_rans_" not in n]
    u_all, u_bulk, u_serial = union(serial + bulk), union(bulk), union(serial)
    per = defaultdict(lambda: [0, 0])
    for a, b, n in win:
        per[n][0] += 1
        per[n][1] += b - a
    print(f"window {ms:.0f} ms: some kernel {u_all / 1e6:.1f} ms, a bulk kernel {u_bulk / 1e6:.1f} ms, a serial kernel {u_serial / 1e6:.1f} ms, "
          f"only serial {(u_all - u_bulk) / 1e6:.1f} ms, nothing {ms - u_all / 1e6:.1f} ms")
    print(f"sum of bulk kernel time {sum(b - a for a, b in bulk) / 1e6:.1f} ms, of serial kernel time {sum(b - a for a, b in serial) / 1e6:.1f} ms")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {c:6d} {t / 1e6:9.2f} ms {t / c / 1e3:9.1f} us  {n}")


if __name__ == "__main__":
    main()
