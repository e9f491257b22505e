#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small text/JSON summaries for profiles/.

    python tools/prof_summary.py stats <dir> <out.txt> [timed]   # --kernel-trace --stats run; `timed`: only the dispatches between
                                                                 # the first and the last k_where sentinel (bench.py launches one on
                                                                 # either side of its timed region when BITSWAP_BENCH_SENTINEL=1)
    python tools/prof_summary.py pmc <dir> <counter> <out.json>   # --pmc <counter> run
    python tools/prof_summary.py traffic <fetch.json> <write.json> <workload> <traffic.json> <rows per launch>
    python tools/prof_summary.py timeline <dir> <out.json>   # --kernel-trace run: who occupies the wall time of the steps
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."


def stats(d, out, timed=False):
    rows = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    note = ""
    for f in find(d, "kernel_trace.csv"):
        recs = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        if timed:
            marks = [i for i, r in enumerate(recs) if "k_where" in r["Kernel_Name"]]
            if len(marks) < 2:
                continue                                  # a process without a timed region (or without the sentinels)
            wall = (int(recs[marks[-1]]["Start_Timestamp"]) - int(recs[marks[0]]["End_Timestamp"])) * 1e-6
            note += f"# timed region of {os.path.basename(f)}: {marks[-1] - marks[0] - 1} dispatches in {wall:.2f} ms between the k_where sentinels\n"
            recs = [r for r in recs[marks[0] + 1: marks[-1]] if "k_where" not in r["Kernel_Name"]]
        for r in recs:
            n = short(r["Kernel_Name"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            a = rows[n]
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
    tot = sum(a[1] for a in rows.values()) or 1.0
    with open(out, "w") as fo:
        fo.write(f"# rocprofv3 --kernel-trace --stats summary of {d}" + (" -- TIMED REGION ONLY" if timed else "") + "\n" + note)
        fo.write(f"# {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel\n")
        for n, a in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"  {a[0]:7d} {a[1]:12.1f} {a[1] / a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100 * a[1] / tot:6.2f}  {n}\n")
    print(open(out).read()[:3000])


def pmc(d, counter, out, timed=False):
    acc = defaultdict(lambda: [0, 0.0])
    for f in find(d, "counter_collection.csv"):
        recs = list(csv.DictReader(open(f)))
        if timed:       # dispatch order: keep what lies between the first and the last k_where sentinel
            key = "Dispatch_Id" if recs and "Dispatch_Id" in recs[0] else None
            if key:
                recs.sort(key=lambda r: int(r[key]))
            marks = [i for i, r in enumerate(recs) if "k_where" in r["Kernel_Name"]]
            if len(marks) < 2:
                continue
            recs = recs[marks[0] + 1: marks[-1]]
        for r in recs:
            if r.get("Counter_Name") != counter or "k_where" in r["Kernel_Name"]:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    res = {n: {"dispatches": a[0], f"{counter}_sum": a[1], f"{counter}_per_dispatch": a[1] / a[0]} for n, a in acc.items()}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1][f"{counter}_sum"])[:25])
    json.dump(res, open(out, "w"), indent=1)
    for n, v in list(res.items())[:8]:
        print(n[:90], v)


SERIAL = ("k_rans_pop", "k_rans_push", "k_layer64")


def timeline(d, out, tail_ms=700.0, skip_ms=100.0):
    """Sweep over the kernel intervals of the window [end - skip_ms - tail_ms, end - skip_ms] of the trace (inside the timed
    steps when the bench ran with --no-roofline --no-extra; the front of a trace is set-up and warm-up): how much wall time has a bulk kernel (tables, GEMM, transforms) running, how much only a serial coder kernel
    (one wavefront per chain: the chip is nearly empty), how much nothing at all, and how much two bulk kernels at once
    (the two chain groups' streams; such kernels are time-sliced)."""
    ev = []
    for f in find(d, "kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")))
    ev.sort()
    # the batched GEMM has one name for the 36-position (3x3) and the 64-position (5x5) products: tell them apart by the
    # transform kernel in front of them on the same queue
    prev, tagged = {}, []
    for a, b, n, q in ev:
        if "k_wino_gemm<4" in n:
            p = prev.get(q, "")
            n = n[:24] + (" [T=64]" if "<8, 8>" in p or "<6, 8>" in p or "<0, 8>" in p else " [T=36]" if "wino" in p else " [after ?]")
        prev[q] = n
        tagged.append((a, b, n))
    ev = tagged
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    if tail_ms <= 0:          # auto: the longest stretch of the trace without a gap of more than 5 ms (the step loop), middle 80 %
        segs, cur, end = [], [ev[0]], ev[0][1]
        for e in ev[1:]:
            if e[0] - end > 5e6:
                segs.append(cur)
                cur = []
            cur.append(e)
            end = max(end, e[1])
        segs.append(cur)
        seg = max(segs, key=lambda g: max(x[1] for x in g) - g[0][0])
        a0, b0 = seg[0][0], max(x[1] for x in seg)
        lo, hi = a0 + (b0 - a0) // 10, b0 - (b0 - a0) // 10
        tail_ms, skip_ms = (hi - lo) * 1e-6, (t1 - hi) * 1e-6
    else:
        hi = t1 - int(skip_ms * 1e6)
        lo = hi - int(tail_ms * 1e6)
    ev = [(max(a, lo), min(b, hi), n) for a, b, n in ev if b > lo and a < hi]
    pts = []
    for a, b, n in ev:
        ser = any(k in n for k in SERIAL)
        pts.append((a, 1, ser)), pts.append((b, -1, ser))
    pts.sort()
    nb = ns = 0
    last = pts[0][0]
    acc = defaultdict(float)
    for t, dlt, ser in pts:
        key = ("idle" if nb == 0 and ns == 0 else "serial_only" if nb == 0 else "bulk_1" + ("+serial" if ns else "")
               if nb == 1 else "bulk_2+" + ("+serial" if ns else ""))
        acc[key] += (t - last) * 1e-6
        last = t
        if ser:
            ns += dlt
        else:
            nb += dlt
    span = (pts[-1][0] - pts[0][0]) * 1e-6
    # per call of a bulk kernel: the part of its interval it had no other bulk kernel next to it, and the part a serial
    # kernel ran beside it -> mean duration of the calls that ran (a) alone, (b) beside a serial kernel only
    import bisect
    times = [p[0] for p in pts]
    cb, cs, nb2, ns2 = [0.0], [0.0], 0, 0          # cumulative time with exactly one bulk kernel / with a serial kernel
    for i, (t, dlt, ser) in enumerate(pts):
        if i:
            dt = t - pts[i - 1][0]
            cb.append(cb[-1] + (dt if nb2 == 1 else 0)), cs.append(cs[-1] + (dt if ns2 > 0 else 0))
        if ser:
            ns2 += dlt
        else:
            nb2 += dlt
    def cum(c, t):
        i = bisect.bisect_right(times, t) - 1
        return c[i]               # (events of equal time share a value: intervals start and end ON event times)
    solo = defaultdict(lambda: {"alone": [0, 0.0], "beside_serial": [0, 0.0], "shared": [0, 0.0]})
    for a, b, n in ev:
        if any(k in n for k in SERIAL) or b <= a:
            continue
        alone = (cum(cb, b) - cum(cb, a)) / (b - a)
        sfrac = (cum(cs, b) - cum(cs, a)) / (b - a)
        k = "shared" if alone < 0.95 else "beside_serial" if sfrac > 0.9 else "alone" if sfrac < 0.1 else None
        if k:
            solo[n][k][0] += 1
            solo[n][k][1] += (b - a) * 1e-3
    per = defaultdict(lambda: [0, 0.0])
    for a, b, n in ev:
        per[n][0] += 1
        per[n][1] += (b - a) * 1e-6
    gaps = sorted(((ev[i + 1][0] - max(e[1] for e in ev[max(0, i - 8):i + 1])) * 1e-3, ev[i][2][:50], ev[i + 1][2][:50])
                  for i in range(len(ev) - 1))[-8:]
    res = {"span_ms": span, "wall_ms": dict(acc), "wall_frac": {k: v / span for k, v in acc.items()},
           "kernel_ms_sum": {n: {"calls": c, "ms": ms} for n, (c, ms) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]},
           "bulk_call_us": {n: {k: {"calls": c, "mean_us": (t / c if c else None)} for k, (c, t) in v.items()}
                            for n, v in sorted(solo.items(), key=lambda kv: -sum(x[1] for x in kv[1].values()))[:10]},
           "largest_gaps_us": gaps[::-1], "note": "window of %.0f ms ending %.0f ms before the last kernel of the trace" % (tail_ms, skip_ms)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("span_ms", "wall_ms", "wall_frac")}, indent=1))
    for n, v in res["bulk_call_us"].items():
        print(n[:70], {k: (x["calls"], x["mean_us"] and round(x["mean_us"], 1)) for k, x in v.items()})


def traffic(fetch_json, write_json, workload, out, rows_per_launch):
    """HBM bytes per launch of the roofline kernel (fused logistic -> integer table, decode side, K = 1024), corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: counters are in KiB, FETCH_SIZE under-reports wide
    coalesced reads by exactly 2x (doubled here), WRITE_SIZE taken as is.  Both hand-off flavours when present: whole rows
    (k_logistic<16, float, 3, ...>: BS_LAYOUT_WAVE) and 64 cumulative values per row (<16, float, 4, ...>: BS_LAYOUT_PIVOT),
    and the pop kernels that read them back."""
    def per(path, counter, prefix):
        """The flavour with the most dispatches among the kernels whose name starts with `prefix` (the CDF spec is the last
        template argument of k_logistic: whichever spec the run used -- 4 since round 6, 3 in round 5 -- is the one counted)."""
        d = json.load(open(path))
        hits = [(v["dispatches"], v[f"{counter}_per_dispatch"] * 1024.0) for k, v in d.items() if k.startswith(prefix)]
        if not hits:
            return None, 0
        n, val = max(hits)
        return val, n
    res = json.load(open(out)) if os.path.exists(out) else {}
    rows = float(rows_per_launch)
    entry = res.get(workload, {})
    entry.update({"rows_per_launch": int(rows), "fetch_correction": 2.0, "source": [os.path.basename(fetch_json), os.path.basename(write_json)]})
    for tag, prefix in (("decode", "void k_logistic<16, float, 3, "), ("pivot", "void k_logistic<16, float, 4, "),
                        ("pop_wave", "void k_rans_pop_wave<16"), ("pop_pivot", "void k_rans_pop_pivot<16")):
        f, nf = per(fetch_json, "FETCH_SIZE", prefix)
        w, nw = per(write_json, "WRITE_SIZE", prefix)
        if f is None or w is None:
            continue
        name = f"k_logistic_{tag}" if tag in ("decode", "pivot") else f"k_rans_{tag}"
        entry[f"{name}_bytes_per_launch"] = int(2 * f + w)
        entry[f"{name}_bytes_per_row"] = (2 * f + w) / rows
        entry[f"{name}_fetch_bytes_raw"], entry[f"{name}_write_bytes"], entry[f"{name}_dispatches"] = f, w, [nf, nw]
    res[workload] = entry
    json.dump(res, open(out, "w"), indent=1)
    print(res[workload])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], timed=len(sys.argv) > 4 and sys.argv[4] == "timed")
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], *[float(x) for x in sys.argv[4:6]])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], timed=len(sys.argv) > 5 and sys.argv[5] == "timed")
