#!/usr/bin/env python3
"""Condense rocprofv3 output directories into small text/JSON summaries for profiles/.

    python tools/prof_summary.py stats <dir> <out.txt>       # --kernel-trace --stats run
    python tools/prof_summary.py pmc <dir> <counter> <out.json>   # --pmc <counter> run
    python tools/prof_summary.py traffic <fetch.json> <write.json> <workload> <traffic.json> <rows per launch>
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


def stats(d, out):
    rows = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for f in find(d, "kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            a = rows[n]
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
    tot = sum(a[1] for a in rows.values()) or 1.0
    with open(out, "w") as fo:
        fo.write(f"# rocprofv3 --kernel-trace --stats summary of {d}\n")
        fo.write(f"# {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel\n")
        for n, a in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"  {a[0]:7d} {a[1]:12.1f} {a[1] / a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100 * a[1] / tot:6.2f}  {n}\n")
    print(open(out).read()[:3000])


def pmc(d, counter, out):
    acc = defaultdict(lambda: [0, 0.0])
    for f in find(d, "counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    res = {n: {"dispatches": a[0], f"{counter}_sum": a[1], f"{counter}_per_dispatch": a[1] / a[0]} for n, a in acc.items()}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1][f"{counter}_sum"])[:25])
    json.dump(res, open(out, "w"), indent=1)
    for n, v in list(res.items())[:8]:
        print(n[:90], v)


def traffic(fetch_json, write_json, workload, out, rows_per_launch):
    """HBM bytes per launch of the roofline kernel (fused logistic -> integer table, decode side, K = 1024), corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: counters are in KiB, FETCH_SIZE under-reports wide
    coalesced reads by exactly 2x (doubled here), WRITE_SIZE taken as is.  Both hand-off flavours when present: whole rows
    (k_logistic<16, float, 3, ...>: BS_LAYOUT_WAVE) and 64 cumulative values per row (<16, float, 4, ...>: BS_LAYOUT_PIVOT),
    and the pop kernels that read them back."""
    def per(path, counter, prefix):
        d = json.load(open(path))
        for k, v in d.items():
            if k.startswith(prefix):
                return v[f"{counter}_per_dispatch"] * 1024.0, v["dispatches"]
        return None, 0
    res = json.load(open(out)) if os.path.exists(out) else {}
    rows = float(rows_per_launch)
    entry = res.get(workload, {})
    entry.update({"rows_per_launch": int(rows), "fetch_correction": 2.0, "source": [os.path.basename(fetch_json), os.path.basename(write_json)]})
    for tag, prefix in (("decode", "void k_logistic<16, float, 3, true"), ("pivot", "void k_logistic<16, float, 4, true"),
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
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
