#!/usr/bin/env python3
"""profiles/valu_busy.json from the SQ-counter passes of tools/pmc_valu.sh (three separate --pmc runs of tools/microbench.py,
per-dispatch means per kernel):  VALU busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
Keys name the kernel FLAVOUR (pivot / decode / encode x CDF spec), which is what bench.py looks up for the kernel
it actually timed.     python tools/valu_busy.py gpurun_out/r04Z profiles/valu_busy.json"""
import glob
import json
import re
import sys

MODES = {"0": "encode", "1": "linear", "2": "linear_vec", "3": "decode", "4": "pivot"}


def flavour(name):
    m = re.search(r"k_logistic<(\d+), (\w+), (\d), (\d)>", name)      # <NPL, param type, mode, CDF spec> since round 5
    if m:
        return f"k_logistic<{m.group(1)},{m.group(2)},{MODES.get(m.group(3), m.group(3))},spec{m.group(4)}>"
    m = re.search(r"k_layer64<(\d+), (\w+), (\d), (true|false)>", name)
    if m:
        return f"k_layer64<{m.group(1)},{m.group(2)},spec{m.group(3)},{'push' if m.group(4) == 'true' else 'pop'}>"
    return None


def main():
    prefix, out = sys.argv[1], sys.argv[2]
    acc = {}
    for f in glob.glob(prefix + "_pmc_*.json"):
        if "FETCH" in f or "WRITE" in f:
            continue
        for k, d in json.load(open(f)).items():
            fl = flavour(k)
            if fl:
                acc.setdefault(fl, {}).update(d)
    kernels = {}
    for fl, d in acc.items():
        if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            simd_cycles = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            kernels[fl] = {"valu_busy": round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 4),
                           "valu_insts_per_wave": round(d.get("SQ_INSTS_VALU", 0) / max(d.get("SQ_WAVES", 1), 1), 1),
                           "gui_active_cycles_per_xcd": round(d["GRBM_GUI_ACTIVE"] / 8.0)}
    json.dump({"source": f"{prefix}_pmc_*.json (tools/pmc_valu.sh: three separate --pmc passes over tools/microbench.py --B 400, per-dispatch means)",
               "how": "VALU busy = SQ_ACTIVE_INST_VALU x 4 (the SQ counters tick in quad-cycles) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
               "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
