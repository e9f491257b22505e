#!/usr/bin/env python3
"""Emit the explicit (zero-skipping) C++ formulas of a Cook-Toom transform for csrc/net_epilogue.hip.
    python tools/gen_wino.py 4 5        # F(4,5): B^T (8x8) and A^T (4x8) on the points of winograd.POINTS8
The compiler may not drop `0.0f * x` (IEEE semantics, no fast-math), so the formulas are written out."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import winograd  # noqa: E402


def lit(v):
    from fractions import Fraction
    f = Fraction(v).limit_denominator(1 << 12)
    assert abs(float(f) - v) < 1e-12, v
    return f"{float(f)!r}f"


def emit(name, Mx, src, dst):
    print(f"// {name}: {Mx.shape[0]} x {Mx.shape[1]}")
    for i, row in enumerate(Mx):
        terms = []
        for k, v in enumerate(row):
            if abs(v) < 1e-12:
                continue
            sign = "-" if v < 0 else "+"
            a = abs(v)
            t = f"{src}[{k}]" if abs(a - 1.0) < 1e-12 else f"{lit(a)} * {src}[{k}]"
            terms.append((sign, t))
        s = ""
        for j, (sign, t) in enumerate(terms):
            s += (("-" if sign == "-" else "") + t) if j == 0 else f" {sign} {t}"
        print(f"    {dst}[{i}] = {s};")


if __name__ == "__main__":
    m, r = int(sys.argv[1]), int(sys.argv[2])
    pts = winograd.POINTS if m + r - 1 == 6 else winograd.POINTS8
    AT, G, BT = winograd.cook_toom(m, r, pts)
    np.set_printoptions(linewidth=200, precision=8, suppress=True)
    print("/* points", pts, "*/")
    emit("B^T", BT, "d", "o")
    emit("A^T", AT, "m", "y")
    print("/* G =\n", G, "*/")
