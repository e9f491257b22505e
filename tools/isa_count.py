#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (which kernels are issue-bound and by what).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/x.s bitswap_amd/csrc/bitswap_hip.hip
    python tools/isa_count.py /tmp/x.s k_logisticILi16EfLi3ELb1E [more substrings of mangled names ...]
"""
import sys
from collections import Counter


def body(lines, name):
    for i, l in enumerate(lines):
        if l.startswith('_ZN') and name in l and l.split(';')[0].rstrip().endswith(':'):
            out = []
            for t in lines[i + 1:]:
                t = t.strip()
                if t.startswith('s_endpgm'):
                    break
                out.append(t)
            return out
    return None


def main():
    lines = open(sys.argv[1]).read().split('\n')
    for name in sys.argv[2:]:
        b = body(lines, name)
        if b is None:
            print(name, "not found")
            continue
        ins = [l for l in b if l and not l.startswith(('.', ';', '//')) and not l.split(';')[0].rstrip().endswith(':')]
        c = Counter(l.split()[0] for l in ins)
        pick = lambda f: sum(v for k, v in c.items() if f(k))
        print(f"{name}: total {len(ins)}  valu {pick(lambda k: k.startswith('v_'))}  f64 {pick(lambda k: 'f64' in k)}  "
              f"rcp_f64 {pick(lambda k: k.startswith('v_rcp_f64'))}  readlane {pick(lambda k: k.startswith('v_readlane'))}  "
              f"salu {pick(lambda k: k.startswith('s_'))}  ds {pick(lambda k: k.startswith('ds_'))}  "
              f"vmem {pick(lambda k: k.startswith(('global_', 'buffer_', 'flat_')))}")


if __name__ == "__main__":
    main()
