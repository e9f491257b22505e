#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (which kernels are issue-bound and by what).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/x.s bitswap_amd/csrc/tables.hip   (or pop.hip, push.hip, layer64.hip)
    python tools/isa_count.py /tmp/x.s k_logisticILi16EfLi4ELi3E [more substrings of mangled names ...]
    python tools/isa_count.py --blocks /tmp/x.s k_logisticILi16EfLi4ELi3E     per straight-line run between branches / labels

--blocks: since CDF spec 3 the row loop of k_logistic holds two arms (batch inversion / spec 2's arithmetic for peaked rows) and
a row takes ONE of them, so "largest loop" over-counts: the issue slots of a row are the runs it executes -- for
k_logistic<16, float, pivot, spec 3>: row head 38 + batch arm 161 + integer tail 63 + store 6 = 268 (bench.py VALU_SLOTS_PER_ROW).
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


def mix(ins, tag):
    c = Counter(l.split()[0] for l in ins)
    pick = lambda f: sum(v for k, v in c.items() if f(k))
    valu, rcp = pick(lambda k: k.startswith('v_')), pick(lambda k: k.startswith('v_rcp_f64'))
    print(f"{tag}: total {len(ins)}  valu {valu}  f64 {pick(lambda k: 'f64' in k)}  rcp_f64 {rcp}  "
          f"issue slots (valu + 3 per quarter-rate rcp) {valu + 3 * rcp}  readlane {pick(lambda k: k.startswith('v_readlane'))}  "
          f"salu {pick(lambda k: k.startswith('s_'))}  ds {pick(lambda k: k.startswith('ds_'))}  "
          f"vmem {pick(lambda k: k.startswith(('global_', 'buffer_', 'flat_')))}")


def blocks(lines, name):
    import re
    b = body(lines, name)
    if b is None:
        print(name, "not found")
        return
    marks = sorted(set([0] + [i for i, l in enumerate(b) if re.match(r'^\s*s_c?branch', l) or re.match(r'^\.LBB', l)] + [len(b)]))
    for a, e in zip(marks, marks[1:]):
        ins = [l for l in b[a:e] if l and not l.startswith(('.', ';', '//')) and not l.split(';')[0].rstrip().endswith(':')]
        if len(ins) >= 16:
            mix(ins, f"{name} [lines {a}..{e}]")


def async_load_hazards(lines, name):
    """Inline-asm `global_load_dwordx4` (between ;;#ASMSTART / ;;#ASMEND) land asynchronously in registers the compiler
    believes defined: list every instruction of kernel `name` that reads or writes such a register before the next
    `s_waitcnt ... vmcnt(...)` (the correctness of the hand-placed waits of wino_gemm_bf16x3.hip is a property of the
    register allocation, so it is checked on the built code, tests/test_host_cpu.py).  -> (number of asm loads seen, hazards)."""
    import re
    b = body(lines, name)
    if b is None:
        return None

    def regs(text):
        out = set()
        for a, z in re.findall(r'\bv\[(\d+):(\d+)\]', text):
            out |= set(range(int(a), int(z) + 1))
        for a in re.findall(r'\bv(\d+)\b', text):
            out.add(int(a))
        return out
    pending, hazards, inasm, nloads = {}, [], False, 0
    for i, t in enumerate(b):
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not t or t.startswith((';', '.')) or t.split(';')[0].rstrip().endswith(':'):
            continue
        op = t.split()[0]
        if inasm and op.startswith('global_load_dword'):
            nloads += 1
            for r in regs(t.split()[1].rstrip(',')):
                pending[r] = i
            continue
        if op == 's_waitcnt' and 'vmcnt' in t:
            pending = {}
            continue
        if op.startswith('s_'):
            continue
        hit = regs(' '.join(t.split()[1:])) & set(pending)
        if hit:
            hazards.append((i, t, sorted(hit)))
    return nloads, hazards


def code_objects(path, arch="gfx950"):
    """The device code objects inside a built library or object file: clang offload bundles (magic, count, then per entry offset /
    size / id) -> [(id, bytes)] of the entries for `arch`."""
    import struct
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        n, = struct.unpack_from("<Q", data, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            ident = data[p:p + idlen].decode()
            p += idlen
            if arch in ident and size:
                out.append((ident, data[i + off:i + off + size]))
        pos = i + 24


def disassemble(blob, tmpdir, name="co.o", objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    import os
    import subprocess
    path = os.path.join(tmpdir, name)
    with open(path, "wb") as f:
        f.write(blob)
    return subprocess.run([objdump, "-d", path], capture_output=True, text=True, check=True).stdout


def main():
    import re
    if sys.argv[1] == "--async-loads":
        lines = open(sys.argv[2]).read().split('\n')
        for name in sys.argv[3:]:
            print(name, async_load_hazards(lines, name))
        return
    if sys.argv[1] == "--blocks":
        lines = open(sys.argv[2]).read().split('\n')
        for name in sys.argv[3:]:
            blocks(lines, name)
        return
    lines = open(sys.argv[1]).read().split('\n')
    for name in sys.argv[2:]:
        b = body(lines, name)
        if b is None:
            print(name, "not found")
            continue
        real = lambda seq: [l for l in seq if l and not l.startswith(('.', ';', '//')) and not l.split(';')[0].rstrip().endswith(':')]
        mix(real(b), name + " [whole kernel]")
        # the largest backward-branch region = the per-row loop of the table kernels (what the issue model counts)
        labels = {l.split(':')[0]: i for i, l in enumerate(b) if re.match(r'^\.LBB\d+_\d+:', l)}
        loops = []
        for i, l in enumerate(b):
            m = re.match(r'^s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        if loops:
            a, e = max(loops, key=lambda t: t[1] - t[0])
            mix(real(b[a:e + 1]), name + " [largest loop]")


if __name__ == "__main__":
    main()
