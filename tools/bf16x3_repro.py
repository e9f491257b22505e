#!/usr/bin/env python3
"""Repro hunt for the bf16x3 co-residency failure (LABNOTES r04): bs_wino_gemm_bf16x3 WITHOUT its whole-register-
share claim (BITSWAP_BF16X3_DIAG=noclaim / noclaim_strict, shapes 1 and 2) on one stream while small-register kernels of the
codec loop on a second one -- k_logistic<4> (the pixel tables, <= 64 registers), k_wino_fused, k_logistic<16> encode flavour --
bitwise against the solo run; then the forked codec end to end on the same variants (lossless?).
    python tools/bf16x3_repro.py [--reps 40] [--codec]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bitswap_amd import hip  # noqa: E402
from bitswap_amd.bins import uniform_step  # noqa: E402


def fillers(dev, cols, C):
    rng = np.random.RandomState(0)
    D, K, B = 3072, 256, 64
    e_np = np.stack([np.linspace(-1.0, 1.0, K + 1)[1:-1]] * D)
    e = torch.from_numpy(e_np[:1]).to(dev).expand(D, -1)
    step = torch.from_numpy(uniform_step(e_np)).to(dev)
    mu = torch.from_numpy(rng.uniform(-1, 1, (B, D)).astype(np.float32)).to(dev)
    sc = torch.from_numpy(rng.uniform(0.02, 0.7, (B, D)).astype(np.float32)).to(dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    tab = torch.empty((B, D, hip.wave_ld(K)), dtype=torch.int32, device=dev)
    sym = torch.from_numpy(rng.randint(0, K, (B, D)).astype(np.int32)).to(dev)
    fo = (torch.empty((B, D), dtype=torch.int32, device=dev), torch.empty((B, D), dtype=torch.int32, device=dev))
    Dz, Kz = 2048, 1024
    lo, hi = rng.uniform(-8, -2, Dz), rng.uniform(2, 8, Dz)
    ez_np = np.stack([np.linspace(a, b, Kz + 1)[1:-1] for a, b in zip(lo, hi)])
    ez, stepz = torch.from_numpy(ez_np).to(dev), torch.from_numpy(uniform_step(ez_np)).to(dev)
    muz = torch.from_numpy(rng.randn(16, Dz).astype(np.float32)).to(dev)
    scz = torch.from_numpy(rng.uniform(0.1, 1, (16, Dz)).astype(np.float32)).to(dev)
    symz = torch.from_numpy(rng.randint(0, Kz, (16, Dz)).astype(np.int32)).to(dev)
    stz = torch.zeros(16, dtype=torch.int32, device=dev)
    x = torch.randn(cols // 16, C, 16, 16, device=dev)
    bias = torch.randn(C, device=dev)
    return {
        "k_logistic<4> tables": lambda: hip.logistic_tables(e, mu, sc, 31, 8, layout=hip.LAYOUT_WAVE, step=step, status=status),
        "k_logistic<4> fc": lambda: hip.logistic_fc(e, mu, sc, sym, status, 31, 8, step=step)[0],
        "k_logistic<16> fc": lambda: hip.logistic_fc(ez, muz, scz, symz, stz, 31, 10, step=stepz)[0],
        "k_wino_fused": lambda: hip.wino_fused(x, tuple(x.shape), 0, bias, None, True, ts_out=6)[2],
    }


def micro(reps):
    dev = "cuda"
    torch.manual_seed(0)
    T, C, cols = 36, 256, 2048
    U, V = torch.randn(T, C, C, device=dev), torch.randn(T, C, cols, device=dev)
    Uf = hip.frags_bf16x3(U)
    os.environ.pop("BITSWAP_BF16X3_DIAG", None)
    os.environ.pop("BITSWAP_BF16X3_SHAPE", None)
    ref = hip.wino_gemm_bf16x3(Uf, V, 6).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    fl = fillers(dev, cols, C)
    res = {}
    fl_ref = {name: f().clone() for name, f in fl.items()}
    torch.cuda.synchronize()
    for shape in ("2", "1"):
        for diag in (None, "noclaim", "noclaim_strict") + (("stray_exit",) if shape == "1" else ()):
            os.environ["BITSWAP_BF16X3_SHAPE"] = shape
            if diag:
                os.environ["BITSWAP_BF16X3_DIAG"] = diag
            else:
                os.environ.pop("BITSWAP_BF16X3_DIAG", None)
            solo = hip.wino_gemm_bf16x3(Uf, V, 6)
            torch.cuda.synchronize()
            key = f"shape{shape}_{diag or 'claim'}"
            res[key] = {"solo_equals_product": bool(torch.equal(solo, ref))}
            for name, f in fl.items():
                bad, victims, nv = 0, 0, 0
                for _ in range(reps):
                    outs = []
                    with torch.cuda.stream(side):
                        for _ in range(6):
                            outs.append(f())
                    out = hip.wino_gemm_bf16x3(Uf, V, 6)
                    out2 = hip.wino_gemm_bf16x3(Uf, V, 6)
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            outs.append(f())
                    torch.cuda.synchronize()
                    bad += int(not torch.equal(out, ref)) + int(not torch.equal(out2, ref))
                    victims += sum(int(not torch.equal(o, fl_ref[name])) for o in outs)      # the NEIGHBOUR's results
                    nv += len(outs)
                res[key][name] = f"GEMM {bad}/{2 * reps} differ, neighbour {victims}/{nv} differ"
            print(key, res[key], flush=True)
    os.environ.pop("BITSWAP_BF16X3_DIAG", None)
    os.environ.pop("BITSWAP_BF16X3_SHAPE", None)
    return res


def codec_leg():
    """The forked codec (32 and 100 chains, cifar8 full width, bf16x3 arithmetic) per variant: lossless?"""
    import subprocess
    code = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from bitswap_amd import workload
from bitswap_amd.codec import BitSwapCodec, initial_states
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
out = {}
for B in (32, 100):
    images = workload.synthetic_blocks(B * 2, model.xs, seed=19).view(B, 2, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    if os.environ.get("REPRO_EAGER_FORK") == "1":      # round 4's failing scenario (profiles/archive/visits_r04/dbg3_bf16.py)
        codec.use_graphs = False
        codec.fork = "1"
    ok = 0
    for rep in range(3):
        try:
            state, met = codec.compress(images.to("cuda"))
            back = codec.decompress(state, 2)
            ok += int(torch.equal(back.cpu(), images) and state.to_lists() == initial_states(B))
        except Exception as e:
            out[f"B{B}_err{rep}"] = repr(e)[:200]
    out[f"B{B}_lossless"] = f"{ok}/3"
print("RESULT " + json.dumps(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = {}
    for shape, diag, eager in (("2", None, "0"), ("2", "noclaim", "1"), ("1", None, "1"), ("1", "noclaim", "1"), ("1", "noclaim", "0"),
                               ("1", "stray_exit", "1"), ("1", "stray_exit", "0")):
        env = dict(os.environ, BITSWAP_GEMM_ARITH="bf16x3", BITSWAP_BF16X3_SHAPE=shape, REPRO_EAGER_FORK=eager)
        env.pop("BITSWAP_BF16X3_DIAG", None)
        if diag:
            env["BITSWAP_BF16X3_DIAG"] = diag
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        key = f"shape{shape}_{diag or 'claim'}_{'eager_fork' if eager == '1' else 'graph'}"
        res[key] = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-300:]}
        print("codec", key, res[key], flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--codec", action="store_true")
    a = ap.parse_args()
    out = {"micro": micro(a.reps)}
    if a.codec:
        out["codec"] = codec_leg()
    print(json.dumps(out, indent=1))
