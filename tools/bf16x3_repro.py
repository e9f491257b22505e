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


def small_fillers(dev):
    """Kernels of <= 24 registers: the only ones that fit beside two 240-register wavefronts of the unclaimed shape-2 kernel
    (k_rans_push 21, k_head_params 20, k_gather_centres 13)."""
    rng = np.random.RandomState(1)
    B, D, K = 32, 2048, 1024
    x = torch.from_numpy(rng.randn(B, 16, 16, 16).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.randn(16).astype(np.float32)).to(dev)
    cen = torch.from_numpy(rng.randn(D, K)).to(dev)
    sym = torch.from_numpy(rng.randint(0, K, (B, D)).astype(np.int32)).to(dev)
    f = torch.from_numpy(rng.randint(1, 1 << 22, (B, D)).astype(np.int32)).to(dev)
    c = torch.from_numpy(rng.randint(0, 1 << 30, (B, D)).astype(np.int32)).to(dev)
    np.random.seed(2)
    words = np.random.randint(1 << 16, (1 << 32) - 1, size=(B, 3000), dtype=np.uint32)
    st = hip.RansState(B, 3000 + 4 * D, dev)
    st.stack[:, :3000] = torch.from_numpy(words.view(np.int32)).to(dev)
    h0 = torch.from_numpy((words[:, -1].astype(np.uint64) << np.uint64(32)).view(np.int64)).to(dev)

    def push():
        st.len.fill_(2999)
        st.head.copy_(h0)
        hip.rans_push(st, f, c)
        return torch.cat([st.head.clone(), st.len.to(torch.int64), st.stack[:, 2999:3400].reshape(-1).to(torch.int64)])
    return {"k_rans_push": push, "k_head_params": lambda: torch.cat(hip.head_params(x, bias, 0)),
            "k_gather_centres": lambda: hip.gather_centres(cen, sym)}


def micro(reps, small=False):
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
    fl = small_fillers(dev) if small else fillers(dev, cols, C)
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


def storm(n_launch=30000):
    """Who is the victim?  The codec's own GEMM shape (36 x 256 x 256 x 512 columns = 32 chains) launched n_launch times on one
    stream, every result compared ON THE DEVICE with the solo result (no host sync in the loop), while a second stream keeps
    launching the <= 24-register kernels that fit beside two unclaimed shape-2 wavefronts, each compared on the device too."""
    dev = "cuda"
    torch.manual_seed(1)
    T, C, cols = 36, 256, 512
    U, V = torch.randn(T, C, C, device=dev), torch.randn(T, C, cols, device=dev)
    Uf = hip.frags_bf16x3(U)
    for k in ("BITSWAP_BF16X3_DIAG", "BITSWAP_BF16X3_SHAPE"):
        os.environ.pop(k, None)
    ref = hip.wino_gemm_bf16x3(Uf, V, 6).clone()
    fl = small_fillers(dev)
    fl_ref = {k: f().clone() for k, f in fl.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    res = {}
    for diag in ("noclaim", None, "noclaim", None):
        if diag:
            os.environ["BITSWAP_BF16X3_DIAG"] = diag
        else:
            os.environ.pop("BITSWAP_BF16X3_DIAG", None)
        bad_g = torch.zeros((), dtype=torch.int64, device=dev)
        bad_n = {k: torch.zeros((), dtype=torch.int64, device=dev) for k in fl}
        outs = [torch.empty_like(ref) for _ in range(4)]
        side.wait_stream(torch.cuda.current_stream())
        for it in range(n_launch // 4):
            for o in outs:
                hip.wino_gemm_bf16x3(Uf, V, 6, out=o)
            for o in outs:
                bad_g += (o != ref).any()
            with torch.cuda.stream(side):
                for k, f in fl.items():
                    bad_n[k] += (f() != fl_ref[k]).any()
            if it % 500 == 499:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        key = (diag or "claim")
        while key in res:
            key += "_again"
        res[key] = {"gemm_launches": n_launch, "gemm_results_differing": int(bad_g), **{f"{k}_differing": int(v) for k, v in bad_n.items()}}
        print("storm", key, res[key], flush=True)
    os.environ.pop("BITSWAP_BF16X3_DIAG", None)
    return res


def codec_leg(focus=False):
    """The forked codec (32 and 100 chains, cifar8 full width, bf16x3 arithmetic) per variant: lossless?"""
    import subprocess
    code = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from bitswap_amd import workload
from bitswap_amd.codec import BitSwapCodec, initial_states
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
out = {}
from bitswap_amd import hip as _hip
if os.environ.get("REPRO_SELFCHECK") == "1":
    _hip.SELFCHECK_BF16X3 = {}
    _hip.SELFCHECK = {}
for B in [int(b) for b in os.environ.get("REPRO_B", "32,100").split(",")]:
    images = workload.synthetic_blocks(B * 2, model.xs, seed=19).view(B, 2, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    if os.environ.get("REPRO_EAGER_FORK") == "1":      # round 4's failing scenario (profiles/archive/visits_r04/dbg3_bf16.py)
        codec.use_graphs = False
        codec.fork = "1"
    if os.environ.get("REPRO_NOFORK") == "1":          # the reference's order on ONE stream, eager
        codec.use_graphs = False
        codec.fork = "0"
    if os.environ.get("REPRO_KEEPALL") == "1":         # nothing a step allocates is freed before the run's end: no block is reused
        stash = []
        real_empty, real_zeros, real_empty_like = torch.empty, torch.zeros, torch.empty_like
        def _k(f):
            def g(*a, **k):
                t = f(*a, **k); stash.append(t); return t
            return g
        torch.empty, torch.zeros, torch.empty_like = _k(real_empty), _k(real_zeros), _k(real_empty_like)
        orig_net = codec._net
        def wrapped(fn, given):
            o = orig_net(fn, given); stash.extend(o); stash.append(given); return o
        codec._net = wrapped
    ok = 0
    NREP = int(os.environ.get("REPRO_REPS", "3"))
    for rep in range(NREP):
        try:
            state, met = codec.compress(images.to("cuda"))
            back = codec.decompress(state, 2)
            good = torch.equal(back.cpu(), images) and state.to_lists() == initial_states(B)
            ok += int(good)
            if not good:
                wrong = (back.cpu() != images).reshape(B, -1).any(1).nonzero().flatten().tolist()
                out.setdefault(f"B{B}_bad_chains", []).append(wrong[:40])
        except Exception as e:
            out[f"B{B}_err{rep}"] = repr(e)[:200]
    out[f"B{B}_lossless"] = f"{ok}/{NREP}"
if _hip.SELFCHECK_BF16X3:
    torch.cuda.synchronize()
    sc = {}
    for shp, a in _hip.SELFCHECK_BF16X3.items():
        if int(a["differing"]):
            sc[str(shp)] = {"launches": a["launches"], "launch_pairs_differing": int(a["differing"]), "elements": int(a["elements"]),
                            "cols": a["cols"].nonzero().flatten().tolist()[:64], "rows": a["rows"].nonzero().flatten().tolist()[:64],
                            "t": a["t"].nonzero().flatten().tolist()}
        else:
            sc[str(shp)] = {"launches": a["launches"], "launch_pairs_differing": 0}
    out["selfcheck"] = sc
if _hip.SELFCHECK:
    torch.cuda.synchronize()
    out["selfcheck_others"] = {k: {"calls": a["calls"], "call_pairs_differing": int(a["differing"]),
                                   "images": a["images"].nonzero().flatten().tolist()[:64]} for k, a in _hip.SELFCHECK.items()}
print("RESULT " + json.dumps(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = {}
    plan = (("2", None, "0"), ("2", "noclaim", "1"), ("1", None, "1"), ("1", "noclaim", "1"), ("1", "noclaim", "0"),
            ("1", "stray_exit", "1"), ("1", "stray_exit", "0"))
    if focus:      # the scenario that failed in round 5 visit c (shape 2 without the claim, eager forked codec, 32 chains): statistics,
        # with the CONTROL the round-4 hunt never ran: the same eager forked codec on the default fp32 GEMM ("fp32" below)
        plan = (("2", "noclaim", "nofork"), ("2", "noclaim", "keepall"), ("2", "noclaim", "1"), ("2", "noclaim", "nofork"), ("2", "noclaim", "keepall"))
    for shape, diag, eager in plan:
        env = dict(os.environ, BITSWAP_GEMM_ARITH="bf16x3", BITSWAP_BF16X3_SHAPE=shape, REPRO_EAGER_FORK="0" if eager == "0" else "1")
        if eager == "nofork":
            env["REPRO_EAGER_FORK"], env["REPRO_NOFORK"] = "0", "1"
        if eager == "keepall":
            env["REPRO_KEEPALL"] = "1"
        if eager == "nocache":      # no caching allocator: a freed block is never handed to another stream's allocation early
            env["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
            env["PYTORCH_NO_HIP_MEMORY_CACHING"] = "1"
        if shape == "fp32":
            env.pop("BITSWAP_GEMM_ARITH"), env.pop("BITSWAP_BF16X3_SHAPE")
        if focus:
            env.update(REPRO_REPS=os.environ.get("REPRO_FOCUS_REPS", "150"), REPRO_B="32")
        env.pop("BITSWAP_BF16X3_DIAG", None)
        if diag:
            env["BITSWAP_BF16X3_DIAG"] = diag
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        key = f"shape{shape}_{diag or 'claim'}_{ {'1': 'eager_fork', '0': 'graph', 'nocache': 'eager_fork_no_caching_allocator', 'nofork': 'eager_ONE_stream', 'keepall': 'eager_fork_nothing_freed'}[eager]}"
        while key in res:
            key += "_again"
        res[key] = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-300:]}
        print("codec", key, res[key], flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--codec", action="store_true")
    ap.add_argument("--focus", action="store_true", help="codec leg only: the failing scenario with and without counted waits, 14 runs each")
    ap.add_argument("--storm", type=int, default=0, help="GEMM launches of the victim hunt (0: skip)")
    ap.add_argument("--small", action="store_true", help="micro leg with the neighbours that fit beside the unclaimed shape-2 kernel (<= 32 registers)")
    a = ap.parse_args()
    out = {}
    if a.storm:
        out["storm"] = storm(a.storm)
    if (not a.focus and not a.storm) or a.small:
        out["micro"] = micro(a.reps, small=a.small)
    if a.codec or a.focus:
        out["codec"] = codec_leg(focus=a.focus)
    print(json.dumps(out, indent=1))
