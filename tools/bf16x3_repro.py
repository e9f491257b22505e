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
if os.environ.get("REPRO_NOP_AFTER"):       # a one-wavefront no-op launch behind every call of the named kernel wrappers
    _nopbuf = torch.zeros(4, dtype=torch.int32, device="cuda")
    def _nop_after(name):
        orig = getattr(_hip, name)
        def w(*a, **k):
            r = orig(*a, **k)
            _hip.load().bs_debug_where(_hip._ptr(_nopbuf), 1, 0, _hip._stream())
            return r
        setattr(_hip, name, w)
    for _n in os.environ["REPRO_NOP_AFTER"].split(","):
        _nop_after(_n)
for B in [int(b) for b in os.environ.get("REPRO_B", "32,100").split(",")]:
    images = workload.synthetic_blocks(B * 2, model.xs, seed=19).view(B, 2, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    if os.environ.get("REPRO_EAGER_FORK") == "1":      # round 4's failing scenario (profiles/archive/visits_r04/dbg3_bf16.py)
        codec.use_graphs = False
        codec.fork = "1"
    if os.environ.get("REPRO_NOFORK") == "1":          # the reference's order on ONE stream, eager
        codec.use_graphs = False
        codec.fork = "0"
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
        plan = (("fp32", None, "0"), ("fp32", None, "1"), ("fp32", None, "0"), ("fp32", None, "1"))   # the DEFAULT route: graph replay / eager, forked
        if os.environ.get("REPRO_BISECT") == "1":     # swap single kernels of the stacks for their alternatives (no launch added)
            plan = (("2", "noclaim", "1"), ("2", "noclaim_coherent", "1"), ("2", "noclaim", "1"), ("2", "noclaim_coherent", "1"))
    for shape, diag, eager in plan:
        env = dict(os.environ, BITSWAP_GEMM_ARITH="bf16x3", BITSWAP_BF16X3_SHAPE=shape, REPRO_EAGER_FORK="0" if eager == "0" else "1")
        if eager.startswith("nop:"):
            env["REPRO_NOP_AFTER"] = eager[4:]
        if eager == "fusedplain":      # k_wino_fused with ordinary instead of nontemporal loads of M / stores of V
            env["BITSWAP_FUSED_PLAIN"] = "1"
        if eager == "nofusedin":       # the 3x3 input convs through MIOpen + k_wino_fused<0, 6> instead of k_conv3_wino
            env["BITSWAP_FUSED_INPUTS"] = "0"
        if eager == "nofork":
            env["REPRO_EAGER_FORK"], env["REPRO_NOFORK"] = "0", "1"
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
        key = f"shape{shape}_{diag or 'claim'}_nop_after_{eager[4:]}" if eager.startswith("nop:") else f"shape{shape}_{diag or 'claim'}_{ {'1': 'eager_fork', '0': 'graph', 'nocache': 'eager_fork_no_caching_allocator', 'nofork': 'eager_ONE_stream', 'nofusedin': 'eager_fork_without_k_conv3_wino', 'fusedplain': 'eager_fork_k_wino_fused_plain_loads_stores'}[eager]}"
        while key in res:
            key += "_again"
        res[key] = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-300:]}
        print("codec", key, res[key], flush=True)
    return res


TRAIL_CODE = r"""
import os, sys, json, torch
sys.path.insert(0, "__ROOT__")
from bitswap_amd import workload
from bitswap_amd.codec import BitSwapCodec, initial_states
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
B, n = 32, 2
images = workload.synthetic_blocks(B * n, model.xs, seed=19).view(B, n, -1).to(torch.int32)

def make(fork):
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    codec.use_graphs = False
    codec.fork = fork
    trail = {"ops": [], "nets": {}}
    be = codec.backend
    o_pop, o_pp, o_pt, o_net = be.pop, be.push_params, be.push_table, codec._net
    def pop(state, cdf, K, bits, centres=None):
        r = o_pop(state, cdf, K, bits, centres=centres)
        trail["ops"].append(("pop", state.head.clone(), r[0].sum(1)))
        return r
    def push_params(state, *a, **k):
        o_pp(state, *a, **k)
        trail["ops"].append(("push", state.head.clone(), None))
    def push_table(state, *a, **k):
        o_pt(state, *a, **k)
        trail["ops"].append(("push_prior", state.head.clone(), None))
    def net(fn, given):
        CUR["key"] = (fn.bs_key, len(trail["nets"].get(fn.bs_key, [])))
        CUR["trail"] = trail
        mu, sc = o_net(fn, given)
        CUR["key"] = None
        trail["nets"].setdefault(fn.bs_key, []).append((mu.sum(1), sc.sum(1), given.reshape(given.shape[0], -1).sum(1)))
        return mu, sc
    be.pop, be.push_params, be.push_table, codec._net = pop, push_params, push_table, net
    return codec, trail

# per-kernel trail inside a stack: per-image sums of every output tensor of every stack kernel
from bitswap_amd import hip as _hip
CUR = {"key": None, "trail": None}
def _per_image(x, N):
    if x is None or not torch.is_tensor(x) or x.numel() % N:
        return None
    if x.dim() == 4 and x.shape[0] == N:
        return x.reshape(N, -1).sum(1)
    if x.dim() == 3 and x.shape[2] % N == 0:                 # [ts^2 or T, C, N * tiles]
        return x.reshape(x.shape[0] * x.shape[1], N, -1).sum((0, 2))
    if x.dim() == 2 and x.shape[0] == N:
        return x.sum(1)
    return None
def _wrap(name):
    orig = getattr(_hip, name)
    def w(*a, **k):
        out = orig(*a, **k)
        if CUR["key"] is not None:
            o = out if isinstance(out, (tuple, list)) else (out,)
            sums = [_per_image(t, 32) for t in o]
            CUR["trail"].setdefault("kern", {}).setdefault(CUR["key"], []).append((name, [s_ for s_ in sums if s_ is not None]))
        return out
    setattr(_hip, name, w)
for _n in ("conv3_wino", "wino_fused", "wino_gemm", "wino_gemm_bf16x3", "head_params", "small_k_gemm", "wino_in", "wino_out"):
    _wrap(_n)

def run(codec, trail):
    trail["ops"].clear(); trail["nets"].clear(); trail.setdefault("kern", {}).clear()
    state, met = codec.compress(images.to("cuda"))
    back = codec.decompress(state, n)
    torch.cuda.synchronize()
    good = torch.equal(back.cpu(), images) and state.to_lists() == initial_states(B)
    return good, {"ops": [(k, h.cpu(), None if s is None else s.cpu()) for k, h, s in trail["ops"]],
                  "nets": {k: [tuple(t.cpu() for t in e) for e in v] for k, v in trail["nets"].items()},
                  "kern": {k: [(nm, [t.cpu() for t in ss]) for nm, ss in v] for k, v in trail.get("kern", {}).items()}}

ref_codec, ref_trail = make("0")
ok, ref = run(ref_codec, ref_trail)
assert ok
ok2, ref2 = run(ref_codec, ref_trail)
same = all(torch.equal(a[1], b[1]) for a, b in zip(ref["ops"], ref2["ops"]))
codec, trail = make("1")
out = {"reference_run_repeats": bool(same), "ops_per_run": len(ref["ops"]), "findings": [], "lossless": 0}
NREP = int(os.environ.get("REPRO_REPS", "100"))
for rep in range(NREP):
    good, t = run(codec, trail)
    out["lossless"] += int(good)
    if good:
        continue
    f = {"run": rep}
    for i, (a, b) in enumerate(zip(ref["ops"], t["ops"])):
        if a[0] != b[0] or not torch.equal(a[1], b[1]):
            f["first_stack_op"] = {"index": i, "kind": b[0], "of": len(ref["ops"]), "chains": (a[1] != b[1]).nonzero().flatten().tolist(),
                                   "popped_symbols_differ": (None if a[2] is None else (a[2] != b[2]).nonzero().flatten().tolist())}
            break
    nets = []
    for key, lst in ref["nets"].items():
        for j, (a, b) in enumerate(zip(lst, t["nets"].get(key, []))):
            d_in, d_mu, d_sc = (a[2] != b[2]), (a[0] != b[0]), (a[1] != b[1])
            if bool(d_in.any()) or bool(d_mu.any()) or bool(d_sc.any()):
                nets.append({"stack": list(key) if isinstance(key, tuple) else str(key), "call": j, "input_differs": d_in.nonzero().flatten().tolist(),
                             "mu_differs": d_mu.nonzero().flatten().tolist(), "scale_differs": d_sc.nonzero().flatten().tolist()})
                break
    # stacks whose OUTPUT differs although their input is the reference's: the origin
    f["stacks_wrong_on_right_input"] = [x for x in nets if not x["input_differs"]][:6]
    for x in f["stacks_wrong_on_right_input"][:2]:          # ... and inside such a stack: the first kernel whose output differs
        key = (tuple(x["stack"]), x["call"])
        seq_r, seq_t = ref["kern"].get(key, []), t["kern"].get(key, [])
        x["kernels_in_stack"] = [nm for nm, _ in seq_t]
        for ki, ((nr, sr), (nt, st_)) in enumerate(zip(seq_r, seq_t)):
            bad = [i for i, (u, v) in enumerate(zip(sr, st_)) if not torch.equal(u, v)]
            if nr != nt or bad:
                x["first_wrong_kernel"] = {"index": ki, "kernel": nt, "outputs_differing": bad,
                                           "images": sorted(set(sum(((u != v).nonzero().flatten().tolist() for u, v in zip(sr, st_)), [])))}
                break
    f["stacks_with_wrong_input"] = [(x["stack"], x["call"]) for x in nets if x["input_differs"]][:8]
    out["findings"].append(f)
out["lossless"] = f"{out['lossless']}/{NREP}"
print("RESULT " + json.dumps(out))
"""


RECORD_CODE = r"""
import os, sys, json, torch
sys.path.insert(0, "__ROOT__")
from bitswap_amd import workload
from bitswap_amd.codec import BitSwapCodec, initial_states
from bitswap_amd import hip as _hip
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
B, n = 32, 2
images = workload.synthetic_blocks(B * n, model.xs, seed=19).view(B, n, -1).to(torch.int32)
codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
codec.use_graphs = False
codec.fork = os.environ.get("REPRO_FORK", "1")

# NO launch is added to the run: every stack kernel call only leaves references to its arguments and results behind (so no buffer
# of the run is reused before its end); after a failing run every call is repeated alone on one stream and compared bit by bit.
REC, ON = [], [False]
MAIN = torch.cuda.current_stream().cuda_stream
# REPRO_POISON=nan | value: after every run each float32 RESULT of a stack kernel is overwritten with it before its memory goes
# back to the allocator -- a later kernel that reads memory its producer has not (visibly) written yet shows the poison
POISON = os.environ.get("REPRO_POISON")
POISON_VALUE = float(POISON) if POISON else -1.0e30
SWEEP = int(os.environ.get("REPRO_SWEEP_MB", "0"))      # after poisoning: read this much other memory (evicts the poison lines from every L2)
sweep_buf = torch.zeros(SWEEP * (1 << 18), dtype=torch.float32, device="cuda") if SWEEP else None
def _wrap(name):
    orig = getattr(_hip, name)
    def w(*a, **k):
        out = orig(*a, **k)
        if ON[0]:
            REC.append((name, orig, a, k, out, torch.cuda.current_stream().cuda_stream != MAIN))
        return out
    setattr(_hip, name, w)
for _n in ("conv3_wino", "wino_fused", "wino_gemm", "wino_gemm_bf16x3", "head_params", "small_k_gemm", "wino_in", "wino_out"):
    _wrap(_n)

def tensors(o):
    return [t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)]

def describe(x, y):
    d = x != y
    idx = d.nonzero()
    r = {"shape": list(x.shape), "differing": int(idx.shape[0]), "of": x.numel()}
    for dim in range(x.dim()):
        u = idx[:, dim].unique()
        r[f"dim{dim}"] = {"distinct": int(u.numel()), "values": u[:48].tolist()}
    if x.dim() == 3 and x.shape[2] % B == 0:           # [ts^2, C, chain * tiles]
        T = x.shape[2] // B
        r["chains"] = (idx[:, 2] // T).unique().tolist()
        r["tiles"] = (idx[:, 2] % T).unique().tolist()
    elif x.dim() >= 2 and x.shape[0] == B:
        r["chains"] = idx[:, 0].unique().tolist()
    xs, ys = x[d][:12], y[d][:12]
    r["first"] = [{"at": idx[i].tolist(), "run": float(xs[i]), "alone": float(ys[i]),
                   "run_bits": hex(int(xs[i].view(torch.int32)) & 0xffffffff), "alone_bits": hex(int(ys[i].view(torch.int32)) & 0xffffffff)}
                  for i in range(xs.numel())] if x.dtype == torch.float32 else []
    # is the run's value a value the lone launch has elsewhere in the tensor (a shifted or stale read)?
    if x.dtype == torch.float32 and xs.numel():
        r["run_values_found_elsewhere_alone"] = [int((y == v).sum()) for v in xs[:6]]
        r["run_values_zero"] = int((x[d] == 0).sum())
        r["run_values_nan"] = int(torch.isnan(x[d]).sum())
        r["run_values_poison"] = int((x[d] == POISON_VALUE).sum())
    return r

BT6 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)

def autopsy(orig, a, k, v_run, v_alone):
    # A failing wino_fused(M [36, C, 32 * 16] -> V [36, C, 32 * 16]) call: which elements of the ACTIVATED PLANE (between the two
    # transforms) explain V_run - V_alone, by least squares through B^T d B; what value the kernel must have held there; and where
    # in the run's tensors such values exist.
    d = v_run != v_alone
    idx = d.nonzero()
    c, chain = int(idx[0, 1]), int(idx[0, 2]) // 16
    dv = (v_run[:, c, chain * 16:chain * 16 + 16].double() - v_alone[:, c, chain * 16:chain * 16 + 16].double()).cpu()   # [36, 16 tiles]
    # forward operator: plane perturbation [16, 16] -> V of the 16 tiles (window of tile (ty, tx): rows 4 ty - 1 .. 4 ty + 4)
    cols_ = []
    for pr in range(16):
        for pc in range(16):
            out_ = torch.zeros(36, 16, dtype=torch.float64)
            for ty in range(4):
                for tx in range(4):
                    wr, wc = pr - (4 * ty - 1), pc - (4 * tx - 1)
                    if 0 <= wr < 6 and 0 <= wc < 6:
                        out_[:, ty * 4 + tx] += torch.outer(BT6[:, wr], BT6[:, wc]).reshape(36)
            cols_.append(out_.reshape(-1))
    A = torch.stack(cols_, 1)                                  # [576, 256]
    sol = torch.linalg.lstsq(A, dv.reshape(-1, 1)).solution.reshape(16, 16)
    resid = float((A @ sol.reshape(-1, 1) - dv.reshape(-1, 1)).abs().max())
    nz = (sol.abs() > 1e-3 * float(dv.abs().max())).nonzero().tolist()
    r = {"channel": c, "chain": chain, "plane_elements_changed": [(pr, pc, round(float(sol[pr, pc]), 5)) for pr, pc in nz][:40],
         "lstsq_residual": resid, "max_dV": float(dv.abs().max())}
    # the activated plane alone (same call, act plane requested)
    try:
        kk = dict(k); kk["want_act"] = True
        plane = orig(*a, **kk)[1]                              # [32, C, 16, 16]
        if plane is not None:
            r["plane_alone_at_changed"] = [round(float(plane[chain, c, pr, pc]), 5) for pr, pc in nz][:40]
    except Exception as e:
        r["plane_error"] = repr(e)[:200]
    return r, nz, sol

def find_vector(vec, tol):
    # 16 consecutive floats (64-byte aligned) anywhere in the run's recorded float32 tensors within tol of vec (a [16] tensor)
    hits, seen = [], set()
    v = vec.to("cuda", torch.float32).reshape(1, 16)
    for i, (name, orig, a, k, o, aux) in enumerate(REC):
        for role, ts in (("arg", [t for t in a if torch.is_tensor(t)]), ("out", tensors(o))):
            for j, t in enumerate(ts):
                if t.dtype != torch.float32 or t.numel() % 16 or t.data_ptr() in seen or not t.is_contiguous():
                    continue
                seen.add(t.data_ptr())
                dist = (t.reshape(-1, 16) - v).abs().amax(1)
                m = float(dist.min())
                if m <= tol:
                    row = int(dist.argmin())
                    hits.append({"call": i, "kernel": name, "role": role, "tensor": j, "shape": list(t.shape), "row16": row,
                                 "index": [int(x) for x in torch.unravel_index(torch.tensor(row * 16), t.shape)], "max_abs_diff": m})
    return hits[:12]

def find_lanes(targets, scale, skip_call):
    # targets: {name: [16] tensor}: the 16 tiles' values of ONE plane element.  Sources: for every plane-shaped float32 tensor of the
    # run [32, C, 16, 16] (arguments, results, and the sum / activated planes of every TS_IN = 6 fused call repeated alone) the
    # 16 values X[n, c, r' + 4 ty, q' + 4 tx]; for Winograd-domain tensors [.., C, 32 * 16] the 16 consecutive values of a chain.
    hits, seen = [], set()
    tg = {k_: v_.to("cuda", torch.float32) for k_, v_ in targets.items()}
    def scan(x, label):
        if x is None or x.dtype != torch.float32:
            return
        if x.dim() == 4 and x.shape[0] == B and x.shape[2:] == (16, 16):
            g = x.reshape(B, x.shape[1], 4, 4, 4, 4).permute(0, 1, 3, 5, 2, 4).reshape(B, x.shape[1], 16, 16)   # [n, c, (r', q'), (ty, tx)]
        elif x.dim() == 3 and x.shape[2] == B * 16:
            g = x.reshape(x.shape[0], x.shape[1], B, 16).permute(2, 1, 0, 3)                                    # [n, c, t, tile]
        else:
            return
        for nm, t in tg.items():
            dist = (g - t.reshape(1, 1, 1, 16)).abs().amax(3)
            m = float(dist.min())
            if m <= 1e-4 * scale:
                i = int(dist.argmin())
                n_, c_, e_ = i // (g.shape[1] * g.shape[2]), (i // g.shape[2]) % g.shape[1], i % g.shape[2]
                hits.append({"target": nm, "source": label, "shape": list(x.shape), "chain": n_, "channel": c_, "element_or_t": e_, "max_abs_diff": m})
    for i, (name, orig, a, k, o, aux) in enumerate(REC):
        for role, ts in (("arg", [t for t in a if torch.is_tensor(t)]), ("out", tensors(o))):
            for j, t in enumerate(ts):
                if t.data_ptr() in seen:
                    continue
                seen.add(t.data_ptr())
                scan(t, f"call {i} {name} {role}{j}" + (" aux" if aux else " main"))
        if name == "wino_fused" and len(a) >= 3 and a[2] == 6:
            try:
                so, ao, _ = orig(*a, **dict(k, want_sum=True, want_act=True))
                tag = f"call {i} wino_fused" + (" aux" if aux else " main") + (" (THE FAILING CALL)" if i == skip_call else "")
                scan(so, tag + " sum plane alone")
                scan(ao, tag + " act plane alone")
            except Exception as e:
                hits.append({"error": repr(e)[:200], "call": i})
        if len(hits) > 40:
            break
    return hits

def replay():
    found = []
    for i, (name, orig, a, k, out, aux) in enumerate(REC):
        out2 = orig(*a, **k)
        bad = [(j, x, y) for j, (x, y) in enumerate(zip(tensors(out), tensors(out2))) if x.shape != y.shape or not torch.equal(x, y)]
        if bad:
            found.append({"call": i, "of": len(REC), "kernel": name, "aux_stream": bool(aux),
                          "before": [r[0] for r in REC[max(0, i - 3):i]],
                          "args": [list(t.shape) if torch.is_tensor(t) else (type(t).__name__ if not isinstance(t, (int, float, bool, tuple, type(None))) else t) for t in a],
                          "outputs": {j: describe(x, y) for j, x, y in bad}})
            if name == "wino_fused" and len(found) == 1 and len(a) >= 3 and a[2] == 6:
                try:
                    j, x, y = bad[0]
                    rep_, nz, sol = autopsy(orig, a, k, x, y)
                    found[-1]["autopsy"] = rep_
                    # hypothesis: ONE element of the inverse transform's input was different per lane.  For plane element (0, 0)
                    # of a tile that is M position t = 0, for (3, 0) it is t = 30 (the only M positions that reach one output alone)
                    rq = sorted({(pr % 4, pc % 4) for pr, pc in nz})
                    found[-1]["autopsy"]["tile_elements"] = rq
                    M = a[0]
                    c, chain = rep_["channel"], rep_["chain"]
                    # raw material for an offline analysis: the 36 inputs of each of the 16 lanes, the bias, the plane alone / its change
                    found[-1]["autopsy"]["M_lanes"] = [[float(v) for v in row] for row in M[:, c, chain * 16:chain * 16 + 16].cpu()]
                    found[-1]["autopsy"]["bias"] = None if a[3] is None else float(a[3][c])
                    found[-1]["autopsy"]["plane_delta"] = [[float(v) for v in row] for row in sol]
                    pl_ = orig(*a, **dict(k, want_act=True, want_sum=True))
                    found[-1]["autopsy"]["plane_act_alone"] = [[float(v) for v in row] for row in pl_[1][chain, c].cpu()]
                    found[-1]["autopsy"]["plane_sum_alone"] = [[float(v) for v in row] for row in pl_[0][chain, c].cpu()]
                    if len(rq) == 1:
                        pos = [(4 * (t // 4) + rq[0][0], 4 * (t % 4) + rq[0][1]) for t in range(16)]
                        a_al = torch.stack([pl_[1][chain, c, i_, j_] for i_, j_ in pos]).double().cpu()
                        s_al = torch.stack([pl_[0][chain, c, i_, j_] for i_, j_ in pos]).double().cpu()
                        a_sn = a_al + torch.stack([sol[i_, j_] for i_, j_ in pos])
                        s_sn = torch.where(a_sn > 0, a_sn, torch.log1p(a_sn.clamp(min=-0.999999)))
                        found[-1]["autopsy"]["a_seen"] = [float(v) for v in a_sn]
                        found[-1]["autopsy"]["lane_search"] = find_lanes({"a_seen": a_sn, "s_seen": s_sn, "delta_s": s_sn - s_al, "delta_a": a_sn - a_al},
                                                                         float(a_al.abs().max()) + 1.0, i)
                    if len(rq) == 1 and rq[0] in ((0, 0), (3, 0), (0, 3), (3, 3)):
                        tp = {(0, 0): 0, (3, 0): 30, (0, 3): 5, (3, 3): 35}[rq[0]]
                        kk = dict(k); kk["want_sum"] = True
                        s_alone = orig(*a, **kk)[0]                                   # pre-activation sums [32, C, 16, 16]
                        pl = orig(*a, **dict(k, want_act=True))[1]
                        a_alone = torch.stack([pl[chain, c, 4 * (t // 4) + rq[0][0], 4 * (t % 4) + rq[0][1]] for t in range(16)]).double().cpu()
                        dlt = torch.stack([sol[4 * (t // 4) + rq[0][0], 4 * (t % 4) + rq[0][1]] for t in range(16)])
                        a_seen = a_alone + dlt
                        s_seen = torch.where(a_seen > 0, a_seen, torch.log1p(a_seen.clamp(min=-0.999999)))      # ELU^-1
                        s_al = torch.stack([s_alone[chain, c, 4 * (t // 4) + rq[0][0], 4 * (t % 4) + rq[0][1]] for t in range(16)]).double().cpu() if s_alone is not None else None
                        m_true = M[tp, c, chain * 16:chain * 16 + 16].double().cpu()
                        if s_al is not None:
                            m_seen = m_true + (s_seen - s_al)
                            found[-1]["autopsy"].update(M_position=tp, M_true=[round(float(v), 4) for v in m_true], M_seen=[round(float(v), 4) for v in m_seen])
                            scale = float(m_true.abs().max()) + 1.0
                            found[-1]["autopsy"]["M_seen_found_at"] = find_vector(m_seen, 2e-3 * scale)
                            found[-1]["autopsy"]["M_true_found_at"] = find_vector(m_true, 1e-6)[:3]
                except Exception as e:
                    found[-1]["autopsy_error"] = repr(e)[:300]
            if len(found) >= 2:
                break
    return found

out = {"runs": 0, "lossless": 0, "baseline_replay_mismatches": None, "failures": [], "poison": POISON, "sweep_mb": SWEEP}
NREP = int(os.environ.get("REPRO_REPS", "100"))
for rep in range(NREP):
    REC.clear()
    ON[0] = True
    state, met = codec.compress(images.to("cuda"))
    back = codec.decompress(state, n)
    torch.cuda.synchronize()
    ON[0] = False
    good = torch.equal(back.cpu(), images) and state.to_lists() == initial_states(B)
    out["runs"] += 1
    out["lossless"] += int(good)
    if good and out["baseline_replay_mismatches"] is None:
        out["calls_per_run"] = len(REC)
        out["baseline_replay_mismatches"] = [(f["kernel"], f["call"]) for f in replay()]      # [] expected: every call repeats
    if not good:
        wrong = (back.cpu() != images).reshape(B, -1).any(1).nonzero().flatten().tolist()
        out["failures"].append({"run": rep, "bad_chains": wrong, "calls": len(REC), "first_calls_that_do_not_repeat": replay()})
        print("PARTIAL " + json.dumps(out["failures"][-1]), flush=True)
        if len(out["failures"]) >= int(os.environ.get("REPRO_MAX_FAIL", "4")):
            break
    if POISON and rep >= 1:
        seen = set()
        for (name, orig, a, k, o, aux) in REC:
            for t in tensors(o):
                if t.dtype == torch.float32 and t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    t.fill_(POISON_VALUE)
        if sweep_buf is not None:
            sweep_buf.add_(1.0)
        torch.cuda.synchronize()
print("RESULT " + json.dumps(out))
"""


def record_leg(reps=150):
    """Which kernel call of a failing forked run does not repeat?  (no launch added to the run, see RECORD_CODE)"""
    import subprocess
    code = RECORD_CODE.replace("__ROOT__", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, BITSWAP_GEMM_ARITH="bf16x3", BITSWAP_BF16X3_SHAPE="2", BITSWAP_FORK="1", REPRO_REPS=str(reps))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    res = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-1500:],
                                                 "partial": [l[8:] for l in r.stdout.splitlines() if l.startswith("PARTIAL ")]}
    print("record", json.dumps(res), flush=True)
    return res


def pair_leg(rounds=100, ring=160):
    """Two kernels only: k_wino_fused<6,6> (M [36, 256, 32 x 16] -> V, the call that fails in the codec) `ring` times into `ring`
    different buffers on a side stream while the main stream runs the GEMM of the 32-chain codec on other operands (144
    workgroups: the chip half empty); after a synchronize every V is compared with the first one.  Per library
    (BITSWAP_HIP_LIB: the packed build / the product) and per GEMM (bf16x3 shape 2, fp32, none)."""
    dev = "cuda"
    torch.manual_seed(0)
    T, C, cols, N = 36, 256, 512, 32
    U, Vg = torch.randn(T, C, C, device=dev), torch.randn(T, C, cols, device=dev)
    Uf = hip.frags_bf16x3(U)
    M = torch.randn(T, C, cols, device=dev) * 2.0
    bias = torch.randn(C, device=dev)
    shape = (N, C, 16, 16)
    ref = hip.wino_fused(M, shape, 6, bias, None, True, ts_out=6)[2].clone()
    side = torch.cuda.Stream()
    res = {"library": os.environ.get("BITSWAP_HIP_LIB", "product")}
    for name, bulk in (("beside_bf16x3_gemm", lambda: hip.wino_gemm_bf16x3(Uf, Vg, 6)), ("beside_fp32_gemm", lambda: hip.wino_gemm(U, Vg)),
                       ("alone", None), ("beside_bf16x3_gemm_again", lambda: hip.wino_gemm_bf16x3(Uf, Vg, 6))):
        calls = differing = 0
        where = []
        for _ in range(rounds):
            outs = []
            torch.cuda.synchronize()
            for i in range(ring):
                with torch.cuda.stream(side):
                    outs.append(hip.wino_fused(M, shape, 6, bias, None, True, ts_out=6)[2])
                if bulk is not None:
                    bulk()
            torch.cuda.synchronize()
            for o in outs:
                calls += 1
                if not torch.equal(o, ref):
                    differing += 1
                    idx = (o != ref).nonzero()
                    where.append({"values": int(idx.shape[0]), "channels": idx[:, 1].unique().tolist()[:4], "chains": (idx[:, 2] // 16).unique().tolist()[:4],
                                  "t": idx[:, 0].unique().tolist()[:8]})
        res[name] = {"fused_calls": calls, "differing": differing, "where": where[:6]}
        print("pair", res["library"][-24:], name, res[name], flush=True)
    # the codec's own shape: BOTH streams run GEMM -> k_wino_fused chains (the layers of two conv stacks side by side)
    for name, gemm in (("two_stacks_bf16x3", lambda v: hip.wino_gemm_bf16x3(Uf, v, 6)), ("two_stacks_fp32", lambda v: hip.wino_gemm(U, v))):
        def stack(v, depth=3):
            outs = []
            for _ in range(depth):
                m = gemm(v)
                v = hip.wino_fused(m, shape, 6, bias, None, True, ts_out=6)[2]
                outs.append(v)
            return outs
        want = [o.clone() for o in stack(Vg[:, :, :cols].contiguous())]
        torch.cuda.synchronize()
        calls = differing = 0
        where = []
        for _ in range(rounds):
            got = []
            torch.cuda.synchronize()
            for i in range(ring // 8):
                with torch.cuda.stream(side):
                    got.append(stack(Vg))
                got.append(stack(Vg))
            torch.cuda.synchronize()
            for outs in got:
                for d_, (o, w) in enumerate(zip(outs, want)):
                    calls += 1
                    if not torch.equal(o, w):
                        differing += 1
                        idx = (o != w).nonzero()
                        where.append({"layer": d_, "values": int(idx.shape[0]), "channels": idx[:, 1].unique().tolist()[:4],
                                      "chains": (idx[:, 2] // 16).unique().tolist()[:4]})
                        break
        res[name] = {"fused_calls": calls, "differing": differing, "where": where[:6]}
        print("pair", res["library"][-24:], name, res[name], flush=True)
    return res


def trio_leg(rounds=60, ring=160):
    """Three kernels: k_wino_fused<6,6> on one stream, the 32-chain GEMM (bf16x3 / fp32) on a second, a float64 table kernel
    (k_logistic<16>, the (f, c) flavour at 32 chains) on a third.  Packed float32 instructions execute on the float64 datapath of
    a SIMD: is the table kernel the missing partner?"""
    dev = "cuda"
    torch.manual_seed(0)
    T, C, cols, N = 36, 256, 512, 32
    U, Vg = torch.randn(T, C, C, device=dev), torch.randn(T, C, cols, device=dev)
    Uf = hip.frags_bf16x3(U)
    M = torch.randn(T, C, cols, device=dev) * 2.0
    bias = torch.randn(C, device=dev)
    shape = (N, C, 16, 16)
    rng = np.random.RandomState(0)
    Dz, Kz = 2048, 1024
    lo, hi = rng.uniform(-8, -2, Dz), rng.uniform(2, 8, Dz)
    ez_np = np.stack([np.linspace(a, b, Kz + 1)[1:-1] for a, b in zip(lo, hi)])
    ez, stepz = torch.from_numpy(ez_np).to(dev), torch.from_numpy(uniform_step(ez_np)).to(dev)
    muz = torch.from_numpy(rng.randn(N, Dz).astype(np.float32)).to(dev)
    scz = torch.from_numpy(rng.uniform(0.1, 1, (N, Dz)).astype(np.float32)).to(dev)
    symz = torch.from_numpy(rng.randint(0, Kz, (N, Dz)).astype(np.int32)).to(dev)
    stz = torch.zeros(N, dtype=torch.int32, device=dev)
    tables = lambda: hip.logistic_fc(ez, muz, scz, symz, stz, 31, 10, step=stepz)
    ref = hip.wino_fused(M, shape, 6, bias, None, True, ts_out=6)[2].clone()
    s_fused, s_tab = torch.cuda.Stream(), torch.cuda.Stream()
    res = {"library": os.environ.get("BITSWAP_HIP_LIB", "product")}
    for name, gemm in (("fused + bf16x3 gemm + f64 tables", lambda: hip.wino_gemm_bf16x3(Uf, Vg, 6)), ("fused + fp32 gemm + f64 tables", lambda: hip.wino_gemm(U, Vg)),
                       ("fused + f64 tables", None), ("fused + bf16x3 gemm + f64 tables (again)", lambda: hip.wino_gemm_bf16x3(Uf, Vg, 6))):
        calls = differing = 0
        where = []
        for _ in range(rounds):
            outs = []
            torch.cuda.synchronize()
            for i in range(ring):
                with torch.cuda.stream(s_fused):
                    outs.append(hip.wino_fused(M, shape, 6, bias, None, True, ts_out=6)[2])
                if gemm is not None:
                    gemm()
                if i % 4 == 0:
                    with torch.cuda.stream(s_tab):
                        tables()
            torch.cuda.synchronize()
            for o in outs:
                calls += 1
                if not torch.equal(o, ref):
                    differing += 1
                    idx = (o != ref).nonzero()
                    where.append({"values": int(idx.shape[0]), "channels": idx[:, 1].unique().tolist()[:4], "chains": (idx[:, 2] // 16).unique().tolist()[:4]})
        res[name] = {"fused_calls": calls, "differing": differing, "where": where[:6]}
        print("trio", res["library"][-24:], name, res[name], flush=True)
    return res


STACKS_CODE = r"""
import os, sys, json, torch
sys.path.insert(0, "__ROOT__")
from bitswap_amd import workload
from bitswap_amd.codec import BitSwapCodec, deterministic_convs
model, zend, zcen = workload.build("cifar8", "cuda", quantbits=10)
codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)      # (puts the model into its coding configuration)
B = 32
torch.manual_seed(3)
z = torch.randn(B, model.zdim_flat, device="cuda") * 1.5
inf = [model.infer(i) for i in range(1, model.nz)]
gen = [model.generate(i) for i in range(1, model.nz)]
side = torch.cuda.Stream()
def run(fns):
    with torch.no_grad(), deterministic_convs():
        return [tuple(t.contiguous() for t in f(z)) for f in fns]
ref_i, ref_g = run(inf), run(gen)
torch.cuda.synchronize()
out = {"library": os.environ.get("BITSWAP_HIP_LIB", "product"), "gemm_arith": model.gemm_arith, "stack_pairs": 0, "differing_stacks": 0, "where": []}
for rep in range(int(os.environ.get("REPRO_REPS", "1500"))):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        a = run(inf)
    g = run(gen)
    torch.cuda.synchronize()
    for name, got, want in (("infer", a, ref_i), ("generate", g, ref_g)):
        for i, ((m1, s1), (m0, s0)) in enumerate(zip(got, want)):
            out["stack_pairs"] += 1
            if not (torch.equal(m1, m0) and torch.equal(s1, s0)):
                out["differing_stacks"] += 1
                bad = ((m1 != m0) | (s1 != s0)).reshape(B, -1).any(1).nonzero().flatten().tolist()
                out["where"].append({"rep": rep, "stack": f"{name}({i + 1})", "stream": "side" if name == "infer" else "main", "chains": bad})
out["where"] = out["where"][:12]
print("RESULT " + json.dumps(out))
"""


def stacks_leg(reps=1500):
    """Two conv stacks side by side and NOTHING else (no coder kernel, no table kernel): infer(1..7) of the cifar8 model on a side stream
    while generate(1..7) runs on the main stream, 32 chains, bf16x3 and fp32 arithmetic, compared with the one-stream results."""
    import subprocess
    code = STACKS_CODE.replace("__ROOT__", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = {}
    for arith in ("bf16x3", "fp32"):
        env = dict(os.environ, BITSWAP_GEMM_ARITH=arith, BITSWAP_BF16X3_SHAPE="2", REPRO_REPS=str(reps))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        res[arith] = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-800:]}
        print("stacks", arith, json.dumps(res[arith]), flush=True)
    return res


def trail_leg(reps=120):
    """Where does a failing forked run first leave the one-stream run?  Head of every chain after every stack operation and a
    per-chain checksum of every conv stack's input and output, compared with the one-stream run of the same codec."""
    import subprocess
    code = TRAIL_CODE.replace("__ROOT__", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = {}
    for diag in ("noclaim", None):
        env = dict(os.environ, BITSWAP_GEMM_ARITH="bf16x3", BITSWAP_BF16X3_SHAPE="2", BITSWAP_FORK="1", REPRO_REPS=str(reps))
        env.pop("BITSWAP_BF16X3_DIAG", None)
        if diag:
            env["BITSWAP_BF16X3_DIAG"] = diag
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        res[diag or "claim"] = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-600:]}
        print("trail", diag or "claim", json.dumps(res[diag or "claim"]), flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--codec", action="store_true")
    ap.add_argument("--trio", action="store_true", help="three kernels: packed k_wino_fused<6,6> + GEMM + a float64 table kernel on three streams")
    ap.add_argument("--stacks", action="store_true", help="two conv stacks side by side and nothing else (BITSWAP_HIP_LIB picks the build)")
    ap.add_argument("--pair", action="store_true", help="two kernels only: k_wino_fused<6,6> on a side stream beside the bf16x3 / fp32 GEMM (BITSWAP_HIP_LIB picks the build)")
    ap.add_argument("--record", action="store_true", help="forked bf16x3 codec with every stack kernel call kept; failing runs are replayed call by call")
    ap.add_argument("--focus", action="store_true", help="codec leg only: the failing scenario with and without counted waits, 14 runs each")
    ap.add_argument("--trail", type=int, default=0, help="runs of the checksum-trail leg (0: skip)")
    ap.add_argument("--storm", type=int, default=0, help="GEMM launches of the victim hunt (0: skip)")
    ap.add_argument("--small", action="store_true", help="micro leg with the neighbours that fit beside the unclaimed shape-2 kernel (<= 32 registers)")
    a = ap.parse_args()
    out = {}
    if a.trio:
        out["trio"] = trio_leg()
        print(json.dumps(out, indent=1))
        sys.exit(0)
    if a.stacks:
        out["stacks"] = stacks_leg()
        print(json.dumps(out, indent=1))
        sys.exit(0)
    if a.pair:
        out["pair"] = pair_leg()
        print(json.dumps(out, indent=1))
        sys.exit(0)
    if a.record:
        out["record"] = record_leg(int(os.environ.get("REPRO_RECORD_REPS", "150")))
        print(json.dumps(out, indent=1))
        sys.exit(0)
    if a.trail:
        out["trail"] = trail_leg(a.trail)
    if a.storm:
        out["storm"] = storm(a.storm)
    if (not a.focus and not a.storm and not a.trail) or a.small:
        out["micro"] = micro(a.reps, small=a.small)
    if a.codec or a.focus:
        out["codec"] = codec_leg(focus=a.focus)
    print(json.dumps(out, indent=1))
