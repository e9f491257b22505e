#!/usr/bin/env python3
"""bench.py -- pixels/s (encode+decode) of the Bit-Swap compression path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--workload cifar8|imagenet4|mnist2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N            (no torchrun: spawns the N ranks itself)

One step = one 32x32 block of every chain coded in lock-step, sender AND receiver: the timed
region runs K sender block-steps followed by the K receiver block-steps that undo them, bracketed
by barrier + synchronize on both sides (max over ranks).  Each rank owns `--chains` independent
chains (weak scaling: chains never talk to each other; the only collective is the gather of the
finished bitstreams, outside the timed region).  The inputs (blocks, weights, bins, initial
stacks) are resident in HBM before the timed region starts.

Headline workload = BASELINE.json configs[1]: CIFAR-10-shaped 8-latent-layer Bit-Swap model
(reswidth 252, Z = 2048, X = 3072, K = 1024 / 256), synthetic data and seeded random-init weights
(no datasets/checkpoints offline).  At N = 1 the same JSON line also carries, under `extra`, the
ImageNet32 nz=4 shape north_star quotes its target on (configs[2]) and the reference's own 100-chain
shape (100 "experiments"), and the opt-in 64-state stream format at 800 and at 13 chains, each measured by the same
procedure with fewer steps.

Extra objects on the JSON line: `roofline` for the dominant hot-path kernel (the fused
logistic-CDF -> integer-table kernel, decode flavour), timed with HIP events on its launch stream in an
EXCLUSIVE single-stream pass right after the timed region (inside the pipeline a launch shares the chip with
the other chain group's GEMMs; that time-shared figure is reported next to it).  The headline bound is the
one the kernel sits on -- VALU issue, `frac` = issue slots / time / peak -- with the HBM figures beside it
computed on bytes that actually move (batched algorithmic bytes, PMC counter bytes): every fraction <= 1.
`roofline.mfma` is the same for the kernel that dominates the rocprof summary (the Winograd-domain batched
GEMM of the conv stacks, fp32 MFMA), `path_frac` the whole-path figure of SURVEY.md 8(d).  `cpu_baseline`: the
oracle (C restatement of the reference, libm CDF) + the same conv stacks on the host cores, timed on a
bounded sample of the same workload, next to the committed measurement of the reference's own Python
path (profiles/r02_ref_cpu_baseline.json, tools/ref_cpu_baseline.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue model of k_logistic (the bound it actually runs into): 256 CUs x 4 SIMDs, one 64-lane VALU instruction
# occupies a 16-lane SIMD for 4 cycles (v_rcp_f64: 16), 2.4 GHz nominal -> 614.4 G wave-instructions / s
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4
# issue slots per (chain, dim) row of the K = 1024 decode-flavour kernel, from `tools/isa_count.py` ("largest loop") on
# the shipped code object: VALU instructions of the per-row loop + 3 extra slots per quarter-rate v_rcp_f64.  Under
# sustained float64 load the chip runs at about 4/4.9 of the nominal clock (tools/probes/instr_rate.hip), so 0.82 here is the
# practical ceiling
VALU_SLOTS_PER_ROW = {2: 381, 1: 648}   # CDF spec 2 (uniform bins, BS_LAYOUT_PIVOT hand-off; 403 with whole rows) / spec 1
# float64 flops of one row (64 lanes x [2 per fma + 1 per add/mul/rcp] in that loop): SURVEY 8(d) asks for the FP64
# utilisation next to the HBM figure.  Vector FP64 peak 78.6 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
FP64_FLOPS_PER_ROW = {2: 353 * 64, 1: 699 * 64}
FP64_PEAK_TFLOPS = 78.6
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense, MI355X_MICROARCH.md (155 measured)

TITLES = {"mnist2": "MNIST-shaped 2-latent-layer", "cifar8": "CIFAR-10-shaped 8-latent-layer",
          "imagenet4": "ImageNet32-shaped 4-latent-layer", "imagenetcrop4": "ImageNet-crop-shaped 4-latent-layer"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cifar8", choices=["mnist2", "cifar8", "imagenet4", "imagenetcrop4"])
    ap.add_argument("--chains", type=int, default=1000,
                    help="independent chains (rANS streams) per GPU; the reference runs 100 'experiments' one block at a time, "
                         "an MI355X wants a few hundred in lock-step (conv efficiency grows with the batch, DESIGN.md 6)")
    ap.add_argument("--quantbits", type=int, default=10)
    ap.add_argument("--bitswap", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (no imagenet4 / 100-chain sub-results)")
    ap.add_argument("--cpu-blocks", type=int, default=20, help="blocks per chain in the CPU baseline sample")
    ap.add_argument("--no-timeline", action="store_true", help="skip per-kernel events (roofline becomes null)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the exclusive roofline passes after the timed region")
    ap.add_argument("--regime", default=None, choices=["lowrate"],
                    help="lowrate: the calibrated synthetic model coding its own samples at a trained model's rate (workload.py)")
    ap.add_argument("--groups", type=int, default=2,
                    help="chain groups per GPU on separate HIP streams (serial rANS of one group under the convs of another)")
    ap.add_argument("--tables-on", default="bulk", choices=["bulk", "serial"],
                    help="stream of the table kernels when groups > 1 (see BitSwapCodec.tables_on)")
    ap.add_argument("--cdf-spec", type=int, default=2, choices=[1, 2])
    ap.add_argument("--no-graphs", action="store_true", help="never replay the block step from a hipGraph (single-stream runs)")
    ap.add_argument("--format", default="reference", choices=["reference", "wave64"],
                    help="reference: the reference's single-state word stream (default, the headline); wave64: the opt-in "
                         "64-state format -- table + coding step fused in one launch, one stream per chain group")
    return ap.parse_args()


def cpu_baseline(args, name):
    """Oracle (kind 'port') + conv stacks on the host cores, bounded sample of the same workload."""
    import oracle as O
    from oracle.backend import OracleBackend
    from bitswap_amd import workload
    from bitswap_amd.codec import BitSwapCodec

    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 16))
    torch.set_num_threads(threads)
    model, zend, zcen = workload.build(name, "cpu", quantbits=args.quantbits)
    B, n = threads, args.cpu_blocks
    images = workload.synthetic_blocks(B * n, model.xs, seed=11).view(B, n, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=args.quantbits, bitswap=bool(args.bitswap),
                         backend=OracleBackend(O.MODE_LIBM, threads=threads))
    state = codec.new_states(B, n)
    t0 = time.perf_counter()
    for xi in range(n):
        codec.encode_block(state, images[:, xi])
    outs = [codec.decode_block(state) for _ in range(n)]
    dt = time.perf_counter() - t0
    ok = all(torch.equal(outs[n - 1 - xi], images[:, xi]) for xi in range(n))
    out = {"value": B * n * 1024 / dt, "unit": "pixels/s", "cores": threads, "kind": "port",
           "sample": f"{B} chains x {n} block(s) of {name}, sender+receiver, oracle C (libm CDF) + torch-CPU convs, "
                     f"{dt:.1f} s, lossless={ok}"}
    out["reference_python"] = reference_python_baseline(name)
    return out


def reference_python_baseline(name):
    """The reference's OWN Python path (mnist_compress.py:164-358 replayed around the imported reference classes,
    tools/ref_cpu_baseline.py).  Same-host when the reference is present on this machine (/root/reference or
    $BITSWAP_REFERENCE): measured live on a bounded sample.  The project's GPU boxes carry no copy of the reference: there the
    committed measurement from the build container rides along, and `host` says so."""
    import subprocess
    ref = os.environ.get("BITSWAP_REFERENCE", "/root/reference")
    if os.path.isdir(ref) and name in ("cifar8", "imagenet4", "mnist2"):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_cpu_baseline.py"), "--config", name,
                                "--where", "same host as the GPU run (measured live by bench.py)"],
                               capture_output=True, text=True, timeout=240, env=dict(os.environ, BITSWAP_REFERENCE=ref))
            if r.returncode == 0:
                d = json.loads(r.stdout[r.stdout.index("{"):])
                d["same_host"] = True
                return d
        except Exception:
            pass
    rp = os.path.join(ROOT, "profiles", "r02_ref_cpu_baseline.json")
    try:
        d = json.load(open(rp))
        d["same_host"] = False
        d["host"] += " -- the reference is not present on this host, committed measurement quoted"
        return d
    except Exception:
        return None


def _valu_busy(spec):
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "valu_busy.json")))["kernels"]
        return d["k_logistic<16,float,decode,uniform>" if spec == 2 else "k_logistic<16,float,decode,generic>"]["valu_busy"]
    except Exception:
        return None


def algorithmic_bytes_per_block(codec):
    """SURVEY.md 8(d), one direction: per z-table op (2nz-1 of them) Z(K-1)8 + 2Z4 + Z4; x-op 2X4 + X; prior Z(8+4)."""
    Z, X, K, nz = codec.Z, codec.X, codec.K, codec.codecs[0].nz
    return (2 * nz - 1) * (Z * (K - 1) * 8 + 2 * Z * 4 + Z * 4) + (2 * X * 4 + X) + Z * 12


def exclusive_pass(codec, states, block, rkey, tl, dev):
    """The roofline kernel with the chip to itself: ONE chain group codes one block (sender, then the receiver that undoes
    it) on a single stream -- no other group's GEMMs next to the table launches -- with the same Timeline events.  Runs
    after the timed region; the state ends where it started.  -> (seconds, launches) of `rkey`, or None."""
    c, st = codec.codecs[0], states[0]
    saved = (c.bulk, c.serial, c.use_graphs)
    try:
        torch.cuda.synchronize()
        c.bulk = c.serial = None
        c.use_graphs = False
        tl.reset()
        sl = codec.split(block.shape[0])[0]
        c.encode_block(st, block[sl, 0])
        back = c.decode_block(st)
        torch.cuda.synchronize()
        if not torch.equal(back, block[sl, 0]):
            return None
        t = tl.totals().get(rkey)
        return t if t and t[1] else None
    except Exception:
        return None
    finally:
        c.bulk, c.serial, c.use_graphs = saved
        tl.reset()


def gemm_roofline(model, chains, dev, warm=100, reps=100):
    """The kernel that dominates the rocprof summary is not on the entropy path: the Winograd-domain batched GEMM of the
    conv stacks (bs_wino_gemm_f32, fp32 MFMA).  Its own roofline, at the shape one chain group launches most often
    (36 transform positions x [C x C] x [C x 16 tiles per block]), HIP events around an exclusive back-to-back loop -- 100
    launches of warm-up first: the clock of an MI355X follows the load of the last tens of milliseconds, a handful of launches
    measures the clock history of whatever ran before (profiles/r03b vs r03f: 93 vs 117 TFLOP/s for the same cycle count)."""
    try:
        from bitswap_amd import hip
        if not (getattr(model, "fused", False) and getattr(model, "own_gemm", False)):
            return None
        C = int(model._cp)
        cols = int(chains) * 16
        U = torch.randn(36, C, C, device=dev)
        V = torch.randn(36, C, cols, device=dev)
        if not hip.wino_gemm_supported(U, V):
            return None
        out = torch.empty(36, C, cols, device=dev)
        for _ in range(warm):
            hip.wino_gemm(U, V, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            hip.wino_gemm(U, V, out=out)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps * 1e-3
        fl = 2.0 * 36 * C * C * cols
        return {"kernel": "k_wino_gemm<4,2> (bs_wino_gemm_f32: v_mfma_f32_32x32x2_f32, persistent balanced tiles, LDS-DMA staging)", "bound": "mfma",
                "shape": f"T36 x [{C}x{C}] x [{C}x{cols}]", "achieved": round(fl / t / 1e12, 1), "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(fl / t / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "avg_launch_ms": round(t * 1e3, 4),
                "launches": reps, "hbm_bytes_per_launch": int(4 * 36 * C * cols * 2 + 4 * 36 * C * C),
                "timing": f"exclusive: HIP events around {reps} back-to-back launches on the current stream after {warm} of warm-up"}
    except Exception as e:
        return {"error": repr(e)}


def run_workload(args, name, B, groups, K, W, dev, rank, world, dist, want_gather=False, want_roofline=True, regime=None):
    """One measurement by the contract's procedure.  Returns a dict (timings are max over ranks)."""
    from bitswap_amd import workload
    from bitswap_amd.codec import GroupedCodec, Timeline, initial_states

    model, zend, zcen = workload.build(name, dev, quantbits=args.quantbits, regime=regime)
    n = K + W
    if regime == "lowrate":      # blocks from the calibrated model's own generative path (workload.lowrate_blocks)
        images = workload.lowrate_blocks(model, B * n, seed=1000 + rank, batch=400).view(B, n, -1).to(torch.int32).to(dev)
    else:
        images = workload.synthetic_blocks(B * n, model.xs, seed=1000 + rank).view(B, n, -1).to(torch.int32).to(dev)
    tl = Timeline(enabled=not args.no_timeline)
    backend = None
    if args.format == "wave64":
        from bitswap_amd.codec import Hip64Backend
        backend = Hip64Backend(dev)          # groups > 1: one stream per group, the groups overlap each other
    codec = GroupedCodec(model, zend, zcen, groups=groups, quantbits=args.quantbits, bitswap=bool(args.bitswap),
                         timeline=tl, cdf_spec=args.cdf_spec, backend=backend)
    for c in codec.codecs:
        c.tables_on = args.tables_on
        if args.no_graphs:
            c.use_graphs = False
    init = initial_states(B, 10000, seed=100 + rank)
    states = codec.new_states(B, n, states=init)
    rest_lens = [torch.zeros_like(st.len) for st in states]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W sender steps, undone by W receiver steps (first block also records restbits)
    if W:
        codec.encode_blocks(states, images[:, :W], rest_lens)
        codec.decode_blocks(states, W)
    codec.check(states, "warmup")
    if len(codec.codecs) == 1 and W:       # one stream: the launch-bound block step is replayed from a hipGraph;
        for c, st in zip(codec.codecs, states):   # capture it here, not inside the timed region
            c.prepare_graphs(st)
    tl.reset()

    barrier()
    t0 = time.perf_counter()
    codec.encode_blocks(states, images[:, W:], rest_lens if W == 0 else None)
    len_sent = torch.cat([st.len for st in states]).clone()
    # the finished bitstreams (what a sender would ship): device-side snapshot, gathered after the clock stops
    sent = ([(st.stack.clone(), st.len.clone(), st.head.clone()) for st in states]
            if (want_gather and args.format == "reference") else None)
    decoded = codec.decode_blocks(states, K)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- verification (outside the timed region): lossless + stream fully unwound
    codec.check(states, "bench")
    ok = bool(torch.equal(decoded, images[:, W:]))
    if args.format == "wave64":
        from bitswap_amd.hip import split_state
        ok = ok and codec.to_lists(states) == [split_state(s) for s in init]
        bits = (len_sent.cpu().numpy().astype(np.int64) - np.array([len(s) - 64 for s in init])) * 32
    else:
        ok = ok and codec.to_lists(states) == init
        bits = (len_sent.cpu().numpy().astype(np.int64) - np.array([len(s) - 1 for s in init])) * 32
    bpd = float(bits.sum()) / (B * K * codec.X)

    # ---- the path's only exchange (not timed): gather of the finished bitstreams to rank 0 and two scalars
    # for bits/dim -- RCCL over xGMI when world > 1 (bitswap_amd/dist.py), a local no-op otherwise
    gather = None
    if sent is not None:
        from bitswap_amd import dist as bdist
        try:
            streams = []
            for stack, ln, hd in sent:
                stack, ln, hd = stack.cpu().numpy().view(np.uint32), ln.cpu().numpy(), hd.cpu().numpy().view(np.uint64)
                for b in range(stack.shape[0]):   # stream = stack words + the 64-bit head as two words (demo container order)
                    streams.append(np.concatenate([stack[b, : ln[b]], np.array([hd[b] & 0xffffffff, hd[b] >> 32], dtype=np.uint32)]))
            mine = [rank + world * c for c in range(B)]            # global chain ids, round-robin like shard_chains()
            tg = time.perf_counter()
            got = bdist.gather_streams(streams, mine, world * B)
            tg = time.perf_counter() - tg
            if rank == 0:
                words = int(sum(len(a) for a in got))
                gather = {"chains": len(got), "bytes": 4 * words, "ms": round(tg * 1e3, 2),
                          "complete": all(a is not None for a in got),
                          "own_streams_intact": all(np.array_equal(got[c], a) for c, a in zip(mine, streams))}
        except Exception as e:   # never lose the bench line to the reporting exchange
            gather = {"error": repr(e)}
    if dist is not None:
        tot = torch.tensor([float(bits.sum()), float(B * K * codec.X), float(ok)], device=dev, dtype=torch.float64)
        dist.all_reduce(tot)
        bpd = float(tot[0] / tot[1])
        ok = bool(tot[2].item() == world)

    totals = tl.totals()
    roof = None
    # the roofline kernel: the table kernel of the reference format; in the 64-state format the table is built inside
    # the fused pop launch, which is then the kernel that carries the algorithmic bytes
    rkey = "pop_z" if args.format == "wave64" else "tables_z"
    if args.format == "wave64":
        totals.pop("tables_z", None), totals.pop("tables_x", None)      # lazy handles: nothing is launched there
    if rkey in totals and totals[rkey][1] and want_roofline:
        shared_sec, shared_cnt = totals[rkey]
        excl = exclusive_pass(codec, states, images[:, W:W + 1], rkey, tl, dev)
        Kb, Z = codec.K, codec.Z
        rows = B * Z / max(1, groups)                     # rows one launch processes (one chain group)
        sec, cnt = excl if excl else (shared_sec, shared_cnt)
        avg = sec / cnt
        spec = args.cdf_spec if any(s is not None for s in codec.codecs[0].zstep) else 1
        from bitswap_amd import hip as _hip
        be = codec.codecs[0].backend
        pivot = bool(spec == 2 and args.format == "reference" and hasattr(be, "table_layout")
                     and be.table_layout(Kb, True, Z, int(rows // Z)) == _hip.LAYOUT_PIVOT)
        slots = VALU_SLOTS_PER_ROW[spec] if (Kb == 1024 and args.format == "reference") else None
        if slots is not None and spec == 2 and not pivot:
            slots = 403                                   # whole-row hand-off (BITSWAP_PIVOT=0)
        kname = (f"k_layer64<16,float,{'uniform' if spec == 2 else 'generic'},pop> (logistic CDF -> integer table -> rANS pop in "
                 f"one launch, rows in registers, CDF spec {spec})" if args.format == "wave64" else
                 f"k_logistic<16,float,{'pivot' if pivot else 'decode'},{'uniform' if spec == 2 else 'generic'}> (fused logistic CDF -> "
                 f"integer table -> {'64 cumulative values per row for bs_rans_pop_pivot' if pivot else 'cdf rows'}, CDF spec {spec})")
        # --- HBM side, on bytes that move.  SURVEY 8(d) prices a z-row at (K-1)*8 + 12 B because it counts the [Z, K-1]
        # float64 endpoint table once per BLOCK; a launch over `chains` blocks reads that table from HBM once (the rest are
        # L2 hits), so the batched algorithmic bytes are table + 12 B/row + the hand-off to the pop kernel (512 B/row of
        # cumulative values with BS_LAYOUT_PIVOT; a whole 4 (K + 64) B row before: counted by the PMC figure)
        table_bytes = Z * (Kb - 1) * 8
        alg_batched = int(table_bytes + rows * (12 + (512 if pivot else 0)))
        survey_alg = int(rows * ((Kb - 1) * 8 + 12))
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and args.format == "reference":
            try:
                per_row = json.load(open(tp)).get(name, {}).get("k_logistic_pivot_bytes_per_row" if pivot else "k_logistic_decode_bytes_per_row")
                traffic = None if per_row is None else int(per_row * rows)   # PMC bytes/row x rows of one launch
            except Exception:
                traffic = None
        a_block = algorithmic_bytes_per_block(codec)
        path = 2.0 * a_block * world * B * K / dt / 1e9
        hbm = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "alg_bytes_per_launch": alg_batched, "achieved": round(alg_batched / avg / 1e9, 1),
               "frac": round(alg_batched / avg / 1e9 / HBM_PEAK_GBPS, 5),
               "traffic_bytes_per_launch": traffic,
               "traffic_achieved": None if traffic is None else round(traffic / avg / 1e9, 1),
               "traffic_frac": None if traffic is None else round(traffic / avg / 1e9 / HBM_PEAK_GBPS, 4),
               "note": "alg = endpoint table once per launch + 12 B/row (mu, scale, symbol)"
                       + (" + the 512 B/row hand-off of cumulative values to bs_rans_pop_pivot" if pivot else "")
                       + "; traffic = PMC FETCH x2 + WRITE (profiles/traffic.json)"
                       + ("" if pivot else ", 98 % of it the cdf-row hand-off to k_rans_pop_wave")
                       + f"; SURVEY 8(d)'s per-block count would be {survey_alg} B per launch, of which all but the first table "
                         "pass are L2 hits"}
        if slots is not None:
            ach = rows * slots / avg / 1e9
            roof = {"kernel": kname, "bound": "valu_issue", "achieved": round(ach, 1), "peak": round(VALU_PEAK_GINSTR, 1),
                    "unit": "Ginstr/s", "frac": round(ach / VALU_PEAK_GINSTR, 4), "slots_per_row": slots,
                    # measured by SQ counters (profiles/valu_busy.json, tools/pmc_valu.sh): share of the SIMD cycles the
                    # VALU pipe is busy while the kernel runs, at the clock the chip actually holds under float64 load
                    "valu_busy_pmc": _valu_busy(spec)}
        else:       # no issue model for this kernel shape: the counter-side HBM figure is the headline
            roof = {"kernel": kname, "bound": "hbm", "achieved": hbm["traffic_achieved"] or hbm["achieved"],
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": hbm["traffic_frac"] if hbm["traffic_frac"] is not None else hbm["frac"]}
        roof.update({"traffic": traffic, "launches": cnt, "avg_launch_ms": round(avg * 1e3, 4),
                     "timing": ("exclusive: HIP events on the launch stream, single-stream pass of one chain group after the "
                                "timed region" if excl else "time-shared: HIP events inside the timed region"),
                     "avg_launch_ms_in_pipeline": round(shared_sec / shared_cnt * 1e3, 4), "rows_per_launch": int(rows),
                     "hbm": hbm,
                     "fp64": None if slots is None else {
                         "flops_per_row": FP64_FLOPS_PER_ROW[spec],
                         "achieved_TFLOPs": round(rows * FP64_FLOPS_PER_ROW[spec] / avg / 1e12, 2), "peak_TFLOPs": FP64_PEAK_TFLOPS,
                         "frac": round(rows * FP64_FLOPS_PER_ROW[spec] / avg / 1e12 / FP64_PEAK_TFLOPS, 4)},
                     # whole path, SURVEY.md 8(d): 2 * A_block * n_blocks / (t_sender + t_receiver) against the HBM peak
                     "path_alg_bytes_per_block": a_block, "path_achieved": round(path / world, 1),
                     "path_frac": round(path / world / HBM_PEAK_GBPS, 4),
                     "mfma": gemm_roofline(model, B // max(1, groups), dev)})
    breakdown = {k: round(v[0] / dt, 4) for k, v in sorted(totals.items())} if totals else None
    res = {"workload": name, "chains_per_gpu": B, "chain_groups": groups, "steps": K, "warmup": W,
           "value": world * B * K * 1024 / dt, "ms_per_step": dt / K * 1e3, "lossless": ok, "bits_per_dim": bpd,
           "stream_time_fraction": breakdown, "roofline": roof, "stream_gather": gather,
           "codec": codec, "model": model}
    return res


def extra_in_child(args, wn, ch, gr, fmt, reg, steps, warmup):
    """One `extra` sub-result measured the way the headline is: the first workload of a fresh process.  (Measured in the
    headline's process, after its buffers came and went, ImageNet32 nz=4 at 1000 chains reads 6.3 Mpixel/s against 6.8 on
    its own -- profiles/r03z: device memory handed out late in a process's life is more fragmented.)  -> dict or None."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", wn, "--chains", str(ch), "--groups", str(gr), "--format", fmt,
           "--steps", str(steps), "--warmup", str(warmup), "--quantbits", str(args.quantbits), "--bitswap", str(args.bitswap),
           "--cdf-spec", str(args.cdf_spec), "--no-extra", "--no-cpu-baseline", "--no-roofline"]
    if reg:
        cmd += ["--regime", reg]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"workload": wn, "chains_per_gpu": ch, "chain_groups": gr, "steps": d["steps"], "warmup": d["warmup"],
                "value": d["value"], "ms_per_step": d["ms_per_step"], "lossless": d["lossless"], "bits_per_dim": d["bits_per_dim"],
                "stream_time_fraction": d.get("stream_time_fraction"), "roofline": None,
                "process": "own process (python bench.py --no-extra ...), like the headline"}
    except Exception:
        return None


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the coding path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # BENCH_DIST_BACKEND=gloo lets the control flow be exercised with several ranks on ONE device
        # (RCCL refuses two ranks per GPU); the driver never sets it
        dist.init_process_group(os.environ.get("BENCH_DIST_BACKEND") or "nccl", timeout=datetime.timedelta(seconds=300))

    name = args.workload
    r = run_workload(args, name, args.chains, args.groups, args.steps, args.warmup, dev, rank, world, dist, want_gather=True,
                     want_roofline=not args.no_roofline, regime=args.regime)
    codec, model = r.pop("codec"), r.pop("model")
    gemm = ("bs_wino_gemm_f32 (own fp32 MFMA kernel, every product: results independent of chains per call)"
            if getattr(model, "own_gemm", False) else f"BLAS backend {model.gemm_backend}")
    conv_path = (f"{model.conv_algo} (fp32; ResNet/head convs as transform-domain batched GEMMs on {gemm}; 3x3 input convs: "
                 f"{'bs_conv3_wino_f32' if getattr(model, 'fused_inputs', False) else 'MIOpen'}; 5x5 input conv: "
                 f"{'Winograd domain (bs_small_k_gemm_f32)' if getattr(model, 'wino_in5', False) else 'MIOpen'})"
                 if getattr(model, "fused", False) else "torch modules")
    Z, X = codec.Z, codec.X
    del codec, model

    extra = None
    if world == 1 and not args.no_extra and name == "cifar8" and args.format == "reference":
        # driver-visible numbers for the other shapes DESIGN.md quotes: north_star's target config and the reference's
        # 100-experiment shape, same procedure, fewer steps
        extra = []
        ks, ws = min(args.steps, 6), min(args.warmup, 1)
        import copy
        import gc
        for (wn, ch, gr, fmt, reg) in (("imagenet4", 1000, 2, "reference", None), ("cifar8", 100, 2, "reference", None),
                                       ("cifar8", 1000, 2, "reference", "lowrate"),
                                       ("cifar8", 800, 2, "wave64", None), ("cifar8", 13, 1, "wave64", None)):
            gc.collect()                       # the previous workload's model, bins and states go before the next is built
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            try:
                e = extra_in_child(args, wn, ch, gr, fmt, reg, ks, max(ws, 2))     # a fresh process, like the headline's
                if e is None:
                    a2 = copy.copy(args)
                    a2.format = fmt
                    e = run_workload(a2, wn, ch, gr, ks, max(ws, 2), dev, rank, world, dist, want_roofline=False, regime=reg)
                    e.pop("codec"), e.pop("model"), e.pop("stream_gather")
                    e["process"] = "in the headline's process (a later workload in one process measures up to 8 % low)"
                e["value"], e["ms_per_step"] = round(e["value"], 1), round(e["ms_per_step"], 3)
                e["bits_per_dim"] = round(e["bits_per_dim"], 4)
                e["stream_format"] = fmt
                e["regime"] = reg or "random-init weights, unrelated synthetic blocks"
                e["config"] = (f"{TITLES[wn]} {'Bit-Swap' if args.bitswap else 'BB-ANS'}, {ch} chains / {gr} group(s)"
                               + (", calibrated low-rate regime (workload.calibrate_lowrate: latent scales at the 0.1 clamp, pixel "
                                  "scale 0.0035, blocks drawn from the model's own generative path)" if reg == "lowrate" else "")
                               + (", opt-in 64-state stream format (not the reference's word stream)" if fmt == "wave64" else ""))
                extra.append(e)
            except Exception as ex:   # a sub-result never costs the headline
                extra.append({"workload": wn, "chains_per_gpu": ch, "stream_format": fmt, "regime": reg, "error": repr(ex)})

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    cpu = None
    if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
        try:
            cpu = cpu_baseline(args, name)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            cpu = {"value": None, "unit": "pixels/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    out = {
        "metric": "pixels/s (encode+decode)", "value": round(r["value"], 1), "unit": "pixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": TITLES[name] + f" {'Bit-Swap' if args.bitswap else 'BB-ANS'}, batch of 32x32 blocks",
                   "chains_per_gpu": args.chains, "chain_groups": args.groups, "blocks_per_chain": args.steps,
                   "quantbits": args.quantbits, "ansbits": 31, "cdf_spec": args.cdf_spec, "stream_format": args.format,
                   "latent_dims": Z, "pixel_dims": X, "conv_dtype": "f32", "conv_path": conv_path,
                   "weights": "seeded random init (no checkpoints offline)"
                              + (", calibrated to the low-rate regime (workload.calibrate_lowrate)" if args.regime else "")},
        "rccl_ranks": (world if (world > 1 and (os.environ.get("BENCH_DIST_BACKEND") or "nccl") == "nccl") else 0),
        "lossless": r["lossless"], "bits_per_dim": round(r["bits_per_dim"], 4),
        "stream_time_fraction": r["stream_time_fraction"],
        "roofline": r["roofline"], "cpu_baseline": cpu, "stream_gather": r["stream_gather"], "extra": extra,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _spawned(rank, args, world, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    main(args)


if __name__ == "__main__":
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched without torchrun: start the N ranks ourselves, one process per GPU -- never measure one GPU and call
        # it N (with BENCH_DIST_BACKEND=gloo the ranks may share a device: control-flow tests only)
        import socket
        import torch.multiprocessing as mp
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not os.environ.get("BENCH_DIST_BACKEND"):
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible")
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_spawned, args=(args, args.gpus, port), nprocs=args.gpus, join=True)
    else:
        if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and "WORLD_SIZE" in os.environ:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: using WORLD_SIZE", file=sys.stderr)
        main(args)
