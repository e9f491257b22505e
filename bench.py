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
# (round 5: tools/isa_count.py --blocks, the blocks one row executes -- spec 3's loop holds two arms, a row takes one)
# (round 6, spec 4 = blocks of 8 + one Newton correction per quotient: row head 38 + batch arm 199 + integer tail 64 + store 6)
VALU_SLOTS_PER_ROW = {4: 307, 3: 268, 2: 371, 1: 643}   # CDF spec 4 / 3 / 2 (uniform bins, BS_LAYOUT_PIVOT hand-off; +29 with whole rows) / spec 1
WHOLE_ROW_EXTRA_SLOTS = 29
# float64 flops of one row (64 lanes x [2 per fma + 1 per add/mul/rcp] in that loop): SURVEY 8(d) asks for the FP64
# utilisation next to the HBM figure.  Vector FP64 peak 78.6 TFLOP/s (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
FP64_FLOPS_PER_ROW = {4: 301 * 64, 3: 229 * 64, 2: 362 * 64, 1: 763 * 64}
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
    ap.add_argument("--extra", default="core", choices=["core", "all", "none"],
                    help="sub-results measured in child processes at N = 1: core (default) = the six shapes the record needs "
                         "(north_star's ImageNet32 shape at 1000 and at 100 chains, configs[1] at 100 / 50 / 25 / 13 chains: the "
                         "predicted strong-scaling curve); all = every shape DESIGN.md quotes (twenty children, ~2 more minutes)")
    ap.add_argument("--full-record", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the FULL record goes (every sub-result, every roofline annotation); stdout carries only the "
                         "compact headline line, < 6 KB (the driver keeps 8 KB of stdout)")
    ap.add_argument("--cpu-blocks", type=int, default=20, help="blocks per chain in the CPU baseline sample")
    ap.add_argument("--no-timeline", action="store_true", help="skip per-kernel events (roofline becomes null)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the exclusive roofline passes after the timed region")
    ap.add_argument("--regime", default=None, choices=["lowrate"],
                    help="lowrate: the calibrated synthetic model coding its own samples at a trained model's rate (workload.py)")
    ap.add_argument("--groups", type=int, default=0,
                    help="chain groups per GPU on separate HIP streams (serial rANS of one group under the convs of another); "
                         "0 = auto: 2 above 128 chains per GPU, else 1 (few chains: ONE group whose forked block step is replayed "
                         "from a hipGraph -- 100 chains: 24.0 ms per step against 24.5 in two groups and 34.6 in three, "
                         "profiles/r04e: concurrent graph replays cost each other more than the overlap returns)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (the driver's default): every rank codes its own --chains chains.  strong: --total-chains chains "
                         "IN TOTAL -- the reference's 100 experiments / 100 crop images, BASELINE configs 4 and 5 -- sharded over "
                         "the ranks by bitswap_amd.dist.shard_chains (round-robin; LPT by block count for imagenetcrop4)")
    ap.add_argument("--total-chains", type=int, default=100, help="chains in total over all ranks (--scaling strong)")
    ap.add_argument("--tables-on", default="bulk", choices=["bulk", "serial"],
                    help="stream of the table kernels when groups > 1 (see BitSwapCodec.tables_on)")
    ap.add_argument("--cdf-spec", type=int, default=None, choices=[1, 2, 3, 4], help="default: bitswap_amd.meta.DEFAULT_CDF_SPEC")
    ap.add_argument("--nn-batch", type=int, default=0,
                    help="EXPERIMENT (VERDICT r5 #6, cache-resident layer chunks): run every conv stack over column chunks of this "
                         "many chains, one after the other (Model.nn_batch: fixed-shape micro-batches), so that the Winograd operands "
                         "V + M of a chunk can stay in the 256 MiB Infinity Cache between producer and consumer; 0 = one launch over "
                         "all the chains of a group (the default).  Same bits either way: every conv kernel is batch-invariant")
    ap.add_argument("--no-graphs", action="store_true", help="never replay the block step from a hipGraph (single-stream runs)")
    ap.add_argument("--format", default="reference", choices=["reference", "wave64"],
                    help="reference: the reference's single-state word stream (default, the headline); wave64: the opt-in "
                         "64-state format -- table + coding step fused in one launch, one stream per chain group")
    a = ap.parse_args()
    if a.cdf_spec is None:
        from bitswap_amd.meta import DEFAULT_CDF_SPEC
        a.cdf_spec = DEFAULT_CDF_SPEC
    return a


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
    import platform
    out = {"value": B * n * 1024 / dt, "unit": "pixels/s", "cores": threads, "kind": "port",
           "host": f"the host of this GPU run ({platform.node()}, {cores} hardware threads, {threads} used)", "same_host": True,
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


def _valu_busy(kernel_key):
    """SQ-counter VALU-busy share of the kernel that was actually timed (profiles/valu_busy.json, tools/pmc_valu.sh), keyed by
    its flavour -- e.g. "k_logistic<16,float,pivot,uniform>"; None when no counter run of THAT flavour is on file (round 3
    printed the whole-row flavour's figure under the pivot kernel's name)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "valu_busy.json")))
        k = d["kernels"].get(kernel_key)
        return None if k is None else {"valu_busy": k["valu_busy"], "source": d.get("source")}
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


BF16_MFMA_PEAK_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_bf16, dense, MI355X_MICROARCH.md (sustained on limb data, one wavefront per
#                                    SIMD, nothing but MFMAs in the loop: 1,782 -- tools/probes/mfma_rate.hip, profiles/r06b_mfma_rate.txt)


def gemm_roofline(model, chains, dev, warm=100, reps=100):
    """The kernel that dominates the rocprof summary is not on the entropy path: the Winograd-domain batched GEMM of the
    conv stacks.  Its own roofline, at the shape one chain group launches most often (36 transform positions x [C x C] x
    [C x 16 tiles per block]), for the kernel the model actually runs: bs_wino_gemm_bf16x3 (default since round 6: six bf16 limb
    products per float32 product -- `achieved` counts the limb products the matrix pipe executes, against the dense bf16 peak;
    `fp32_equivalent_TFLOPs` the float32 products they stand for) or bs_wino_gemm_f32 (fp32 MFMA).  HIP events around an
    exclusive back-to-back loop -- 100 launches of warm-up first: the clock of an MI355X follows the load of the last tens of
    milliseconds (profiles/r03b vs r03f: 93 vs 117 TFLOP/s for the same cycle count)."""
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
        arith = getattr(model, "gemm_arith", "fp32") if getattr(model, "_ufrags", None) else "fp32"
        if arith != "fp32":
            Uf, nprod = hip.frags_bf16x3(U), (9 if arith == "bf16x3x9" else 6)
            fn = lambda: hip.wino_gemm_bf16x3(Uf, V, nprod, out=out)
        else:
            nprod, fn = 1, (lambda: hip.wino_gemm(U, V, out=out))
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps * 1e-3
        fl = 2.0 * 36 * C * C * cols
        peak = MFMA_F32_PEAK_TFLOPS if arith == "fp32" else BF16_MFMA_PEAK_TFLOPS
        kern = ("k_wino_gemm<4,2> (bs_wino_gemm_f32: v_mfma_f32_32x32x2_f32, persistent balanced tiles, LDS-DMA staging)" if arith == "fp32" else
                f"k_wino_gemm_bf16x3_wsp<{nprod}> (bs_wino_gemm_bf16x3: {nprod} v_mfma_f32_32x32x16_bf16 limb products per float32 product, one "
                "multiplying wavefront per SIMD, the operand split on wavefronts of its own; at this size workgroups of two consecutive "
                "256 x 128 units -- the pipeline's choice, alone a few percent slower than one workgroup per CU)")
        return {"kernel": kern, "bound": "mfma", "arith": arith,
                "shape": f"T36 x [{C}x{C}] x [{C}x{cols}]", "achieved": round(nprod * fl / t / 1e12, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(nprod * fl / t / 1e12 / peak, 4), "fp32_equivalent_TFLOPs": round(fl / t / 1e12, 1),
                "avg_launch_ms": round(t * 1e3, 4),
                "launches": reps, "hbm_bytes_per_launch": int(4 * 36 * C * cols * 2 + 4 * 36 * C * C),
                "timing": f"exclusive: HIP events around {reps} back-to-back launches on the current stream after {warm} of warm-up"}
    except Exception as e:
        return {"error": repr(e)}


def strong_plan(total, world, name, steps, seed=100):
    """--scaling strong: which chains every rank codes, and how long they are.  -> per rank (chain ids, blocks per chain).
    Equal chains (the reference's 100 experiments, mnist_compress.py:147-161: configs 2, 3, 5): round-robin, `steps` blocks
    each.  imagenetcrop4 (config 4, imagenetcrop_compress.py:279-300: one image = one chain): image sizes H, W ~ U{256..512}
    cropped to multiples of 32 as SURVEY 8(d) has them -- 64..256 blocks -- scaled so that a 512 x 512 image is `steps`
    blocks long, sharded longest-processing-time-first by block count."""
    from bitswap_amd import dist as bdist
    if name == "imagenetcrop4":
        rng = np.random.RandomState(seed)
        tiles = rng.randint(256, 513, size=(total, 2)) // 32
        lengths = np.maximum(1, np.round(tiles[:, 0] * tiles[:, 1] * (steps / 256.0))).astype(int).tolist()
        weights = lengths
    else:
        lengths, weights = [int(steps)] * total, None
    out = []
    for r in range(world):
        ids = bdist.shard_chains(total, world, r, weights=weights)
        out.append((ids, [lengths[c] for c in ids]))
    return out


def gather_and_digest(parts, mine, total, rank):
    """The path's only exchange (not timed): the finished bitstreams to rank 0 (RCCL over xGMI when world > 1, a local no-op
    otherwise).  parts: device-side snapshots (stack, len, head) of the chain groups; they are packed ON THE DEVICE and the
    packed tensor goes straight into the gather (bitswap_amd.dist.gather_streams_device) -- one device-to-host copy, on rank 0.
    The digest -- CRC-32 over the streams in chain order -- does not depend on how the chains were sharded when the conv route
    is batch-invariant: the same value at 1, 2, 4 and 8 GPUs (--scaling strong)."""
    import zlib
    from bitswap_amd import dist as bdist
    try:
        tg = time.perf_counter()
        got = bdist.gather_streams_device(parts, mine, total)
        tg = time.perf_counter() - tg
        if rank != 0:
            return None
        crc = 0
        for a in got:
            crc = zlib.crc32(np.ascontiguousarray(a, dtype=np.uint32).tobytes() if a is not None else b"missing", crc)
        return {"chains": len(got), "bytes": 4 * int(sum(len(a) for a in got if a is not None)), "ms": round(tg * 1e3, 2),
                "complete": all(a is not None for a in got), "packed_on": "device",
                "crc32_of_streams_in_chain_order": f"{crc:08x}"}
    except Exception as e:   # never lose the bench line to the reporting exchange
        return {"error": repr(e)}


def run_workload(args, name, B, groups, K, W, dev, rank, world, dist, want_gather=False, want_roofline=True, regime=None,
                 chain_ids=None, total=None):
    """One measurement by the contract's procedure.  Returns a dict (timings are max over ranks).  chain_ids / total:
    --scaling strong -- this rank codes the chains `chain_ids` of `total`; inputs and initial words are a function of the
    GLOBAL chain id, so a chain's stream does not depend on the rank that codes it."""
    from bitswap_amd import workload
    from bitswap_amd.codec import GroupedCodec, Timeline, initial_states

    model, zend, zcen = workload.build(name, dev, quantbits=args.quantbits, regime=regime, nn_batch=args.nn_batch or None)
    n = K + W
    strong = chain_ids is not None
    if strong:
        assert regime is None and B == len(chain_ids)
        images = workload.synthetic_blocks(total * n, model.xs, seed=1000).view(total, n, -1)[chain_ids].to(torch.int32).to(dev)
    elif regime == "lowrate":      # blocks from the calibrated model's own generative path (workload.lowrate_blocks)
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
    if strong:
        every = initial_states(total, 10000, seed=100)
        init = [every[c] for c in chain_ids]
    else:
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

    # BITSWAP_BENCH_SENTINEL=1 (profiling runs only): a one-wavefront k_where launch on either side of the timed region, so that
    # tools/prof_summary.py can cut a rocprofv3 kernel trace / counter pass down to the timed region (the
    # whole-process summary also holds model building and bin sampling)
    sentinel = os.environ.get("BITSWAP_BENCH_SENTINEL") == "1"
    mark_buf = torch.zeros(4, dtype=torch.int32, device=dev) if sentinel else None

    def mark():
        if sentinel:
            from bitswap_amd import hip as _h
            _h.load().bs_debug_where(_h._ptr(mark_buf), 1, 0, _h._stream())

    barrier()
    mark()
    t0 = time.perf_counter()
    codec.encode_blocks(states, images[:, W:], rest_lens if W == 0 else None)
    len_sent = torch.cat([st.len for st in states]).clone()
    # the finished bitstreams (what a sender would ship): device-side snapshot, gathered after the clock stops
    sent = ([(st.stack.clone(), st.len.clone(), st.head.clone()) for st in states]
            if (want_gather and args.format == "reference") else None)
    decoded = codec.decode_blocks(states, K)
    barrier()
    dt = time.perf_counter() - t0
    mark()
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
        mine = list(chain_ids) if strong else [rank + world * c for c in range(B)]   # global chain ids, round-robin like shard_chains()
        gather = gather_and_digest(sent, mine, total if strong else world * B, rank)
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
        pivot = bool(spec >= 2 and args.format == "reference" and hasattr(be, "table_layout")
                     and be.table_layout(Kb, True, Z, int(rows // Z)) == _hip.LAYOUT_PIVOT)
        slots = VALU_SLOTS_PER_ROW[spec] if (Kb == 1024 and args.format == "reference") else None
        if slots is not None and spec >= 2 and not pivot:
            slots += WHOLE_ROW_EXTRA_SLOTS                # whole-row hand-off (BITSWAP_PIVOT=0)
        kname = (f"k_layer64<16,float,{'uniform' if spec >= 2 else 'generic'},pop> (logistic CDF -> integer table -> rANS pop in "
                 f"one launch, rows in registers, CDF spec {spec})" if args.format == "wave64" else
                 f"k_logistic<16,float,{'pivot' if pivot else 'decode'},spec{spec}> (fused logistic CDF -> "
                 f"integer table -> {'64 cumulative values per row for bs_rans_pop_pivot' if pivot else 'cdf rows'}, CDF spec {spec})")
        # --- HBM side, on bytes that move.  SURVEY 8(d) prices a z-row at (K-1)*8 + 12 B because it counts the [Z, K-1]
        # float64 endpoint table once per BLOCK; a launch over `chains` blocks reads that table from HBM once (the rest are
        # L2 hits), so the batched algorithmic bytes are table + 12 B/row + the hand-off to the pop kernel (512 B/row of
        # cumulative values with BS_LAYOUT_PIVOT; a whole 4 (K + 64) B row before: counted by the PMC figure)
        table_bytes = Z * (Kb - 1) * 8
        alg_batched = int(table_bytes + rows * (12 + (512 if pivot else 0)))
        survey_alg = int(rows * ((Kb - 1) * 8 + 12))
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and args.format == "reference":
            try:
                tj = json.load(open(tp)).get(name, {})
                per_row = tj.get("k_logistic_pivot_bytes_per_row" if pivot else "k_logistic_decode_bytes_per_row")
                traffic = None if per_row is None else int(per_row * rows)   # PMC bytes/row x rows of one launch
                traffic_src = [f if f.startswith("profiles/") else "profiles/" + f for f in tj.get("source", [])]
            except Exception:
                traffic = None
        a_block = algorithmic_bytes_per_block(codec)
        path = 2.0 * a_block * (total if strong else world * B) * K / dt / 1e9
        hbm = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "alg_bytes_per_launch": alg_batched, "achieved": round(alg_batched / avg / 1e9, 1),
               "frac": round(alg_batched / avg / 1e9 / HBM_PEAK_GBPS, 5),
               "traffic_bytes_per_launch": traffic, "traffic_source": traffic_src,
               "traffic_achieved": None if traffic is None else round(traffic / avg / 1e9, 1),
               "traffic_frac": None if traffic is None else round(traffic / avg / 1e9 / HBM_PEAK_GBPS, 4),
               "note": "alg = endpoint table once per launch + 12 B/row (mu, scale, symbol)"
                       + (" + the 512 B/row hand-off of cumulative values to bs_rans_pop_pivot" if pivot else "")
                       + "; traffic = PMC FETCH x2 + WRITE (profiles/traffic.json)"
                       + ("" if pivot else ", 98 % of it the cdf-row hand-off to k_rans_pop_wave")
                       + f"; SURVEY 8(d)'s per-block count would be {survey_alg} B per launch, of which all but the first table "
                         "pass are L2 hits"}
        kkey = f"k_logistic<16,float,{'pivot' if pivot else 'decode'},spec{spec}>"
        # the SURVEY 8(d)-literal figure for this kernel, kept visible: (K-1)*8 + 12 B per row x rows / launch time against the
        # HBM peak.  Above 1 whenever more than one chain shares a launch: all but the first pass over the [Z, K-1] endpoint
        # table are L2 hits, so it is not an HBM figure -- which is why `frac` is quoted on the bound the kernel sits on
        survey = {"hbm_survey_bytes_per_launch": survey_alg, "hbm_survey_achieved_GBps": round(survey_alg / avg / 1e9, 1),
                  "hbm_survey_frac": round(survey_alg / avg / 1e9 / HBM_PEAK_GBPS, 4),
                  "hbm_survey_note": "SURVEY 8(d) literal: 8196 B/row x rows_per_launch / avg_launch_ms / 8 TB/s; > 1 = L2 hits "
                                     "(the endpoint table is re-read from L2 by every chain of a launch), not HBM traffic"}
        # The record's headline (`bound`, `achieved`, `peak`, `frac`) is the WHOLE-PATH figure SURVEY 8(d) / BASELINE.md 2.3 define
        # -- 2 x A_block x blocks / (t_sender + t_receiver) against the HBM peak -- the one 8(d) number that stays <= 1 whatever
        # the batch.  The dominant kernel's own figures sit under "kernel_*": it is bound by VALU issue (`valu_issue_frac`), and
        # the 8(d)-literal per-launch byte count over its time exceeds the HBM peak because all but the first pass over the
        # endpoint table are L2 hits (`hbm_survey_frac`, annotated).
        roof = {"bound": "hbm", "achieved": round(path / world, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(path / world / HBM_PEAK_GBPS, 4),
                "definition": "whole path, SURVEY 8(d): 2 x algorithmic bytes per block x blocks / (t_sender + t_receiver) / 8 TB/s",
                "path_frac": round(path / world / HBM_PEAK_GBPS, 4), "kernel": kname, **survey}
        if slots is not None:
            ach = rows * slots / avg / 1e9
            roof.update({"kernel_bound": "valu_issue", "kernel_achieved_Ginstr": round(ach, 1), "kernel_peak_Ginstr": round(VALU_PEAK_GINSTR, 1),
                         "valu_issue_frac": round(ach / VALU_PEAK_GINSTR, 4), "slots_per_row": slots,
                         # measured by SQ counters (profiles/valu_busy.json, tools/pmc_valu.sh): share of the SIMD cycles the
                         # VALU pipe is busy while THIS flavour of the kernel runs; None: no counter run of it on file
                         "valu_busy_pmc": _valu_busy(kkey)})
        else:       # no issue model for this kernel shape: the counter-side HBM figure stands for the kernel
            roof.update({"kernel_bound": "hbm", "kernel_hbm_frac": hbm["traffic_frac"] if hbm["traffic_frac"] is not None else hbm["frac"]})
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
           "value": (total if strong else world * B) * K * 1024 / dt, "ms_per_step": dt / K * 1e3, "lossless": ok, "bits_per_dim": bpd,
           "stream_time_fraction": breakdown, "roofline": roof, "stream_gather": gather,
           "codec": codec, "model": model}
    return res


def run_ragged(args, name, chain_ids, lengths, total, K, W, dev, rank, world, dist, want_gather=True):
    """BASELINE config 4's shape: one chain per image, chains of different lengths in ONE lock-step run (the active set
    shrinks as short images finish: BitSwapCodec.compress_ragged / decompress_ragged), the crop model on its fixed conv
    micro-batches (nn_batch: an image's stream does not depend on the images it is coded with, imagenetcrop_compress.py:
    279-300 codes them one at a time).  Timed like the contract says: W untimed block steps of every chain first, then the
    whole sender run and the receiver run that undoes it between barriers; value = blocks coded by all ranks x 1024 / t."""
    from bitswap_amd import workload
    from bitswap_amd.codec import BitSwapCodec, initial_states

    nn_batch = int(os.environ.get("BITSWAP_CROP_NN_BATCH", "32"))
    model, zend, zcen = workload.build(name, dev, quantbits=args.quantbits, nn_batch=nn_batch)
    codec = BitSwapCodec(model, zend, zcen, quantbits=args.quantbits, bitswap=bool(args.bitswap), cdf_spec=args.cdf_spec)
    if args.no_graphs:
        codec.use_graphs = False
    B = len(chain_ids)
    chains = [workload.synthetic_blocks(n, model.xs, seed=5000 + c).to(torch.int32).to(dev) for c, n in zip(chain_ids, lengths)]
    one = initial_states(1, 10000, 100)[0]               # every image starts from the same words (:249,122)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if W:       # warm-up: W block steps of every chain, undone
        wch = [workload.synthetic_blocks(W, model.xs, seed=9000 + c).to(torch.int32).to(dev) for c in chain_ids]
        st, order, met = codec.compress_ragged(wch)
        codec.decompress_ragged(st, met["nblocks"])
        del st, wch
    staged = codec.stage_ragged(chains)
    state = codec.new_states(B, staged[2][0], states=[list(one) for _ in range(B)])
    barrier()
    t0 = time.perf_counter()
    state, order, met = codec.compress_ragged(chains, state=state, staged=staged)
    sent = (state.stack.clone(), state.len.clone(), state.head.clone()) if want_gather else None
    out = codec.decompress_ragged(state, met["nblocks"])
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = all(torch.equal(out[k], chains[i]) for k, i in enumerate(order)) and state.to_lists() == [list(one)] * B
    nblk = int(sum(lengths))
    bits = float(met["total"].sum())
    gather = None
    if sent is not None:      # row k of the ragged state is chain order[k] of this rank's list
        gather = gather_and_digest(sent, [chain_ids[i] for i in order], total, rank)
    tot = torch.tensor([bits, float(nblk * codec.X), float(ok), float(nblk)], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tot)
    return {"workload": name, "chains_per_gpu": B, "chain_groups": 1, "steps": K, "warmup": W,
            "value": float(tot[3]) * 1024 / dt, "ms_per_step": dt / K * 1e3, "lossless": bool(tot[2].item() == world),
            "bits_per_dim": float(tot[0] / tot[1]), "stream_time_fraction": None, "roofline": None, "stream_gather": gather,
            "blocks_total": int(tot[3].item()), "blocks_this_rank": nblk, "nn_batch": nn_batch,
            "codec": codec, "model": model}


def extra_in_child(args, spec, steps, warmup):
    """One `extra` sub-result measured the way the headline is: the first workload of a fresh process.  (Measured in the
    headline's process, after its buffers came and went, ImageNet32 nz=4 at 1000 chains reads 6.3 Mpixel/s against 6.8 on
    its own -- profiles/r03z: device memory handed out late in a process's life is more fragmented.)  -> dict."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", spec["workload"], "--groups", str(spec.get("groups", 0)),
           "--format", spec.get("format", "reference"), "--steps", str(spec.get("steps", steps)), "--warmup", str(warmup),
           "--quantbits", str(args.quantbits), "--bitswap", str(spec.get("bitswap", args.bitswap)), "--cdf-spec", str(args.cdf_spec),
           "--no-extra", "--no-cpu-baseline", "--no-roofline", "--full-record", os.devnull]
    if spec.get("scaling") == "strong":
        cmd += ["--scaling", "strong", "--total-chains", str(spec["chains"])]
    else:
        cmd += ["--chains", str(spec["chains"])]
    if spec.get("regime"):
        cmd += ["--regime", spec["regime"]]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **spec.get("env", {})))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError((r.stderr or r.stdout)[-400:])
    d = json.loads(lines[-1])
    return {"config": d["config"]["workload"] + f", {d['config']['chains_per_gpu']} chains / {d['config']['chain_groups']} group(s)"
                      + (", calibrated low-rate regime (workload.calibrate_lowrate: latent scales at the 0.1 clamp, pixel scale "
                         "0.0035, blocks drawn from the model's own generative path)" if spec.get("regime") == "lowrate" else "")
                      + (", opt-in 64-state stream format (not the reference's word stream)" if spec.get("format") == "wave64" else ""),
            "why": spec.get("why"), "workload": spec["workload"], "bitswap": int(spec.get("bitswap", args.bitswap)), "tag": spec.get("tag"),
            "chains_per_gpu": d["config"]["chains_per_gpu"],
            "chain_groups": d["config"]["chain_groups"], "scaling": d["scaling"], "steps": d["steps"], "warmup": d["warmup"],
            "value": round(d["value"], 1), "ms_per_step": round(d["ms_per_step"], 3), "lossless": d["lossless"],
            "bits_per_dim": round(d["bits_per_dim"], 4), "stream_format": spec.get("format", "reference"),
            "regime": spec.get("regime") or "random-init weights, unrelated synthetic blocks",
            "forked_block_step": d["config"].get("forked_block_step"), "conv_dtype": d["config"].get("conv_dtype"),
            "stream_time_fraction": d.get("stream_time_fraction"), "stream_gather": d.get("stream_gather"), "roofline": None,
            "process": "own process (python bench.py --no-extra ...), like the headline"}


# driver-visible numbers for the other shapes DESIGN.md quotes, each measured like the headline in a process of its own
EXTRAS = (
    dict(workload="imagenet4", chains=1000, groups=2, core=True, why="north_star's target shape (configs[2]) at the headline's batch"),
    dict(workload="cifar8", chains=100, scaling="strong", core=True, why="configs[1] at the reference's own shape: 100 experiments in total"),
    dict(workload="imagenet4", chains=100, scaling="strong", core=True, why="configs[2]: 100 experiments x 32x32 blocks"),
    dict(workload="imagenet4", chains=100, scaling="strong", bitswap=0, why="configs[4] (BB-ANS) at N = 1"),
    dict(workload="imagenetcrop4", chains=100, scaling="strong", steps=16,
         why="configs[3] at N = 1: 100 ragged image chains (a 512 x 512 image = 16 blocks here), crop model on nn_batch 32"),
    dict(workload="cifar8", chains=13, groups=1, core=True, why="one GPU's share of 100 chains on 8 GPUs"),
    dict(workload="imagenet4", chains=13, groups=1, bitswap=0, why="configs[4]: one GPU's share (13 of 100 chains) on 8 GPUs"),
    dict(workload="imagenetcrop4", chains=13, scaling="strong", steps=16, why="configs[3]: one GPU's share (13 of 100 images) on 8 GPUs"),
    # the 2- and 4-GPU shares of the same 100 chains: with the 100- and 13-chain lines they give the predicted
    # 1 / 2 / 4 / 8-GPU strong-scaling curve of configs[1], [3], [4] (predicted_scaling below)
    dict(workload="cifar8", chains=50, groups=1, share_of=2, core=True, why="one GPU's share of 100 chains on 2 GPUs"),
    dict(workload="cifar8", chains=25, groups=1, share_of=4, core=True, why="one GPU's share of 100 chains on 4 GPUs"),
    dict(workload="imagenet4", chains=50, groups=1, bitswap=0, share_of=2, why="configs[4]: one GPU's share on 2 GPUs"),
    dict(workload="imagenet4", chains=25, groups=1, bitswap=0, share_of=4, why="configs[4]: one GPU's share on 4 GPUs"),
    dict(workload="imagenetcrop4", chains=50, scaling="strong", steps=16, share_of=2, why="configs[3]: one GPU's share (50 of 100 images) on 2 GPUs"),
    dict(workload="imagenetcrop4", chains=25, scaling="strong", steps=16, share_of=4, why="configs[3]: one GPU's share (25 of 100 images) on 4 GPUs"),
    dict(workload="cifar8", chains=1500, groups=2, why="a bigger batch than the headline's (the headline stays at 1000 chains per GPU for continuity "
                                                        "with rounds 3-5): what the chip gives with 1500"),
    dict(workload="cifar8", chains=1000, groups=2, regime="lowrate", why="peaked tables: a trained model's rate"),
    dict(workload="cifar8", chains=1000, groups=2, env={"BITSWAP_GEMM_ARITH": "fp32"},
         why="the conv arithmetic of rounds 2-5, now the opt-out: the ResNet products on the fp32 matrix pipe (bs_wino_gemm_f32, "
             "v_mfma_f32_32x32x2_f32) instead of three bf16 limbs per operand on the bf16 pipe (DESIGN 3.4)"),
    dict(workload="cifar8", chains=100, scaling="strong", env={"BITSWAP_GEMM_ARITH": "fp32"},
         why="the fp32-MFMA arithmetic at the reference's own shape (100 chains, forked two-stream step)"),
    dict(workload="cifar8", chains=1000, groups=2, env={"BITSWAP_SERIAL_CUS": "32", "BITSWAP_GEMM_CUS": "224"}, tag="cumask32",
         why="EXPERIMENT, a loss (DESIGN 8): CU-masked streams -- the serial pop / push streams on 32 compute units (4 per XCD), the bulk "
             "streams on the other 224 (bs_stream_create_cu_mask); more masks and the bf16x3 pairing: profiles/r05b_cu_mask_ab.txt"),
    dict(workload="cifar8", chains=800, groups=2, format="wave64", why="opt-in 64-state format"),
    dict(workload="cifar8", chains=13, groups=1, format="wave64", why="opt-in 64-state format, few chains"),
)


def config_title(name, args, strong):
    t = TITLES[name] + f" {'Bit-Swap' if args.bitswap else 'BB-ANS'}, batch of 32x32 blocks"
    if strong:
        cfg = {"imagenetcrop4": "BASELINE configs[3]", "cifar8": "BASELINE configs[1]"}.get(
            name, "BASELINE configs[4]" if (name == "imagenet4" and not args.bitswap) else "BASELINE configs[2]" if name == "imagenet4" else None)
        t += (f"; {args.total_chains} chains IN TOTAL sharded over the ranks"
              + (" (ragged: one image = one chain, LPT by block count)" if name == "imagenetcrop4" else " (round-robin)")
              + (f" -- {cfg}" if cfg else ""))
    return t


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
    strong = args.scaling == "strong"
    plan = None
    if strong:
        if args.total_chains < world:
            raise SystemExit(f"--scaling strong: {args.total_chains} chains cannot occupy {world} ranks")
        plan = strong_plan(args.total_chains, world, name, args.steps)
        ids, lengths = plan[rank]
        chains = len(ids)
    else:
        chains = args.chains
    groups = args.groups or (2 if chains > 128 else 1)
    if strong and name == "imagenetcrop4":
        r = run_ragged(args, name, ids, lengths, args.total_chains, args.steps, args.warmup, dev, rank, world, dist)
        groups = 1
    else:
        r = run_workload(args, name, chains, groups, args.steps, args.warmup, dev, rank, world, dist, want_gather=True,
                         want_roofline=not args.no_roofline and not strong, regime=args.regime,
                         chain_ids=ids if strong else None, total=args.total_chains if strong else None)
    codec, model = r.pop("codec"), r.pop("model")
    gemm = ("bs_wino_gemm_f32 (own fp32 MFMA kernel, every product: results independent of chains per call)"
            if getattr(model, "own_gemm", False) else f"BLAS backend {model.gemm_backend}")
    conv_path = (f"{model.conv_algo} (fp32; ResNet/head convs as transform-domain batched GEMMs on {gemm}; 3x3 input convs: "
                 f"{'bs_conv3_wino_f32' if getattr(model, 'fused_inputs', False) else 'MIOpen'}; 5x5 input conv: "
                 f"{'Winograd domain (bs_small_k_gemm_f32)' if getattr(model, 'wino_in5', False) else 'MIOpen'})"
                 if getattr(model, "fused", False) else "torch modules")
    Z, X = codec.Z, codec.X
    arith = getattr(model, "gemm_arith", "fp32") if getattr(model, "_ufrags", None) else "fp32"
    if arith != "fp32":
        conv_path = conv_path.replace("(fp32;", f"(float32 activations and weights, {arith} arithmetic in the ResNet products: "
                                      "three bf16 limbs per operand on the bf16 matrix cores, float32 accumulate;")
        conv_path = conv_path.replace("bs_wino_gemm_f32 (own fp32 MFMA kernel", "bs_wino_gemm_bf16x3 / bs_wino_gemm_f32 for the head products (own MFMA kernels")
    forked = bool(sum(c.forked_steps for c in getattr(codec, "codecs", [codec])))
    del codec, model

    extra = None
    if (world == 1 and not args.no_extra and args.extra != "none" and not strong and name == "cifar8"
            and args.format == "reference"):
        extra = []
        ks, ws = min(args.steps, 6), max(min(args.warmup, 1), 2)
        import gc
        gc.collect()                       # the headline's model, bins and states go before the children are started
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        for spec in EXTRAS:
            if args.extra == "core" and not spec.get("core"):
                continue
            try:
                extra.append(extra_in_child(args, spec, ks, ws))
            except Exception as ex:   # a sub-result never costs the headline
                extra.append({"workload": spec["workload"], "chains_per_gpu": spec["chains"], "why": spec.get("why"), "error": repr(ex)})

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- the other shapes in compact form, and the strong-scaling curve they predict -----------------------------------
    # predicted N-GPU rate of a fixed 100-chain job = N x the rate ONE GPU reaches on its share of the chains (ceil(100 / N)
    # chains, measured above in a process of its own); the ranks never talk while coding (bitswap_amd/dist.py) and the gather
    # of the finished streams is outside the timed region, so what the prediction leaves out is box-to-box spread only.
    shapes, predicted = None, None
    headline_dtype = "f32" if arith == "fp32" else arith
    if extra:
        def label(e):
            return (f"{e['workload']}{'' if e.get('bitswap', 1) else '_bbans'}_{e['chains_per_gpu']}"
                    + ("_wave64" if e.get("stream_format") == "wave64" else "")
                    + ("_lowrate" if e.get("regime") == "lowrate" else "")
                    + ("_" + e["conv_dtype"] if e.get("conv_dtype") not in (None, headline_dtype) else "") + ("_" + e["tag"] if e.get("tag") else ""))
        shapes = {label(e): [round(e["value"] / 1e6, 3), e["ms_per_step"], e["lossless"]] for e in extra if "value" in e}
        predicted = {}
        for cfg, wl, bs in (("configs[1] cifar8 Bit-Swap", "cifar8", 1), ("configs[4] imagenet4 BB-ANS", "imagenet4", 0),
                            ("configs[3] imagenetcrop4 ragged", "imagenetcrop4", 1)):
            pick = lambda n: next((e["value"] for e in extra if e.get("workload") == wl and e.get("bitswap", 1) == bs and "value" in e
                                   and e.get("chains_per_gpu") == n and e.get("stream_format", "reference") == "reference"
                                   and e.get("conv_dtype", headline_dtype) == headline_dtype and e.get("regime") != "lowrate" and not e.get("tag")), None)
            v = {1: pick(100), 2: pick(50), 4: pick(25), 8: pick(13)}
            if v[1]:
                predicted[cfg] = {"Mpixel_per_s": {str(n): (None if x is None else round(n * x / 1e6, 2)) for n, x in v.items()},
                                  "speedup": {str(n): (None if x is None else round(n * x / v[1], 2)) for n, x in v.items()}}
        predicted["how"] = ("N x the measured 1-GPU rate of one rank's share (100, 50, 25, 13 chains) of the same 100-chain job; no "
                            "collective while coding; reference stream format: the serial rANS chain bounds few-chain steps (DESIGN 5)")

    cpu = None
    if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
        try:
            cpu = cpu_baseline(args, name)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            cpu = {"value": None, "unit": "pixels/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    out = {
        "metric": "pixels/s (encode+decode)", "value": round(r["value"], 1), "unit": "pixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": config_title(name, args, strong),
                   "chains_per_gpu": r["chains_per_gpu"], "chain_groups": r["chain_groups"], "blocks_per_chain": args.steps,
                   "quantbits": args.quantbits, "ansbits": 31, "cdf_spec": args.cdf_spec, "stream_format": args.format,
                   "latent_dims": Z, "pixel_dims": X, "conv_dtype": "f32" if arith == "fp32" else arith, "conv_path": conv_path,
                   "forked_block_step": forked, "nn_batch": args.nn_batch or None,
                   "weights": "seeded random init (no checkpoints offline)"
                              + (", calibrated to the low-rate regime (workload.calibrate_lowrate)" if args.regime else ""),
                   # the other shapes measured by this run, each in a process of its own like the headline (full records:
                   # `extra`): label -> [Mpixel/s enc+dec, ms per step, lossless]; and the strong-scaling curve they predict
                   "measured_shapes": shapes, "predicted_scaling": predicted},
        "rccl_ranks": (world if (world > 1 and (os.environ.get("BENCH_DIST_BACKEND") or "nccl") == "nccl") else 0),
        "lossless": r["lossless"], "bits_per_dim": round(r["bits_per_dim"], 4),
        "stream_time_fraction": r["stream_time_fraction"],
        "roofline": r["roofline"], "cpu_baseline": cpu, "stream_gather": r["stream_gather"], "extra": extra,
    }
    if strong:
        out["config"].update({"total_chains": args.total_chains, "chains_per_rank": [len(p[0]) for p in plan],
                              "blocks_per_rank": [int(sum(p[1])) for p in plan]})
        for k in ("blocks_total", "nn_batch"):
            if k in r:
                out["config"][k] = r[k]
    # the FULL record (every sub-result and annotation) goes to a file; stdout's LAST line is the compact headline: the
    # driver keeps 8 KB of stdout, and round 5's 26 KB line reached it cut in two (BENCH_r05.json: parsed null)
    try:
        if args.full_record != os.devnull:
            os.makedirs(os.path.dirname(args.full_record), exist_ok=True)
        with open(args.full_record, "w") as f:
            json.dump(out, f)
    except OSError as e:
        print(f"bench.py: full record not written: {e!r}", file=sys.stderr)
    print(headline_line(out, args.full_record), flush=True)
    if dist is not None:
        dist.destroy_process_group()


HEADLINE_MAX_BYTES = 6144
BENCH_SCHEMA = 6   # 6: compact stdout line + full record on file; roofline.frac = whole path as in schema 5,
#                       the dominant kernel's own figures under roofline.kernel (rounds 1-4: roofline.frac = kernel VALU issue)


def headline(out, full_path=None):
    """The compact line the driver parses: the contract's keys, ONE copy of the measured shapes and the predicted curve, the
    whole-path roofline with the dominant hot-path kernel's and the GEMM's own figures beside it, the CPU baseline.  Everything
    else (twenty-odd annotations per object) stays in the full record."""
    def pick(d, keys):
        return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}
    cfg = out["config"]
    roof = out.get("roofline")
    hroof = None
    if roof:
        hb = roof.get("hbm") or {}
        mf = roof.get("mfma") or {}
        hroof = {"bound": roof["bound"], "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"],
                 "frac": roof["frac"], "traffic": roof.get("traffic"),
                 "what": "frac = whole path, SURVEY 8(d): 2 x A_block x blocks / (t_sender + t_receiver) / 8 TB/s; kernel.* = the "
                         "dominant hot-path kernel, exclusive HIP-event launch time; traffic = its PMC bytes per launch",
                 "alg_bytes_per_block": roof.get("path_alg_bytes_per_block"),
                 "kernel": {"name": str(roof.get("kernel", "")).split(" (")[0], "rows_per_launch": roof.get("rows_per_launch"),
                            "avg_launch_ms": roof.get("avg_launch_ms"), "avg_launch_ms_in_pipeline": roof.get("avg_launch_ms_in_pipeline"),
                            "bound": roof.get("kernel_bound"), "valu_issue_frac": roof.get("valu_issue_frac"),
                            "slots_per_row": roof.get("slots_per_row"),
                            "valu_busy_pmc": (roof.get("valu_busy_pmc") or {}).get("valu_busy"),
                            "hbm_survey_frac": roof.get("hbm_survey_frac"), "hbm_alg_frac": hb.get("frac"),
                            "hbm_traffic_frac": hb.get("traffic_frac"), "fp64_frac": (roof.get("fp64") or {}).get("frac")},
                 "mfma": pick(mf, ("kernel", "arith", "shape", "achieved", "peak", "unit", "frac", "fp32_equivalent_TFLOPs", "avg_launch_ms", "error"))}
        if hroof["mfma"] and "kernel" in hroof["mfma"]:
            hroof["mfma"]["kernel"] = hroof["mfma"]["kernel"].split(" (")[0]
    cpu = out.get("cpu_baseline")
    hcpu = None
    if cpu:
        hcpu = pick(cpu, ("value", "unit", "cores", "kind", "same_host", "sample"))
        if hcpu.get("value") is not None:
            hcpu["value"] = round(hcpu["value"], 1)
        rp = cpu.get("reference_python")
        hcpu["reference_python"] = pick(rp, ("value", "unit", "cores", "same_host", "tool"))
    ps = cfg.get("predicted_scaling") or {}
    h = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data")}
    h["config"] = {k: cfg[k] for k in ("workload", "chains_per_gpu", "chain_groups", "blocks_per_chain", "quantbits", "ansbits",
                                       "cdf_spec", "stream_format", "conv_dtype", "forked_block_step", "total_chains",
                                       "chains_per_rank", "blocks_per_rank", "blocks_total", "nn_batch") if k in cfg}
    h["config"]["weights"] = "seeded random init" + (", low-rate calibrated" if "low-rate" in str(cfg.get("weights")) else "")
    h.update({"lossless": out["lossless"], "bits_per_dim": out["bits_per_dim"], "rccl_ranks": out["rccl_ranks"],
              "roofline": hroof, "cpu_baseline": hcpu,
              "stream_gather": pick(out.get("stream_gather"), ("chains", "bytes", "ms", "complete", "crc32_of_streams_in_chain_order", "error")),
              # label -> [Mpixel/s enc+dec, ms per step, lossless], each the first workload of a process of its own
              "measured_shapes": cfg.get("measured_shapes"),
              "predicted_scaling": {k: v.get("Mpixel_per_s") for k, v in ps.items() if isinstance(v, dict)} or None,
              "schema": BENCH_SCHEMA, "full_record": None if full_path is None else os.path.relpath(full_path, ROOT)})
    return h


def headline_line(out, full_path=None):
    """headline() as ONE line below HEADLINE_MAX_BYTES: optional objects are dropped, longest first, before the contract's keys
    would be cut (never needed with the shapes bench.py measures today; a guard, not a code path)."""
    h = headline(out, full_path)
    for drop in (None, "measured_shapes", "predicted_scaling", "stream_gather"):
        if drop:
            h[drop] = None
        line = json.dumps(h, separators=(",", ":"))
        if len(line) < HEADLINE_MAX_BYTES:
            return line
    raise RuntimeError(f"bench headline is {len(line)} bytes")


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
