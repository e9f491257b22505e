#!/usr/bin/env python3
"""bench.py -- pixels/s (encode+decode) of the Bit-Swap compression path on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--workload cifar8|imagenet4|mnist2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one 32x32 block of every chain coded in lock-step, sender AND receiver: the timed
region runs K sender block-steps followed by the K receiver block-steps that undo them, bracketed
by barrier + synchronize on both sides (max over ranks).  Each rank owns `--chains` independent
chains (weak scaling: chains never talk to each other; the only collective is the gather of the
finished bitstreams, outside the timed region).  The inputs (blocks, weights, bins, initial
stacks) are resident in HBM before the timed region starts.

Default workload = BASELINE.json configs[1]: CIFAR-10-shaped 8-latent-layer Bit-Swap model
(reswidth 252, Z = 2048, X = 3072, K = 1024 / 256), 100 chains, synthetic data and seeded
random-init weights (no datasets/checkpoints offline).

Extra objects on the JSON line: `roofline` for the dominant hot-path kernel (the fused
logistic-CDF -> integer-table kernel, decode flavour) from HIP events recorded on the launch stream
inside the timed region, and `cpu_baseline`: the oracle (C restatement of the reference, libm CDF)
+ the same conv stacks on the host cores, timed on a bounded sample of the same workload.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cifar8", choices=["mnist2", "cifar8", "imagenet4", "imagenetcrop4"])
    ap.add_argument("--chains", type=int, default=800,
                    help="independent chains (rANS streams) per GPU; the reference runs 100 'experiments' one block at a time, "
                         "an MI355X wants a few hundred in lock-step (conv efficiency grows with the batch, DESIGN.md 6)")
    ap.add_argument("--quantbits", type=int, default=10)
    ap.add_argument("--bitswap", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (no imagenet4 / 100-chain sub-results)")
    ap.add_argument("--cpu-blocks", type=int, default=20, help="blocks per chain in the CPU baseline sample")
    ap.add_argument("--no-timeline", action="store_true", help="skip per-kernel events (roofline becomes null)")
    ap.add_argument("--groups", type=int, default=2,
                    help="chain groups per GPU on separate HIP streams (serial rANS of one group under the convs of another)")
    ap.add_argument("--tables-on", default="bulk", choices=["bulk", "serial"],
                    help="stream of the table kernels when groups > 1 (see BitSwapCodec.tables_on)")
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
    return {"value": B * n * 1024 / dt, "unit": "pixels/s", "cores": threads, "kind": "port",
            "sample": f"{B} chains x {n} block(s) of {name}, sender+receiver, oracle C (libm CDF) + torch-CPU convs, "
                      f"{dt:.1f} s, lossless={ok}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_DIST_BACKEND=gloo lets the control flow be exercised with several ranks on ONE device
        # (RCCL refuses two ranks per GPU); the driver never sets it
        dist.init_process_group(os.environ.get("BENCH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo"),
                                timeout=datetime.timedelta(seconds=300))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the coding path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False

    from bitswap_amd import hip, workload
    from bitswap_amd.codec import GroupedCodec, Timeline, initial_states

    name = args.workload
    model, zend, zcen = workload.build(name, dev, quantbits=args.quantbits)
    B, K, W = args.chains, args.steps, args.warmup
    n = K + W
    images = workload.synthetic_blocks(B * n, model.xs, seed=1000 + rank).view(B, n, -1).to(torch.int32).to(dev)
    tl = Timeline(enabled=not args.no_timeline)
    codec = GroupedCodec(model, zend, zcen, groups=args.groups, quantbits=args.quantbits, bitswap=bool(args.bitswap),
                         timeline=tl)
    for c in codec.codecs:
        c.tables_on = args.tables_on
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
    tl.reset()

    barrier()
    t0 = time.perf_counter()
    codec.encode_blocks(states, images[:, W:], rest_lens if W == 0 else None)
    len_sent = torch.cat([st.len for st in states]).clone()
    # the finished bitstreams (what a sender would ship): device-side snapshot, gathered after the clock stops
    sent = [(st.stack.clone(), st.len.clone(), st.head.clone()) for st in states]
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
    ok = ok and codec.to_lists(states) == init
    bits = (len_sent.cpu().numpy().astype(np.int64) - np.array([len(s) - 1 for s in init])) * 32
    bpd = float(bits.sum()) / (B * K * codec.X)

    # ---- the path's only exchange (not timed): gather of the finished bitstreams to rank 0 and two scalars
    # for bits/dim -- RCCL over xGMI when world > 1 (bitswap_amd/dist.py), a local no-op otherwise
    from bitswap_amd import dist as bdist
    gather = None
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

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    totals = tl.totals()
    roof = None
    if "tables_z" in totals:
        sec, cnt = totals["tables_z"]
        Kb, Z = codec.K, codec.Z
        rows = B * Z / max(1, args.groups)                # rows one launch processes (one chain group)
        alg = int(rows * ((Kb - 1) * 8 + 2 * 4 + 4))      # endpoints f64 + mu,scale f32 + symbol i32
        ach = alg / (sec / cnt) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                per_row = json.load(open(tp)).get(name, {}).get("k_logistic_decode_bytes_per_row")
                traffic = None if per_row is None else int(per_row * rows)   # PMC bytes/row x rows of one launch
            except Exception:
                traffic = None
        roof = {"kernel": "k_logistic<16,float,decode> (fused logistic CDF -> integer cdf rows)", "bound": "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "launches": cnt, "avg_launch_ms": round(sec / cnt * 1e3, 4),
                "alg_bytes_per_launch": alg}
    breakdown = {k: round(v[0] / dt, 4) for k, v in sorted(totals.items())} if totals else None

    cpu = None
    if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
        try:
            cpu = cpu_baseline(args, name)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            cpu = {"value": None, "unit": "pixels/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    value = world * B * K * 1024 / dt
    out = {
        "metric": "pixels/s (encode+decode)", "value": round(value, 1), "unit": "pixels/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": {"mnist2": "MNIST-shaped 2-latent-layer", "cifar8": "CIFAR-10-shaped 8-latent-layer",
                                "imagenet4": "ImageNet32-shaped 4-latent-layer",
                                "imagenetcrop4": "ImageNet-crop-shaped 4-latent-layer"}[name] +
                               f" {'Bit-Swap' if args.bitswap else 'BB-ANS'}, batch of 32x32 blocks",
                   "chains_per_gpu": B, "chain_groups": args.groups, "blocks_per_chain": K, "quantbits": args.quantbits, "ansbits": 31,
                   "latent_dims": codec.Z, "pixel_dims": codec.X, "conv_dtype": "f32",
                   "conv_path": (f"{model.conv_algo} (fp32; ResNet/head convs as transform-domain batched GEMMs, "
                                 f"BLAS backend {model.gemm_backend})" if getattr(model, "fused", False) else "torch modules"),
                   "weights": "seeded random init (no checkpoints offline)"},
        "lossless": ok, "bits_per_dim": round(bpd, 4), "stream_time_fraction": breakdown,
        "roofline": roof, "cpu_baseline": cpu, "stream_gather": gather,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
