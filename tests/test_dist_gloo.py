"""CPU, world_size 2 over gloo: chain sharding and the final gather of the bitstreams -- the only
exchange on the path.  Ranks are real processes (torch.multiprocessing.spawn), rendezvous on 127.0.0.1."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as td
    import oracle as O
    from oracle.backend import OracleBackend
    from bitswap_amd import cli, dist
    r, w = dist.init("gloo")
    assert (r, w) == (rank, world)
    # gather of ragged streams
    nch = 5
    mine = dist.shard_chains(nch, world, rank)
    local = [np.arange(10 + 3 * c, dtype=np.uint32) * (c + 1) for c in mine]
    got = dist.gather_streams(local, mine, nch)
    if rank == 0:
        for c in range(nch):
            assert np.array_equal(got[c], np.arange(10 + 3 * c, dtype=np.uint32) * (c + 1))
    else:
        assert got is None
    rows = dist.gather_rows(np.array([[c, c * 2.0] for c in mine]), mine, nch)
    if rank == 0:
        assert np.array_equal(rows, np.array([[c, c * 2.0] for c in range(nch)]))
    assert dist.allreduce_sum([1.0, rank]) == [world, sum(range(world))]
    # LPT sharding by block count (config 4, imagenetcrop_compress.py:279-300): one long chain pins rank 0, every
    # short chain lands on rank 1 -- far more than ceil(nchains / world) -- and nothing may overflow or deadlock
    nch = 9
    weights = [100] + [1] * 8
    mine = dist.shard_chains(nch, world, rank, weights=weights)
    assert len(mine) == (1 if rank == 0 else 8)
    local = [np.arange(5 + c, dtype=np.uint32) + 1000 * c for c in mine]
    got = dist.gather_streams(local, mine, nch)
    rows = dist.gather_rows(np.array([[c, weights[c]] for c in mine], dtype=np.float64), mine, nch)
    if rank == 0:
        for c in range(nch):
            assert np.array_equal(got[c], np.arange(5 + c, dtype=np.uint32) + 1000 * c)
        assert np.array_equal(rows, np.array([[c, weights[c]] for c in range(nch)], dtype=np.float64))
    # a rank that owns nothing at all
    mine0 = [0, 1, 2] if rank == 1 else []
    got = dist.gather_streams([np.full(4, c, dtype=np.uint32) for c in mine0], mine0, 3)
    rows = dist.gather_rows(np.array([[c] for c in mine0], dtype=np.float64).reshape(len(mine0), 1), mine0, 3)
    if rank == 0:
        assert [a.tolist() for a in got] == [[c] * 4 for c in range(3)] and rows[:, 0].tolist() == [0.0, 1.0, 2.0]
    # ... built exactly as imagenetcrop_compress.py builds it for a rank without images: compress_images([]) is empty
    # and np.array([[o[2]] for o in []]) has shape (0,), no width
    res0 = cli.compress_images([], quantbits=6, nz=2, setup=None) if rank == 0 else [(None, 0, 1.5), (None, 0, 2.5)]
    rows = dist.gather_rows(np.array([[o[2]] for o in res0]), [] if rank == 0 else [0, 1], 2)
    if rank == 0:
        assert rows[:, 0].tolist() == [1.5, 2.5]
    # the crop driver's sender with ragged chains, LPT-sharded: every image is coded by exactly one rank and the
    # gathered bits/dim are those of a single-process run with the same nn_batch (batch-invariant convs)
    from bitswap_amd import tiling
    ob = OracleBackend(O.MODE_DET)
    setup = cli.crop_setup(-1, nz=2, quantbits=6, backend=ob, small=8, nn_batch=2)
    rng = np.random.RandomState(5)
    blocks = [tiling.extract_blocks(rng.randint(0, 256, (32 * a, 32 * b, 3)).astype(np.uint8))[0]
              for a, b in ((3, 2), (1, 1), (1, 2), (1, 1))]
    w = [len(b) for b in blocks]
    mine = dist.shard_chains(len(blocks), world, rank, weights=w)
    assert mine == ([0] if rank == 0 else [1, 2, 3])
    res = cli.compress_images([blocks[i] for i in mine], quantbits=6, nz=2, setup=setup, backend=ob)
    bpd = dist.gather_rows(np.array([[r[2]] for r in res]), mine, len(blocks))
    words = dist.gather_streams([np.array(r[0][:-1], dtype=np.uint32) for r in res], mine, len(blocks))
    if rank == 0:
        np.save(os.path.join(outdir, "crop_bpd_world2.npy"), bpd[:, 0])
        np.save(os.path.join(outdir, "crop_words_world2.npy"), np.array([len(a) for a in words]))
    # the full experiment driver, sharded
    res = cli.compress(6, 2, 1, 0, dataset="mnist", experiments=4, ndatapoints=2, decompress=True,
                       outdir=outdir, backend=OracleBackend(O.MODE_DET), small=8, verbose=False)
    if rank == 0:
        np.save(os.path.join(outdir, "cmas_world2.npy"), res["cmas"])
    td.barrier()
    td.destroy_process_group()


def test_shard_chains_policies():
    from bitswap_amd import dist
    assert dist.shard_chains(10, 4, 1) == [1, 5, 9]
    w = [8, 1, 1, 1, 5, 4]
    owners = [dist.shard_chains(6, 2, r, weights=w) for r in range(2)]
    assert sorted(owners[0] + owners[1]) == list(range(6))
    loads = [sum(w[c] for c in o) for o in owners]
    assert max(loads) - min(loads) <= 2           # LPT keeps the two ranks balanced
    assert dist.shard_chains(3, 1, 0) == [0, 1, 2]


def _as_state_parts(streams, split=None):
    """Word streams (stack words + the head as two words, low first) as the (stack, len, head) tensors of a RansState -- one
    triple, or two when `split` cuts the chains into two chain groups (what bench.py snapshots on the device)."""
    import torch

    def one(ss):
        cap = max([len(a) for a in ss] + [2])
        stack = torch.zeros((len(ss), cap), dtype=torch.int32)
        ln = torch.zeros(len(ss), dtype=torch.int32)
        head = torch.zeros(len(ss), dtype=torch.int64)
        for k, a in enumerate(ss):
            a = np.asarray(a, dtype=np.uint32)
            stack[k, : len(a) - 2] = torch.from_numpy(a[:-2].view(np.int32).copy())
            ln[k] = len(a) - 2
            head[k] = int((int(a[-1]) << 32 | int(a[-2])) - (1 << 64) if int(a[-1]) >> 31 else (int(a[-1]) << 32 | int(a[-2])))
        return stack, ln, head
    if split is None or not 0 < split < len(streams):
        return one(streams)
    return [one(streams[:split]), one(streams[split:])]


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_strong_scaling_plan_partitions_and_balances():
    """bench.py --scaling strong (BASELINE configs 4 / 5: 100 chains IN TOTAL over the ranks): at every world size the
    plan is a partition of the chains; equal chains go round-robin (12-13 per rank at 8 ranks), the ragged crop chains
    longest-processing-time-first with the ranks' block counts within one longest chain of each other; a chain's length
    does not depend on the world size."""
    bench = _bench()
    for name in ("imagenet4", "imagenetcrop4"):
        ref = None
        for world in (1, 2, 3, 4, 8):
            plan = bench.strong_plan(100, world, name, 16)
            ids = sorted(c for p in plan for c in p[0])
            assert ids == list(range(100)) and len(plan) == world
            lens = dict((c, n) for p in plan for c, n in zip(*p))
            ref = ref or lens
            assert lens == ref
            loads = [sum(p[1]) for p in plan]
            if name == "imagenetcrop4":
                assert 4 <= min(lens.values()) and max(lens.values()) <= 16 and len(set(lens.values())) > 3
                assert max(loads) - min(loads) <= 16
            else:
                assert set(lens.values()) == {16} and max(len(p[0]) for p in plan) - min(len(p[0]) for p in plan) <= 1
    assert [len(p[0]) for p in bench.strong_plan(100, 8, "imagenet4", 6)] == [13, 13, 13, 13, 12, 12, 12, 12]


def _worker_strong(rank, world, port, outdir, total=11):
    """bench.py's strong-scaling exchange over gloo: every rank makes the streams of ITS chains of the plan
    (here: a function of the global chain id), rank 0 gathers all of them and the digest is the one a single process gets."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as td
    from bitswap_amd import dist
    bench = _bench()
    assert dist.init("gloo") == (rank, world)
    for name in ("imagenet4", "imagenetcrop4"):
        ids, lengths = bench.strong_plan(total, world, name, 8)[rank]
        streams = [np.arange(5 + n, dtype=np.uint32) * (c + 1) for c, n in zip(ids, lengths)]
        g = bench.gather_and_digest(_as_state_parts(streams, split=len(streams) // 2), ids, total, rank)
        if rank == 0:
            assert g["complete"] and g["packed_on"] == "device" and g["chains"] == total
            with open(os.path.join(outdir, f"digest_{name}_w{world}.txt"), "w") as f:
                f.write(g["crc32_of_streams_in_chain_order"])
        else:
            assert g is None
    td.barrier()
    td.destroy_process_group()


def test_strong_scaling_gather_digest_is_independent_of_world_size(tmp_path):
    mp.spawn(_worker_strong, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    bench = _bench()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    for name in ("imagenet4", "imagenetcrop4"):
        ids, lengths = bench.strong_plan(11, 1, name, 8)[0]
        streams = [np.arange(5 + n, dtype=np.uint32) * (c + 1) for c, n in zip(ids, lengths)]
        one = bench.gather_and_digest(_as_state_parts(streams), ids, 11, 0)
        import zlib                                   # ... and the digest is the CRC-32 of the streams themselves, in chain order
        crc = 0
        for a in streams:
            crc = zlib.crc32(a.tobytes(), crc)
        assert one["crc32_of_streams_in_chain_order"] == f"{crc:08x}"
        assert one["crc32_of_streams_in_chain_order"] == open(tmp_path / f"digest_{name}_w2.txt").read()


def test_eight_rank_strong_plan_and_gather(tmp_path):
    """The world size north_star's curve ends at: BASELINE configs 4 / 5 -- 100 chains in total -- planned over
    EIGHT ranks (13, 13, 13, 13, 12, 12, 12, 12 equal chains round-robin; the ragged crop chains by LPT), every rank's streams
    gathered to rank 0 over gloo, and the CRC-32 of the streams in chain order equal to the one a single process computes."""
    mp.spawn(_worker_strong, args=(8, free_port(), str(tmp_path), 100), nprocs=8, join=True)
    bench = _bench()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    for name in ("imagenet4", "imagenetcrop4"):
        ids, lengths = bench.strong_plan(100, 1, name, 8)[0]
        streams = [np.arange(5 + n, dtype=np.uint32) * (c + 1) for c, n in zip(ids, lengths)]
        one = bench.gather_and_digest(_as_state_parts(streams), ids, 100, 0)
        assert one["crc32_of_streams_in_chain_order"] == open(tmp_path / f"digest_{name}_w8.txt").read()


def test_two_rank_gather_and_sharded_experiment(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # sharding is invisible in the results: same metrics as a single-process run
    import oracle as O
    from oracle.backend import OracleBackend
    from bitswap_amd import cli
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    one = cli.compress(6, 2, 1, 0, dataset="mnist", experiments=4, ndatapoints=2, decompress=False,
                       outdir=str(tmp_path / "single"), backend=OracleBackend(O.MODE_DET), small=8, verbose=False)
    two = np.load(tmp_path / "cmas_world2.npy")
    # conv outputs depend (in the last bit) on the batch a chain is coded with, so streams need not be
    # identical across shardings; the bit rates must agree closely and every chain must be present
    assert two.shape == one["cmas"].shape and np.all(two > 0)
    assert np.abs(two - one["cmas"]).max() < 0.5
    # ragged crop chains: with nn_batch the streams do not depend on the sharding at all
    from bitswap_amd import tiling
    ob = OracleBackend(O.MODE_DET)
    setup = cli.crop_setup(-1, nz=2, quantbits=6, backend=ob, small=8, nn_batch=2)
    rng = np.random.RandomState(5)
    blocks = [tiling.extract_blocks(rng.randint(0, 256, (32 * a, 32 * b, 3)).astype(np.uint8))[0]
              for a, b in ((3, 2), (1, 1), (1, 2), (1, 1))]
    res = cli.compress_images(blocks, quantbits=6, nz=2, setup=setup, backend=ob)
    assert np.array_equal(np.load(tmp_path / "crop_bpd_world2.npy"), np.array([r[2] for r in res]))
    assert np.load(tmp_path / "crop_words_world2.npy").tolist() == [len(r[0]) - 1 for r in res]


def _worker_n(rank, world, port, outdir):
    """Any world size: round-robin and LPT shards are a partition, the gathers put every chain in its place on rank 0 (ranks
    with different numbers of chains, streams of different lengths), the sharded experiment driver is lossless."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as td
    import oracle as O
    from oracle.backend import OracleBackend
    from bitswap_amd import cli, dist
    assert dist.init("gloo") == (rank, world)
    for nch, weights in ((7, None), (7, [9, 1, 1, 5, 1, 1, 3]), (2, None)):       # (2 chains on 3 ranks: one rank idle)
        mine = dist.shard_chains(nch, world, rank, weights=weights)
        owners = [dist.shard_chains(nch, world, r, weights=weights) for r in range(world)]
        assert sorted(sum(owners, [])) == list(range(nch)) and owners[rank] == mine
        local = [np.arange(3 + 2 * c, dtype=np.uint32) + 100 * c for c in mine]
        got = dist.gather_streams(local, mine, nch)
        rows = dist.gather_rows(np.array([[c, 0.5 * c] for c in mine], dtype=np.float64).reshape(len(mine), 2), mine, nch)
        if rank == 0:
            assert all(np.array_equal(got[c], np.arange(3 + 2 * c, dtype=np.uint32) + 100 * c) for c in range(nch))
            assert np.array_equal(rows, np.array([[c, 0.5 * c] for c in range(nch)]))
        else:
            assert got is None
    assert dist.allreduce_sum([1.0, float(rank)]) == [float(world), float(sum(range(world)))]
    res = cli.compress(6, 2, 1, 0, dataset="mnist", experiments=5, ndatapoints=2, decompress=True,
                       outdir=outdir, backend=OracleBackend(O.MODE_DET), small=8, verbose=False)
    if rank == 0:
        assert res["cmas"].shape == (5, 2) and np.all(res["cmas"] > 0)
        np.save(os.path.join(outdir, f"cmas_world{world}.npy"), res["cmas"])
    td.barrier()
    td.destroy_process_group()


def test_three_rank_shards_and_gathers(tmp_path):
    """World size 3 (the driver scales the bench over 1, 2, 4, 8 ranks: remainders and an idle rank must work for any count)."""
    port = free_port()
    mp.spawn(_worker_n, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert np.load(tmp_path / "cmas_world3.npy").shape == (5, 2)
    # 5 experiments on 3 ranks are coded 2 + 2 + 1 chains at a time; this (CPU) conv route is not batch-invariant, so the
    # stream record says so for every rank -- and the ranks' own --decompress legs accepted it (round 3: rank 2 used to be
    # refused against rank 0's record)
    import glob
    import json
    meta_file = glob.glob(str(tmp_path / "bitstreams" / "**" / "stream_meta.json"), recursive=True)[0]
    assert json.load(open(meta_file))["conv_route"]["chains_per_call"] == [2, 2, 1]
    # a receiver in ONE process: the record tells it how the sender was sharded, it decodes the shards one after the other
    # (a route that is not batch-invariant reproduces the streams only in the sender's batches) -- and refuses a record it
    # cannot reproduce
    import oracle as O
    from oracle.backend import OracleBackend
    from bitswap_amd import cli, meta
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = cli.decompress_streams(6, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=OracleBackend(O.MODE_DET), small=8,
                                 verbose=False)
    assert tuple(out.shape[:2]) == (5, 2)
    rec = json.load(open(meta_file))
    rec["world_size"] = 2                          # "written by two ranks": 3 + 2 chains per call, not what the record says
    json.dump(rec, open(meta_file, "w"))
    with pytest.raises(meta.StreamMismatch):
        cli.decompress_streams(6, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=OracleBackend(O.MODE_DET), small=8,
                               verbose=False)
