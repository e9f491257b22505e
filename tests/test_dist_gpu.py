"""GPU: the branch of bitswap_amd/dist.py that RCCL takes, entered on hardware with a process group of ONE rank (the
driver's 8-GPU run is the first time more than one rank exists).  Runs in a subprocess so that the
process group of this test never leaks into the rest of the suite."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as td
    sys.path.insert(0, %r)
    from bitswap_amd import dist, workload
    from bitswap_amd.codec import BitSwapCodec
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
    torch.cuda.set_device(dist.local_device())
    td.init_process_group("nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
    assert td.get_backend() == "nccl" and dist._device().type == "cuda"
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    td.all_reduce(t)
    assert t.cpu().tolist() == list(range(8))
    # real streams: two chains of an MNIST-shaped model coded on this GPU, gathered through the collective branch
    model, zend, zcen = workload.build("mnist2", "cuda", quantbits=8, small=16)
    images = workload.synthetic_blocks(4, model.xs, seed=3).view(2, 2, -1).to(torch.int32).cuda()
    state, met = BitSwapCodec(model, zend, zcen, quantbits=8).compress(images, nwords=2000)
    streams = [np.array(s[:-1] + [s[-1] & 0xffffffff, s[-1] >> 32], dtype=np.uint32) for s in state.to_lists()]
    got = dist.gather_streams(streams, [1, 0], 2, collective=True)          # ids out of order on purpose
    assert np.array_equal(got[1], streams[0]) and np.array_equal(got[0], streams[1])
    # the same streams packed ON THE DEVICE and handed to RCCL as a device tensor (no host round trip in front of the gather)
    dev = dist.gather_streams_device((state.stack, state.len, state.head), [1, 0], 2, collective=True)
    assert np.array_equal(dev[1], streams[0]) and np.array_equal(dev[0], streams[1])
    flat, words = dist.pack_streams_device(state.stack, state.len, state.head)
    assert flat.is_cuda and words.cpu().tolist() == [len(a) for a in streams]
    rows = dist.gather_rows(met["cma"], [1, 0], 2, collective=True)
    assert np.array_equal(rows[1], met["cma"][0]) and np.array_equal(rows[0], met["cma"][1])
    assert dist.allreduce_sum([1.5, 2.0], collective=True) == [1.5, 2.0]
    td.barrier()
    td.destroy_process_group()
    print("RCCL_ONE_RANK_OK", sum(len(a) for a in got))
''')


def test_rccl_branch_with_one_rank(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER % ROOT, str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def _bench(args, env=None, timeout=900):
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-extra", "--no-cpu-baseline", "--no-roofline"],
                       capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {})))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.stdout + r.stderr)[-3000:]
    return json.loads(lines[-1])


def test_bench_strong_scaling_mode_one_and_two_ranks():
    """bench.py --scaling strong (BASELINE configs 4 / 5: the chains are a fixed total, sharded by dist.shard_chains): at
    N = 1 and with two ranks on this one device (gloo: control flow of the sharded run, the gather and the max-over-ranks
    timing; RCCL refuses two ranks per GPU) the line says `strong`, is lossless, and -- every convolution runs on
    batch-invariant kernels of our own -- the gathered streams are THE SAME whichever rank coded them (one digest)."""
    base = ["--scaling", "strong", "--total-chains", "6", "--workload", "mnist2", "--steps", "2", "--warmup", "1"]
    one = _bench(base)
    assert one["scaling"] == "strong" and one["lossless"] and one["n_gpus"] == 1
    assert one["config"]["chains_per_rank"] == [6] and one["config"]["total_chains"] == 6
    assert one["stream_gather"]["complete"] and one["value"] > 0
    two = _bench(base + ["--gpus", "2"], env={"BENCH_DIST_BACKEND": "gloo"})
    assert two["scaling"] == "strong" and two["lossless"] and two["n_gpus"] == 2 and two["config"]["chains_per_rank"] == [3, 3]
    assert two["stream_gather"]["complete"] and two["stream_gather"]["chains"] == 6
    assert two["stream_gather"]["crc32_of_streams_in_chain_order"] == one["stream_gather"]["crc32_of_streams_in_chain_order"]
    # the weak default is untouched by the new mode
    weak = _bench(["--chains", "4", "--workload", "mnist2", "--steps", "2", "--warmup", "1"])
    assert weak["scaling"] == "weak" and weak["lossless"] and weak["config"]["chains_per_gpu"] == 4


def test_bench_strong_scaling_ragged_crop_chains():
    """Config 4's shape through bench.py: ragged image chains (LPT-sharded), the crop model on its fixed conv micro-batch;
    lossless, every chain unwound, blocks counted over all ranks; two ranks give the digest of one."""
    base = ["--scaling", "strong", "--total-chains", "5", "--workload", "imagenetcrop4", "--steps", "4", "--warmup", "1"]
    one = _bench(base)
    assert one["scaling"] == "strong" and one["lossless"] and "configs[3]" in one["config"]["workload"]
    assert one["config"]["blocks_total"] == sum(one["config"]["blocks_per_rank"]) > 5
    two = _bench(base + ["--gpus", "2"], env={"BENCH_DIST_BACKEND": "gloo"})
    assert two["lossless"] and two["config"]["blocks_total"] == one["config"]["blocks_total"]
    assert two["stream_gather"]["crc32_of_streams_in_chain_order"] == one["stream_gather"]["crc32_of_streams_in_chain_order"]


def test_bench_strong_scaling_eight_ranks_gather_on_one_device():
    """the one collective of the design end to end at the world size north_star's curve ends at.  `bench.py --gpus 8
    --scaling strong` (BASELINE configs[1]'s 100 chains: 13, 13, 13, 13, 12, 12, 12, 12 per rank) with BENCH_DIST_BACKEND=gloo --
    eight ranks sharing this one device (RCCL refuses two ranks per GPU) -- codes, packs every rank's streams ON THE DEVICE
    (dist.pack_streams_device), gathers them to rank 0 and reports the CRC-32 of all 100 streams in chain order: it must be the
    CRC of the 1-rank run (every conv kernel is batch-invariant, inputs and initial words are functions of the global chain id)."""
    base = ["--scaling", "strong", "--total-chains", "100", "--steps", "2", "--warmup", "1"]
    one = _bench(base)
    assert one["scaling"] == "strong" and one["lossless"] and one["config"]["chains_per_rank"] == [100]
    g1 = one["stream_gather"]
    assert g1["complete"] and g1["chains"] == 100 and g1["bytes"] > 100 * 2 * 3072
    eight = _bench(base + ["--gpus", "8"], env={"BENCH_DIST_BACKEND": "gloo"}, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["lossless"] and eight["rccl_ranks"] == 0
    assert eight["config"]["chains_per_rank"] == [13, 13, 13, 13, 12, 12, 12, 12]
    g8 = eight["stream_gather"]
    assert g8["complete"] and g8["chains"] == 100 and g8["bytes"] == g1["bytes"]
    assert g8["crc32_of_streams_in_chain_order"] == g1["crc32_of_streams_in_chain_order"]
