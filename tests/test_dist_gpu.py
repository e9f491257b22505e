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
