import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bitswap_amd import cli, tiling
sys.path.insert(0, "/root/repo")
import importlib.util
spec = importlib.util.spec_from_file_location("crop", "/root/repo/imagenetcrop_compress.py"); crop = importlib.util.module_from_spec(spec); spec.loader.exec_module(crop)
imgs = crop.synthetic_images(100)
blocks = [tiling.extract_blocks(a)[0] for a in imgs]
for nb in (16, 32, 64):
    setup = cli.crop_setup(0, nz=4, quantbits=10, synthetic=True, nn_batch=nb)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = cli.compress_images(blocks, quantbits=10, nz=4, bitswap=1, gpu=0, hwc_quirk=True, setup=setup)
        dt = time.perf_counter() - t0
    n = sum(len(b) for b in blocks)
    print(f"nn_batch {nb}: {n} blocks in {dt:.2f} s = {n*1024/dt/1e6:.2f} Mpx/s sender; bpd {np.mean([o[2] for o in out]):.3f}", flush=True)
    # single-image decode cost with this nn_batch
    st, mw, _ = out[0]
    from bitswap_amd import container
    arr = container.pack(st, mw, len(blocks[0]), 32, 32)
    st2, nbk, _, _ = container.unpack(arr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o, rest = cli.decompress_image(st2, nbk, quantbits=10, nz=4, setup=setup, hwc_quirk=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   single-image decode ({nbk} blocks): {dt:.2f} s, ok={np.array_equal(o, blocks[0])}", flush=True)
