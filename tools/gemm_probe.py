"""bs_wino_gemm_f32 (persistent balanced kernel, round 3) against its round-2 version (tiled launch; built on the spot from
tools/probes/wino_gemm_r02.hip) and the library (torch.bmm on the backend the model used to use) for the batched shapes of
the bench: sustained loops, TFLOP/s, fraction of the 157.3 TFLOP/s fp32 MFMA peak.
usage: python tools/gemm_probe.py [--quick] [--only-own] [--knobs] [--lab]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lab" in sys.argv:      # the timing-only variants live behind -DBS_GEMM_LAB: a library of their own, outside the tree
    os.environ["BITSWAP_HIP_LIB"] = "/tmp/libbitswap_hip_lab.so"
from bitswap_amd import build, hip  # noqa: E402
if "--lab" in sys.argv:
    build.HIPCC_FLAGS.append("-DBS_GEMM_LAB")
    build.build_hip(force=True)

PEAK = 157.3


def timeit(fn, warm=100, reps=200):
    """Long loops: the clock of an MI355X follows the load of the last tens of milliseconds, a 3 + 30 launch loop measures
    the previous variant's clock as much as this variant's cycles (round-3 visit B: 93 vs 113 TFLOP/s for the same cycle
    count, by position in the loop)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def old_kernel():
    so = "/tmp/libgemm_r02.so"
    src = os.path.join(ROOT, "tools", "probes", "wino_gemm_r02.hip")
    try:
        subprocess.check_call([build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
        L = C.CDLL(so)
        L.bs_wino_gemm_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_int64, C.c_void_p]
        return L
    except Exception as e:   # noqa
        print("round-2 kernel unavailable:", e)
        return None


quick, only_own = "--quick" in sys.argv, "--only-own" in sys.argv
try:
    torch.backends.cuda.preferred_blas_library("ck")
except Exception as e:   # noqa
    print("ck backend unavailable", e)
old = None if only_own else old_kernel()
shapes = ((36, 256, 256, 6400), (36, 256, 256, 1600), (64, 256, 256, 6400), (64, 256, 256, 1600), (36, 256, 256, 25600),
          (36, 256, 256, 7168), (36, 256, 256, 512), (36, 256, 256, 208), (36, 16, 256, 6400), (36, 16, 256, 1600), (36, 24, 256, 512))
if quick:
    shapes = shapes[:1]
if "--lab" in sys.argv:
    shapes = (shapes[0], shapes[4])
for T, Cout, Cin, cols in shapes:
    U = torch.randn(T, Cout, Cin, device="cuda")
    V = torch.randn(T, Cin, cols, device="cuda")
    out = torch.empty(T, Cout, cols, device="cuda")
    fl = 2.0 * T * Cout * Cin * cols
    line = f"T{T} Cout{Cout} Cin{Cin} cols{cols}:"
    def own_with(**env):
        def run():
            os.environ.update(env)
            hip.wino_gemm(U, V, out=out)
            for k in env:
                os.environ.pop(k, None)
        return run
    variants = [("own", lambda: hip.wino_gemm(U, V, out=out))]
    if "--knobs" in sys.argv:
        variants += [("v0", own_with(BITSWAP_GEMM_VARIANT="0")), ("even-ranges", own_with(BITSWAP_GEMM_EVEN_RANGES="1")), ("v3", own_with(BITSWAP_GEMM_VARIANT="3")),
                     ("1wg", own_with(BITSWAP_GEMM_WGS_PER_CU="1"))]
    if "--lab" in sys.argv and Cout == 256:      # timing-only variants (results wrong): what each piece of the K step costs
        variants += [(n, own_with(BITSWAP_GEMM_VARIANT=str(v))) for n, v in
                     (("no-vmcnt", 17), ("no-barrier", 19), ("no-stores", 20), ("no-dma", 24), ("mfma+lds", 31), ("no-A-dma", 32),
                      ("no-B-dma", 48), ("plain-stores", 80))]
    ref = None
    if old is not None and Cout >= 64:
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ref = torch.empty_like(out)

        def run_old():
            assert old.bs_wino_gemm_f32(U.data_ptr(), V.data_ptr(), ref.data_ptr(), T, Cout, Cin, cols, st) == 0
        variants.append(("r02", run_old))
    if not only_own:
        variants.append(("library", lambda: torch.bmm(U, V, out=out)))
    best = {}
    for _ in range(2):                       # round-robin, two passes: the second is the one reported
        for name, fn in variants:
            best[name] = timeit(fn, *((20, 20) if quick else (100, 200)))
    for name, _ in variants:
        t = best[name]
        line += f" {name} {t * 1e6:8.1f} us {fl / t / 1e12:6.1f} TF ({fl / t / 1e12 / PEAK:.2f}) |"
    if ref is not None:
        hip.wino_gemm(U, V, out=out)
        line += f" own == r02 bitwise: {bool(torch.equal(out, ref))}"
    print(line, flush=True)
