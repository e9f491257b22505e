"""Build libbitswap_hip.so in-tree with hipcc for gfx950 (no torch / pybind dependency: the
library is a plain C ABI, see include/bitswap_hip.h)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bitswap_hip.hip")
# one translation unit per kernel family (round 4: the 1,850-line bitswap_hip.hip with ~100 template instantiations took
# 20 s on one core; the units compile in parallel) around the shared device helpers of csrc/bitswap_dev.h
SRCS = [os.path.join(HERE, "csrc", f) for f in ("bitswap_hip.hip", "tables.hip", "pop.hip", "push.hip", "layer64.hip",
                                                 "net_epilogue.hip", "wino_gemm.hip", "wino_gemm_bf16x3.hip")]
HDR = os.path.join(HERE, "..", "include", "bitswap_hip.h")
DEV_HDR = os.path.join(HERE, "csrc", "bitswap_dev.h")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
PRODUCT_LIB = os.path.join(HERE, "csrc", "libbitswap_hip.so")

# -ffp-contract=off: the deterministic CDF spec forbids any fusion the source does not spell out
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-pass-failed"]   # K = 2048 rows cannot reach the occupancy hint of k_logistic; that is expected
EXTRA_FLAGS = [f for f in os.environ.get("BITSWAP_HIPCC_EXTRA", "").split() if f]   # e.g. -DBS_GEMM_LAB (tools/gemm_probe.py)
# Per-file flags.  net_epilogue.hip WITHOUT the SLP vectorizer: under plain -O3 hipcc packs the transforms' float32 additions into
# v_pk_add_f32 with op_sel shuffles (k_wino_fused<6,6>: 112 packed operations + the moves that line their operands up = 1,787
# instructions, 72 registers; without: 1,417 instructions, 66 registers, the same IEEE operations on the same values).  Beside
# the bf16 MFMA wavefronts of the bf16x3 GEMM one such packed addition lost its result in lanes 48..63 about once in 10^5
# workgroups (round 5, visits v-z: DESIGN 3.4); the packed form is slower beside MFMAs anyway.
# The -D is the source-level half of the guard: net_epilogue.hip refuses to compile (#error) unless the build says it turned the
# vectorizer off, so another build system cannot silently produce the packed code; tests/test_host_cpu.py
# disassembles the shipped library for packed float32 arithmetic on top of that.
FILE_FLAGS = {"net_epilogue.hip": ["-fno-slp-vectorize", "-DBS_BUILT_WITHOUT_SLP_VECTORIZER"]}


def _flag_hash():
    import hashlib
    per_file = [k + ":" + ",".join(v) for k, v in sorted(FILE_FLAGS.items())]
    return hashlib.md5(" ".join([f for f in HIPCC_FLAGS if f != "-shared"] + EXTRA_FLAGS + per_file).encode()).hexdigest()[:8]


# A build with extra flags (lab kernels that give WRONG results, selectable by environment) never lands on the product's
# path: its library carries the flag hash in its name, like its object directory -- so a later default-flag process cannot
# pick it up as "fresh", and a lab run never reuses the product library (round 4: both shared libbitswap_hip.so).
LIB = os.environ.get("BITSWAP_HIP_LIB") or (os.path.join(HERE, "csrc", f"libbitswap_hip_{_flag_hash()}.so") if EXTRA_FLAGS
                                             else PRODUCT_LIB)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libbitswap_hip.so cannot be built (set HIPCC=...)")


_warned_by_hand = False


def _stamp():
    return LIB + ".flags"


def is_stale():
    """Missing, older than a source, or built with another flag set (global, extra or per-file: the stamp written beside the
    library at link time holds the flag hash -- round 5's -fno-slp-vectorize is a correctness flag, not a tuning one)."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(p) > t for p in SRCS + [HDR, DEV_HDR] if os.path.exists(p)):
        return True
    if os.environ.get("BITSWAP_HIP_LIB"):       # a library named by hand (diagnostics) is taken as it is -- and says so once
        global _warned_by_hand
        if not _warned_by_hand:
            import sys
            print(f"bitswap_amd: BITSWAP_HIP_LIB={LIB}: a library named by hand, its build flags are NOT checked "
                  "(the product build needs -fno-slp-vectorize for net_epilogue.hip, DESIGN 3.4)", file=sys.stderr)
            _warned_by_hand = True
        return False
    try:
        return open(_stamp()).read().strip() != _flag_hash()
    except OSError:
        return True


ASAN_LIB = os.path.join(HERE, "csrc", "libbitswap_hip_asan.so")


def build_asan(verbose=False):
    """AddressSanitizer build of the same sources (host AND device code instrumented: ROCm's ASan needs an xnack+ target),
    written next to the product library as libbitswap_hip_asan.so.  Diagnostic only: needs an
    ASan-instrumented HIP runtime (/opt/rocm/lib/asan) and HSA_XNACK=1 to RUN -- this image ships none (the ASan runtime's
    hsa_amd_memory_pool_allocate interceptor aborts inside torch's bundled libamdhip64, profiles/r03x_asan.log), so round 3
    only establishes that the sources build instrumented; drive it from a plain HIP program (examples/c_abi_roundtrip.cpp),
    not from torch."""
    flags = [f if not f.startswith("--offload-arch") else "--offload-arch=gfx950:xnack+" for f in HIPCC_FLAGS]
    flags = [f for f in flags if f != "-O3"] + ["-O1", "-g", "-fsanitize=address", "-shared-libsan"] + FILE_FLAGS["net_epilogue.hip"]
    cmd = [hipcc_path()] + flags + ["-o", ASAN_LIB] + SRCS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return ASAN_LIB


def build_hip(force=False, verbose=False):
    """Compile the HIP library if missing or older than its sources: every translation unit to an object file (in parallel,
    only the stale ones), then one link.  Returns the .so path."""
    if not (force or is_stale()):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + EXTRA_FLAGS
    # objects of another flag set (tools/gemm_probe.py --lab appends -DBS_GEMM_LAB) never mix with the product's
    objdir = OBJ_DIR + "_" + _flag_hash()
    os.makedirs(objdir, exist_ok=True)
    deps = max(os.path.getmtime(p) for p in (HDR, DEV_HDR, os.path.abspath(__file__)))

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), deps):
            cmd = [hipcc_path()] + cflags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SRCS), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SRCS))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if not os.environ.get("BITSWAP_HIP_LIB"):
        with open(_stamp(), "w") as f:
            f.write(_flag_hash() + "\n")
    return LIB


# (outside csrc/: a known-bad library does not sit beside the product's)
PACKED_LIB = os.path.join(HERE, "..", "tools", "probes", "_build", "libbitswap_hip_slp.so")


def build_packed_epilogue(verbose=False):
    """DIAGNOSTICS ONLY: the product objects with ONE exception -- net_epilogue.hip compiled WITHOUT its per-file flag, i.e. with
    hipcc's SLP vectorizer and its packed float32 additions -- linked as libbitswap_hip_slp.so.  This is the library that
    reproduces the forked bf16x3 failure of round 5:
        BITSWAP_HIP_LIB=tools/probes/_build/libbitswap_hip_slp.so python tools/bf16x3_repro.py --record"""
    build_hip()
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + EXTRA_FLAGS
    objdir = OBJ_DIR + "_" + _flag_hash()
    src = os.path.join(HERE, "csrc", "net_epilogue.hip")
    packed = os.path.join(objdir, "net_epilogue.packed.o")
    cmd = [hipcc_path()] + cflags + ["-DBS_PACKED_EPILOGUE_DIAGNOSTIC", "-c", "-o", packed, src]
    os.makedirs(os.path.dirname(PACKED_LIB), exist_ok=True)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    objs = [packed if os.path.basename(s_) == "net_epilogue.hip" else os.path.join(objdir, os.path.basename(s_) + ".o") for s_ in SRCS]
    subprocess.check_call([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PACKED_LIB] + objs)
    return PACKED_LIB


if __name__ == "__main__":
    import sys
    print(build_asan(verbose=True) if "--asan" in sys.argv else build_packed_epilogue(verbose=True) if "--packed-epilogue" in sys.argv
          else build_hip(force=True, verbose=True))
