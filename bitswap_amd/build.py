"""Build libbitswap_hip.so in-tree with hipcc for gfx950 (no torch / pybind dependency: the
library is a plain C ABI, see include/bitswap_hip.h)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bitswap_hip.hip")
SRCS = [SRC, os.path.join(HERE, "csrc", "net_epilogue.hip"), os.path.join(HERE, "csrc", "wino_gemm.hip")]
HDR = os.path.join(HERE, "..", "include", "bitswap_hip.h")
LIB = os.path.join(HERE, "csrc", "libbitswap_hip.so")

# -ffp-contract=off: the deterministic CDF spec forbids any fusion the source does not spell out
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-pass-failed"]   # K = 2048 rows cannot reach the occupancy hint of k_logistic; that is expected


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libbitswap_hip.so cannot be built (set HIPCC=...)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRCS + [HDR] if os.path.exists(p))


def build_hip(force=False, verbose=False):
    """Compile the HIP library if missing or older than its sources.  Returns the .so path."""
    if force or is_stale():
        cmd = [hipcc_path()] + HIPCC_FLAGS + ["-o", LIB] + SRCS
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
