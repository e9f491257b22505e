"""CPU: the C-ABI library loads and exports every symbol include/bitswap_hip.h declares, and the
product path refuses to run without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bitswap_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bs_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from bitswap_amd import hip
    assert declared_symbols() == sorted(hip.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from bitswap_amd import build
    lib = ctypes.CDLL(build.build_hip())
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.bs_abi_version.restype = ctypes.c_int
    from bitswap_amd import hip
    assert lib.bs_abi_version() == hip.ABI_VERSION == 7
    lib.bs_strerror.restype = ctypes.c_char_p
    assert b"underflow" in lib.bs_strerror(1)


def test_argument_validation_needs_no_gpu():
    from bitswap_amd import hip
    L = hip.load()
    # null pointers / bad sizes are rejected on the host before any launch
    assert L.bs_rans_push(None, None, None, 0, None, None, 1, 1, 31, None, None) == hip.EINVAL
    assert L.bs_table_rows_f64(None, 1, 16, 31, 4, None, None, 17, None, None) == hip.EINVAL
    assert L.bs_logistic_tables(None, 0, None, 1, None, None, 0, 1, 1, 256, 31, 8, None, 260, 0, None, None) == hip.EINVAL
    assert L.bs_layer_pop64(None, None, None, 0, None, 0, None, 1, None, None, 0, 0, 1, 64, 256, 31, 8, None, None, 0, None,
                            None, None) == hip.EINVAL
    assert L.bs_layer_push64(None, None, None, 0, None, 0, None, 1, None, None, 0, 0, None, 1, 64, 256, 31, 8, None,
                             None) == hip.EINVAL
    # the uniform-bin CDF specs need the bin widths; a spec that does not exist is refused
    buf8 = (ctypes.c_char * 64)()
    q = ctypes.c_void_p((ctypes.addressof(buf8) + 15) & ~15)
    for spec in (2, 3, 0, 4):
        assert L.bs_logistic_tables(q, 0, None, spec, q, q, 0, 1, 1, 256, 31, 8, q, 260, 0, None, None) == hip.EINVAL
        assert L.bs_logistic_fc(q, 0, None, spec, q, q, 0, q, 1, 1, 256, 31, 8, q, q, q, None) == hip.EINVAL
    # the conv-stack products: null operands, a limb-product count that does not exist, a K that is not a multiple of 16
    assert L.bs_wino_gemm_f32(None, None, None, 36, 256, 256, 128, None) == hip.EINVAL
    assert L.bs_wino_gemm_bf16x3(None, None, None, 36, 256, 256, 128, 6, None) == hip.EINVAL
    buf = (ctypes.c_char * 64)()
    addr = (ctypes.addressof(buf) + 15) & ~15
    p = ctypes.c_void_p(addr)
    assert L.bs_wino_gemm_bf16x3(p, p, p, 36, 256, 256, 128, 7, None) == hip.EINVAL
    assert L.bs_wino_gemm_bf16x3(p, p, p, 36, 256, 250, 128, 6, None) == hip.EUNSUPPORTED
    assert L.bs_wino_gemm_bf16x3(p, p, p, 0, 256, 256, 128, 6, None) == hip.OK          # nothing to do: no launch
    assert L.bs_cdf_spec() == 4


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from bitswap_amd import hip
    from bitswap_amd.ans import ANS
    with pytest.raises(hip.BitswapHipError):
        ANS(torch.full((4, 16), 1 / 16, dtype=torch.float64))
    with pytest.raises(hip.BitswapHipError):
        hip.logistic_tables(torch.zeros(4, 255, dtype=torch.float64), torch.zeros(1, 4), torch.ones(1, 4))


def _build_c_example(tmp_path):
    """examples/c_abi_roundtrip.cpp: the library driven from plain C++/HIP through include/bitswap_hip.h only."""
    import subprocess
    from bitswap_amd import build
    lib = build.build_hip()
    exe = str(tmp_path / "c_abi_roundtrip")
    cmd = [build.hipcc_path(), "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_roundtrip.cpp"), "-L", os.path.dirname(lib), "-lbitswap_hip",
           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_c_example_compiles_against_the_header(tmp_path):
    assert os.path.exists(_build_c_example(tmp_path))


@pytest.mark.gpu
def test_c_example_round_trip_on_gpu(tmp_path):
    """No Python between the caller and the kernels: tables -> pop -> push restores every rANS state exactly, in the
    reference's stream format and in the 64-state format."""
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C_ABI_ROUNDTRIP_OK" in r.stdout, r.stdout + r.stderr
