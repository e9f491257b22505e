"""ctypes binding of libbitswap_hip.so (include/bitswap_hip.h) for torch tensors.

There is NO fallback: if the library is missing or there is no GPU, every function here
raises.  torch is used only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import numpy as np

import torch

from . import build as _build

ABI_VERSION = 7   # include/bitswap_hip.h BS_ABI_VERSION this binding was written against
OK, EINVAL, EUNSUPPORTED, ELAUNCH = 0, -1, -2, -3
ST_OK, ST_UNDERFLOW, ST_OVERFLOW, ST_BADTABLE, ST_BADSYMBOL = 0, 1, 2, 3, 4
PARAM_F32, PARAM_F64 = 0, 1
LAYOUT_LINEAR, LAYOUT_WAVE, LAYOUT_PIVOT = 0, 1, 2

SYMBOLS = [
    "bs_abi_version", "bs_cdf_spec", "bs_strerror", "bs_table_rows_f64", "bs_logistic_tables",
    "bs_logistic_fc", "bs_rans_push", "bs_rans_push_table", "bs_rans_pop", "bs_rans_pop_pivot", "bs_gather_centres", "bs_layer_pop64",
    "bs_layer_push64", "bs_stream_create_cu_mask", "bs_stream_destroy", "bs_debug_where",
    "bs_selftest", "bs_sigmoid_f64", "bs_bias_residual_elu_f32", "bs_head_params_f32", "bs_expand_rows5_f32", "bs_wino_in_f32", "bs_wino_out_f32", "bs_wino_fused_f32",
    "bs_small_k_gemm_f32", "bs_conv3_wino_f32", "bs_wino_gemm_f32", "bs_wino_gemm_bf16x3",
]
HEAD_SIGMOID, HEAD_SOFTPLUS = 0, 1


class BitswapHipError(RuntimeError):
    pass


_lib = None


def lib_path():
    return _build.LIB


def load():
    """dlopen libbitswap_hip.so (must have been built: `python -m bitswap_amd.build` or
    __graft_entry__.build()).  Raises if absent -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise BitswapHipError(
            f"{path} not found: build it with `python -m bitswap_amd.build` (hipcc, gfx950). "
            "bitswap_amd has no CPU fallback for the entropy-coding path.")
    L = C.CDLL(path)
    p, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.bs_abi_version.restype = i32
    L.bs_cdf_spec.restype = i32
    L.bs_strerror.restype = C.c_char_p
    L.bs_strerror.argtypes = [i32]
    L.bs_table_rows_f64.argtypes = [p, i64, i32, i32, i32, p, p, i64, p, p]
    L.bs_logistic_tables.argtypes = [p, i64, p, i32, p, p, i32, i32, i32, i32, i32, i32, p, i64, i32, p, p]
    L.bs_logistic_fc.argtypes = [p, i64, p, i32, p, p, i32, p, i32, i32, i32, i32, i32, p, p, p, p]
    L.bs_rans_push.argtypes = [p, p, p, i64, p, p, i32, i32, i32, p, p]
    L.bs_rans_push_table.argtypes = [p, p, p, i64, p, i64, i64, i32, p, i32, i32, i32, i32, p, p]
    L.bs_rans_pop.argtypes = [p, p, p, i64, p, i64, i64, i32, i32, i32, i32, i32, p, p, i64, p, p, p]
    L.bs_rans_pop_pivot.argtypes = [p, p, p, i64, p, i64, p, i64, p, i32, p, p, i32, i32, i32, i32, i32, i32, p, p, i64, p, p, p]
    L.bs_gather_centres.argtypes = [p, i64, p, i32, i32, i32, p, p]
    L.bs_layer_pop64.argtypes = [p, p, p, i64, p, i64, p, i32, p, p, i64, i32, i32, i32, i32, i32, i32, p, p, i64, p, p, p]
    L.bs_stream_create_cu_mask.argtypes = [i32, i32, p]
    L.bs_stream_destroy.argtypes = [p]
    L.bs_debug_where.argtypes = [p, i32, i32, p]
    L.bs_layer_push64.argtypes = [p, p, p, i64, p, i64, p, i32, p, p, i64, i32, p, i32, i32, i32, i32, i32, p, p]
    L.bs_selftest.argtypes = [C.POINTER(C.c_int64), p]
    L.bs_sigmoid_f64.argtypes = [p, i64, p, p]
    L.bs_bias_residual_elu_f32.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    L.bs_head_params_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, p]
    L.bs_expand_rows5_f32.argtypes = [p, p, p, i64, i32, i32, i32, i32, p]
    L.bs_wino_fused_f32.argtypes = [p, i32, p, p, i32, p, p, p, i32, i64, i32, i32, i32, p]
    L.bs_wino_in_f32.argtypes = [p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.bs_small_k_gemm_f32.argtypes = [p, p, p, i32, i32, i32, i64, p]
    L.bs_wino_gemm_f32.argtypes = [p, p, p, i32, i32, i32, i64, p]
    L.bs_wino_gemm_bf16x3.argtypes = [p, p, p, i32, i32, i32, i64, i32, p]
    L.bs_conv3_wino_f32.argtypes = [p, p, p, i32, p, p, i32, i64, i32, i32, i32, i32, p]
    L.bs_wino_out_f32.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    for n in SYMBOLS:
        if n != "bs_strerror":
            getattr(L, n).restype = i32
    if L.bs_abi_version() != ABI_VERSION:
        raise BitswapHipError(f"{path} has ABI version {L.bs_abi_version()}, this binding needs {ABI_VERSION}: "
                              "rebuild with `python -m bitswap_amd.build`")
    _lib = L
    return L


def _check(rc, what):
    if rc != OK:
        msg = load().bs_strerror(rc).decode()
        raise BitswapHipError(f"{what}: {msg} (code {rc})")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    """Every tensor on the CURRENT HIP device: the C side launches on the current device's stream and never
    calls hipSetDevice, so a tensor of another device would be touched by a kernel of the wrong GPU."""
    cur = torch.cuda.current_device() if torch.cuda.is_available() else -1
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise BitswapHipError("bitswap_amd kernels need tensors on a HIP device (no CPU fallback)")
        if t.device.index != cur:
            raise BitswapHipError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: call "
                                  "torch.cuda.set_device() first (one process per GPU)")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _param_dtype(t):
    if t.dtype == torch.float32:
        return PARAM_F32
    if t.dtype == torch.float64:
        return PARAM_F64
    raise BitswapHipError(f"mu/scale must be float32 or float64, got {t.dtype}")


def _row_stride(t, width):
    """Endpoint / centre tables: [D, width] with unit inner stride; row stride may be 0 (expanded)."""
    if t.dim() != 2 or t.shape[1] != width or t.dtype != torch.float64:
        raise BitswapHipError(f"expected float64 [D,{width}] table, got {tuple(t.shape)} {t.dtype}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t, t.stride(0)


def aligned_ld(K):
    """Row stride (uint32 units) selecting 16-byte stores/loads: K+1 entries rounded up to 4."""
    return (K + 1 + 3) // 4 * 4


def wave_ld(K):
    """Row stride of BS_LAYOUT_WAVE rows: K permuted entries + 64 pivots."""
    return K + 64


def table_layout(cdf, K):
    """Layout of a cdf tensor made by logistic_tables(): tensors carry it as the attribute
    `bs_layout`; plain tensors (tests, ANS) are linear."""
    return getattr(cdf, "bs_layout", LAYOUT_LINEAR)


def wave_supported(K):
    return K in (256, 512, 1024, 2048)


PIVOT_LD = 128      # BS_LAYOUT_PIVOT: 64 x (cumulative value, aux) words per row
PIVOT_MAX_D = 15872  # BS_POP_PIVOT_MAX_D (include/bitswap_hip.h): symbols + the 2 KB parameter block in 64 KB of LDS


def pivot_supported(K, D):
    """bs_rans_pop_pivot takes K = 256 .. 2048 and whole 64-symbol chunks (uniform-width bins only: the caller checks)."""
    return K in (256, 512, 1024, 2048) and D % 64 == 0 and D <= PIVOT_MAX_D


def selftest():
    n = C.c_int64(-1)
    _check(load().bs_selftest(C.byref(n), _stream()), "bs_selftest")
    return n.value


def sigmoid_f64(t):
    _need_cuda(t)
    t = t.contiguous().to(torch.float64)
    out = torch.empty_like(t)
    _check(load().bs_sigmoid_f64(_ptr(t), t.numel(), _ptr(out), _stream()), "bs_sigmoid_f64")
    return out


def table_rows(pmf, bits=31, quantbits=8, ld=None, want_f=True):
    """ANS.__init__ on pmf [rows,K] f64 -> (f [rows,K] int32 | None, cdf [rows,ld] int32, status [rows])."""
    _need_cuda(pmf)
    pmf = pmf.contiguous().to(torch.float64)
    rows, K = pmf.shape
    ld = ld or (K + 1)
    f = torch.empty((rows, K), dtype=torch.int32, device=pmf.device) if want_f else None
    cdf = torch.empty((rows, ld), dtype=torch.int32, device=pmf.device)
    status = torch.zeros(rows, dtype=torch.int32, device=pmf.device)
    _check(load().bs_table_rows_f64(_ptr(pmf), rows, K, bits, quantbits, _ptr(f), _ptr(cdf), ld, _ptr(status),
                                    _stream()), "bs_table_rows_f64")
    return f, cdf, status


def _step(step, D):
    if step is None:
        return None
    if step.dtype != torch.float64 or tuple(step.shape) != (D,) or not step.is_contiguous():
        raise BitswapHipError(f"bin_step must be a contiguous float64 [{D}] tensor")
    return step


UNIFORM_CDF_SPECS = (2, 3, 4)
from .meta import DEFAULT_CDF_SPEC as UNIFORM_CDF_SPEC   # CDF spec of rows of uniform-width bins when the caller names none (include/bitswap_hip.h BS_CDF_SPEC)


def _spec(step, spec):
    """The `cdf_spec` argument of the C ABI: 1 without bin widths; with them 2, 3 or 4 as asked (default UNIFORM_CDF_SPEC).
    Sender and receiver must make the same choice: the codec records it in the stream fingerprint (meta.py)."""
    if step is None:
        if spec not in (None, 1):
            raise BitswapHipError(f"CDF spec {spec} needs the bin widths (step)")
        return 1
    spec = UNIFORM_CDF_SPEC if spec is None else int(spec)
    if spec not in UNIFORM_CDF_SPECS:
        raise BitswapHipError(f"CDF spec {spec} is not one of the uniform-bin specifications {UNIFORM_CDF_SPECS}")
    return spec


def logistic_tables(endpoints, mu, scale, bits=31, quantbits=10, ld=None, out=None, layout=LAYOUT_LINEAR, step=None,
                    status=None, spec=None):
    """Fused CDF -> integer cdf rows.  endpoints [D,K-1] f64, mu/scale [B,D] -> cdf [B,D,ld] int32.
    layout=LAYOUT_WAVE writes the wave-native hand-off format (ld = K+64) that rans_pop searches
    with two ballots; the returned tensor remembers its layout (`bs_layout`).  step [D] f64 (bins.uniform_step)
    selects the CDF specs of uniform-width bins (`spec` 2 or 3, default UNIFORM_CDF_SPEC); status [B] int32 receives
    BS_ST_BADTABLE for degenerate parameters."""
    _need_cuda(endpoints, mu, scale, step, status)
    B, D = mu.shape
    K = endpoints.shape[1] + 1
    endpoints, es = _row_stride(endpoints, K - 1)
    step = _step(step, D)
    spec = _spec(step, spec)
    mu, scale = mu.contiguous(), scale.contiguous()
    if out is not None:
        ld = out.shape[-1]
    ld = ld or (wave_ld(K) if layout == LAYOUT_WAVE else PIVOT_LD if layout == LAYOUT_PIVOT else aligned_ld(K))
    if out is None:
        out = torch.empty((B, D, ld), dtype=torch.int32, device=mu.device)
    _check(load().bs_logistic_tables(_ptr(endpoints), es, _ptr(step), spec, _ptr(mu), _ptr(scale), _param_dtype(mu), B, D, K,
                                     bits, quantbits, _ptr(out), ld, layout, _ptr(status), _stream()),
           "bs_logistic_tables")
    out.bs_layout = layout
    if layout == LAYOUT_PIVOT:      # the pop kernel rebuilds rows from what the table kernel saw: keep it with the hand-off
        out.bs_pivot_args = (endpoints, es, step, spec, mu, scale, quantbits)
    return out


def logistic_fc(endpoints, mu, scale, sym, status, bits=31, quantbits=10, out=None, step=None, spec=None):
    """Fused CDF -> (f, c) of the given symbols.  sym [B,D] int32 -> f, c [B,D] int32."""
    _need_cuda(endpoints, mu, scale, sym, status, step)
    B, D = mu.shape
    K = endpoints.shape[1] + 1
    endpoints, es = _row_stride(endpoints, K - 1)
    step = _step(step, D)
    spec = _spec(step, spec)
    mu, scale = mu.contiguous(), scale.contiguous()
    sym = sym.contiguous()
    if sym.dtype != torch.int32:
        sym = sym.to(torch.int32)
    if out is None:
        f = torch.empty((B, D), dtype=torch.int32, device=mu.device)
        c = torch.empty((B, D), dtype=torch.int32, device=mu.device)
    else:
        f, c = out
    _check(load().bs_logistic_fc(_ptr(endpoints), es, _ptr(step), spec, _ptr(mu), _ptr(scale), _param_dtype(mu), _ptr(sym), B,
                                 D, K, bits, quantbits, _ptr(f), _ptr(c), _ptr(status), _stream()), "bs_logistic_fc")
    return f, c


class RansState:
    """B independent rANS states resident in HBM.

    head [B] (uint64 bits in an int64 tensor), stack [B,cap] (uint32 bits in int32), len [B]
    int32, status [B] int32.  One chain is the reference's Python list `state`
    (mnist_compress.py:158-159): stack words followed by the 64-bit head.
    """

    def __init__(self, B, cap, device):
        self.B, self.cap, self.device = B, int(cap), torch.device(device)
        self.head = torch.zeros(B, dtype=torch.int64, device=device)
        self.stack = torch.zeros((B, self.cap), dtype=torch.int32, device=device)
        self.len = torch.zeros(B, dtype=torch.int32, device=device)
        self.status = torch.zeros(B, dtype=torch.int32, device=device)

    def prefix(self, k):
        """The first k chains as a RansState sharing this one's memory (chains sorted by decreasing
        length drop out of a lock-step run from the back: codec.compress_ragged)."""
        if k == self.B:
            return self
        cache = self.__dict__.setdefault("_prefixes", {})
        v = cache.get(int(k))        # the same view object every time: a captured block step (hipGraph) pins its tensors
        if v is None:
            v = object.__new__(RansState)
            v.B, v.cap, v.device = int(k), self.cap, self.device
            v.head, v.stack, v.len, v.status = self.head[:k], self.stack[:k], self.len[:k], self.status[:k]
            cache[int(k)] = v
        ml = getattr(self, "min_len", None)
        if ml is not None and getattr(v, "min_len", None) is None:
            v.min_len = ml[:k]
        return v

    @classmethod
    def from_lists(cls, states, cap=None, device="cuda"):
        """states: list of B Python lists [w0, ..., w_{n-1}, head]."""
        import numpy as np
        B = len(states)
        n = max(len(s) - 1 for s in states)
        cap = max(int(cap or 0), n)
        st = cls(B, cap, device)
        stack = np.zeros((B, cap), dtype=np.uint32)
        head = np.zeros(B, dtype=np.uint64)
        ln = np.zeros(B, dtype=np.int32)
        for b, s in enumerate(states):
            k = len(s) - 1
            stack[b, :k] = np.asarray(s[:-1], dtype=np.uint64).astype(np.uint32)
            head[b] = s[-1]
            ln[b] = k
        st.stack.copy_(torch.from_numpy(stack.view(np.int32)))
        st.head.copy_(torch.from_numpy(head.view(np.int64)))
        st.len.copy_(torch.from_numpy(ln))
        return st

    def to_lists(self):
        import numpy as np
        stack = self.stack.cpu().numpy().view(np.uint32)
        head = self.head.cpu().numpy().view(np.uint64)
        ln = self.len.cpu().numpy()
        return [[int(w) for w in stack[b, : ln[b]]] + [int(head[b])] for b in range(self.B)]

    def check(self, what="rANS"):
        """Synchronising: raise if any chain reported a device-side error."""
        st = self.status.cpu()
        if int(st.abs().max()) != 0:
            b = int(torch.nonzero(st)[0])
            code = int(st[b])
            raise BitswapHipError(f"{what}: chain {b}: {load().bs_strerror(code).decode()} (status {code})")


def rans_push(state, f, c, bits=31):
    _need_cuda(f, c)
    B, D = f.shape
    _check(load().bs_rans_push(_ptr(state.head), _ptr(state.stack), _ptr(state.len), state.cap, _ptr(f), _ptr(c),
                               B, D, bits, _ptr(state.status), _stream()), "bs_rans_push")


def rans_push_table(state, cdf, sym, K, bits=31, layout=None):
    """cdf [B,D,ld] (per chain) or [D,ld] (shared by all chains); sym [B,D] int32."""
    _need_cuda(cdf, sym)
    layout = table_layout(cdf, K) if layout is None else layout
    sym = sym.contiguous()
    if sym.dtype != torch.int32:
        sym = sym.to(torch.int32)
    B, D = sym.shape
    if not cdf.is_contiguous():
        cdf = cdf.contiguous()
    ld = cdf.shape[-1]
    chain_stride = 0 if cdf.dim() == 2 else D * ld
    _check(load().bs_rans_push_table(_ptr(state.head), _ptr(state.stack), _ptr(state.len), state.cap, _ptr(cdf),
                                     chain_stride, ld, layout, _ptr(sym), B, D,
                                     K, bits, _ptr(state.status), _stream()),
           "bs_rans_push_table")


def rans_pop(state, cdf, K, bits=31, centres=None, B=None, layout=None):
    """Pop D symbols per chain.  Returns (sym [B,D] int32, z [B,D] float32 | None)."""
    _need_cuda(cdf, centres)
    layout = table_layout(cdf, K) if layout is None else layout
    if layout == LAYOUT_PIVOT:
        return rans_pop_pivot(state, cdf, K, bits, centres=centres)
    if not cdf.is_contiguous():
        cdf = cdf.contiguous()
    ld = cdf.shape[-1]
    if cdf.dim() == 2:
        D, chain_stride, B = cdf.shape[0], 0, (B or state.B)
    else:
        B, D = cdf.shape[0], cdf.shape[1]
        chain_stride = D * ld
    sym = torch.empty((B, D), dtype=torch.int32, device=cdf.device)
    z, cs = None, 0
    if centres is not None:
        centres, cs = _row_stride(centres, K)
        z = torch.empty((B, D), dtype=torch.float32, device=cdf.device)
    _check(load().bs_rans_pop(_ptr(state.head), _ptr(state.stack), _ptr(state.len), state.cap, _ptr(cdf),
                              chain_stride, ld, layout, B, D, K, bits, _ptr(sym), _ptr(centres), cs, _ptr(z),
                              _ptr(state.status), _stream()), "bs_rans_pop")
    return sym, z


def rans_pop_pivot(state, piv, K, bits=31, centres=None):
    """Pop D symbols per chain from BS_LAYOUT_PIVOT rows made by logistic_tables(layout=LAYOUT_PIVOT): the kernel rebuilds
    the symbol's group of bins from the (endpoints, step, mu, scale) the table kernel used (remembered on `piv`).
    Returns (sym [B,D] int32, z [B,D] float32 | None)."""
    endpoints, es, step, spec, mu, scale, quantbits = piv.bs_pivot_args
    _need_cuda(piv, centres, endpoints, step, mu, scale)
    assert piv.is_contiguous() and piv.dim() == 3
    B, D, ld = piv.shape
    sym = torch.empty((B, D), dtype=torch.int32, device=piv.device)
    z, cs = None, 0
    if centres is not None:
        centres, cs = _row_stride(centres, K)
        z = torch.empty((B, D), dtype=torch.float32, device=piv.device)
    _check(load().bs_rans_pop_pivot(_ptr(state.head), _ptr(state.stack), _ptr(state.len), state.cap, _ptr(piv), ld,
                                    _ptr(endpoints), es, _ptr(step), spec, _ptr(mu), _ptr(scale), _param_dtype(mu), B, D, K, bits,
                                    quantbits, _ptr(sym), _ptr(centres), cs, _ptr(z), _ptr(state.status), _stream()),
           "bs_rans_pop_pivot")
    return sym, z


# ---- BS_FORMAT_WAVE64: 64 rANS states per chain, table + coding step fused (include/bitswap_hip.h) -----------------
NSTATES = 64


def split_state(s, nstates=NSTATES):
    """A reference-style state list [w0 .. w_{n-2}, head] (head = last initial word << 32, mnist_compress.py:158-159)
    dealt round-robin onto `nstates` sub-states of the same form: sub-state j gets the words w[j::nstates], its last
    one shifted up as its head."""
    words = list(s[:-1]) + [s[-1] >> 32]
    out = []
    for j in range(nstates):
        sub = words[j::nstates]
        assert len(sub) >= 2, "too few initial words for the 64-state format"
        out.append(sub[:-1] + [sub[-1] << 32])
    return out


class RansState64:
    """B chains x 64 independent rANS states resident in HBM: head [B,64] (uint64 bits in int64), stack [B,64,cap]
    (uint32 bits in int32), len64 [B,64] int32, status [B] int32."""

    def __init__(self, B, cap, device):
        self.B, self.cap, self.device = B, int(cap), torch.device(device)
        self.head = torch.zeros((B, NSTATES), dtype=torch.int64, device=device)
        self.stack = torch.zeros((B, NSTATES, self.cap), dtype=torch.int32, device=device)
        self.len64 = torch.zeros((B, NSTATES), dtype=torch.int32, device=device)
        self.status = torch.zeros(B, dtype=torch.int32, device=device)

    @property
    def len(self):
        """Stack words of a chain, all states together [B] (what the bit accounting counts)."""
        return self.len64.sum(1, dtype=torch.int32)

    def prefix(self, k):
        if k == self.B:
            return self
        cache = self.__dict__.setdefault("_prefixes", {})
        v = cache.get(int(k))
        if v is None:
            v = object.__new__(RansState64)
            v.B, v.cap, v.device = int(k), self.cap, self.device
            v.head, v.stack, v.len64, v.status = self.head[:k], self.stack[:k], self.len64[:k], self.status[:k]
            cache[int(k)] = v
        ml = getattr(self, "min_len", None)
        if ml is not None and getattr(v, "min_len", None) is None:
            v.min_len = ml[:k]
        return v

    @classmethod
    def from_lists(cls, states, cap=None, device="cuda"):
        """states: per chain either a reference-style list (dealt onto the 64 states by split_state) or a list of 64
        sub-state lists [w0, ..., head]."""
        import numpy as np
        nested = [s if isinstance(s[0], (list, tuple)) else split_state(s) for s in states]
        B = len(nested)
        n = max(len(sub) - 1 for ch in nested for sub in ch)
        cap = max(int(cap or 0), n)
        st = cls(B, cap, device)
        stack = np.zeros((B, NSTATES, cap), dtype=np.uint32)
        head = np.zeros((B, NSTATES), dtype=np.uint64)
        ln = np.zeros((B, NSTATES), dtype=np.int32)
        for b, ch in enumerate(nested):
            assert len(ch) == NSTATES
            for j, sub in enumerate(ch):
                k = len(sub) - 1
                stack[b, j, :k] = np.asarray(sub[:-1], dtype=np.uint64).astype(np.uint32)
                head[b, j] = sub[-1]
                ln[b, j] = k
        st.stack.copy_(torch.from_numpy(stack.view(np.int32)))
        st.head.copy_(torch.from_numpy(head.view(np.int64)))
        st.len64.copy_(torch.from_numpy(ln))
        return st

    def to_lists(self):
        """-> per chain a list of 64 sub-state lists [w0, ..., head]."""
        import numpy as np
        stack = self.stack.cpu().numpy().view(np.uint32)
        head = self.head.cpu().numpy().view(np.uint64)
        ln = self.len64.cpu().numpy()
        return [[[int(w) for w in stack[b, j, : ln[b, j]]] + [int(head[b, j])] for j in range(NSTATES)]
                for b in range(self.B)]

    check = RansState.check


def _layer64_args(state, endpoints, mu, scale, step):
    """mu/scale hold either one row set per chain [B, D] or a single one shared by all chains ([D] or [1, D]: the
    prior) -- p_stride D or 0."""
    B, D = state.B, endpoints.shape[0]
    K = endpoints.shape[1] + 1
    endpoints, es = _row_stride(endpoints, K - 1)
    step = _step(step, D)
    mu, scale = mu.contiguous(), scale.contiguous()
    if mu.shape != scale.shape or mu.numel() not in (D, B * D):
        raise BitswapHipError(f"mu/scale must both hold {D} (shared) or {B}x{D} values")
    ps = D if (mu.numel() == B * D and mu.dim() == 2) else 0
    return B, D, K, endpoints, es, step, mu, scale, ps


def layer_pop64(state, endpoints, mu, scale, bits=31, quantbits=10, centres=None, step=None, spec=None):
    """64-state format: logistic CDF -> integer table -> pop, one launch; mu/scale [B,D] or [D] / [1,D] (one row set
    shared by all chains: the prior).  -> (sym [B,D] int32, z [B,D] float32 | None)."""
    _need_cuda(endpoints, mu, scale, centres, step, state.head)
    B, D, K, endpoints, es, step, mu, scale, ps = _layer64_args(state, endpoints, mu, scale, step)
    sym = torch.empty((B, D), dtype=torch.int32, device=mu.device)
    z, cs = None, 0
    if centres is not None:
        centres, cs = _row_stride(centres, K)
        z = torch.empty((B, D), dtype=torch.float32, device=mu.device)
    _check(load().bs_layer_pop64(_ptr(state.head), _ptr(state.stack), _ptr(state.len64), state.cap, _ptr(endpoints), es,
                                 _ptr(step), _spec(step, spec), _ptr(mu), _ptr(scale), ps, _param_dtype(mu), B, D, K, bits, quantbits,
                                 _ptr(sym), _ptr(centres), cs, _ptr(z), _ptr(state.status), _stream()), "bs_layer_pop64")
    return sym, z


def layer_push64(state, endpoints, mu, scale, sym, bits=31, quantbits=10, step=None, spec=None):
    _need_cuda(endpoints, mu, scale, sym, step, state.head)
    B, D, K, endpoints, es, step, mu, scale, ps = _layer64_args(state, endpoints, mu, scale, step)
    sym = sym.contiguous()
    if sym.dtype != torch.int32:
        sym = sym.to(torch.int32)
    if tuple(sym.shape) != (B, D):
        raise BitswapHipError(f"sym must be [{B},{D}]")
    _check(load().bs_layer_push64(_ptr(state.head), _ptr(state.stack), _ptr(state.len64), state.cap, _ptr(endpoints), es,
                                  _ptr(step), _spec(step, spec), _ptr(mu), _ptr(scale), ps, _param_dtype(mu), _ptr(sym), B, D, K, bits,
                                  quantbits, _ptr(state.status), _stream()), "bs_layer_push64")


class MaskedStream:
    """A HIP stream whose kernels run on the compute units of mask bits [first_cu, first_cu + n_cus) only
    (bs_stream_create_cu_mask: 8 m consecutive bits = m CUs on each of the 8 XCDs), wrapped for `torch.cuda.stream()`.
    Scheduling only: results do not depend on where a kernel runs."""

    def __init__(self, first_cu, n_cus, device=None):
        h = C.c_void_p()
        _check(load().bs_stream_create_cu_mask(int(first_cu), int(n_cus), C.byref(h)), "bs_stream_create_cu_mask")
        self.handle, self.first_cu, self.n_cus = h.value, int(first_cu), int(n_cus)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.ExternalStream(self.handle, device=dev)

    def close(self):
        if self.handle:
            self.stream.synchronize()
            load().bs_stream_destroy(C.c_void_p(self.handle))
            self.handle = None


def where(n, spin_cycles=20000, stream=None):
    """Diagnostics: launch n one-wavefront workgroups on `stream` (torch stream; default: the current one) and return where
    each ran as (xcc, se, sh, cu) tuples (bs_debug_where)."""
    ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    if stream is None:
        _check(load().bs_debug_where(_ptr(ids), n, spin_cycles, _stream()), "bs_debug_where")
    else:
        with torch.cuda.stream(stream):
            _check(load().bs_debug_where(_ptr(ids), n, spin_cycles, _stream()), "bs_debug_where")
    torch.cuda.synchronize()
    v = ids.cpu().numpy().view(np.uint32)
    return [(int(x >> 16) & 15, int(x >> 13) & 7, int(x >> 12) & 1, int(x >> 8) & 15) for x in v]


def gather_centres(centres, sym):
    """z[b,d] = float32(centres[d, sym[b,d]])  (mnist_compress.py:181,196 + Model's .float())."""
    _need_cuda(centres, sym)
    B, D = sym.shape
    K = centres.shape[1]
    centres, cs = _row_stride(centres, K)
    sym = sym.contiguous()
    if sym.dtype != torch.int32:
        sym = sym.to(torch.int32)
    out = torch.empty((B, D), dtype=torch.float32, device=sym.device)
    _check(load().bs_gather_centres(_ptr(centres), cs, _ptr(sym), B, D, K, _ptr(out), _stream()),
           "bs_gather_centres")
    return out


# ---- conv-stack epilogues (net_epilogue.hip) ---------------------------------------------------------
def bias_residual_elu(x, bias=None, res=None, want_sum=False, want_act=True, inplace=True):
    """s = x + bias[c] (+ res) on an NCHW float32 tensor -> (s | None, ELU(s) | None).
    With inplace=True the (single) requested output overwrites x (a conv result nobody else reads)."""
    _need_cuda(x, bias, res)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    assert res is None or (res.is_contiguous() and res.shape == x.shape and res.dtype == torch.float32)
    N, Cc, H, W = x.shape
    s_out = (x if inplace else torch.empty_like(x)) if want_sum else None
    a_out = (x if (inplace and not want_sum) else torch.empty_like(x)) if want_act else None
    _check(load().bs_bias_residual_elu_f32(_ptr(x), _ptr(bias), _ptr(res), _ptr(s_out), _ptr(a_out), N, Cc, H * W,
                                           _stream()), "bs_bias_residual_elu_f32")
    return s_out, a_out


def head_params(x, bias, mode):
    """x [N,2C,H,W] (mu and std filters stacked) -> mu, scale [N, C*H*W] float32."""
    _need_cuda(x, bias)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[1] % 2 == 0
    N, C2, H, W = x.shape
    Cc = C2 // 2
    mu = torch.empty((N, Cc * H * W), dtype=torch.float32, device=x.device)
    scale = torch.empty_like(mu)
    _check(load().bs_head_params_f32(_ptr(x), _ptr(bias), _ptr(mu), _ptr(scale), N, Cc, H * W, mode, _stream()),
           "bs_head_params_f32")
    return mu, scale


def expand_rows5(x, bias=None, act=True):
    """[N,C,H,W] -> [N, C*5, H+4, W]: ELU(x + bias) zero-padded and shifted along x for the five kernel
    columns (operand of the 5x5 conv-as-GEMM, see include/bitswap_hip.h)."""
    _need_cuda(x, bias)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc * 5, H + 4, W), dtype=torch.float32, device=x.device)
    _check(load().bs_expand_rows5_f32(_ptr(x), _ptr(bias), _ptr(out), N, Cc, H, W, 1 if act else 0, _stream()),
           "bs_expand_rows5_f32")
    return out


def wino_in(x, bias=None, act=True, cfg=(6, 4)):
    """[N,C,H,W] -> V [ts*ts, C, N*T]: Winograd input transform of ELU(x + bias); cfg = (tile size, tile stride)
    from winograd.tile_config() (see include/bitswap_hip.h)."""
    _need_cuda(x, bias)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    ts, ms = cfg
    N, Cc, H, W = x.shape
    V = torch.empty((ts * ts, Cc, N * (H // ms) * (W // ms)), dtype=torch.float32, device=x.device)
    _check(load().bs_wino_in_f32(_ptr(x), _ptr(bias), _ptr(V), N, Cc, H, W, ts, ms, 1 if act else 0, _stream()),
           "bs_wino_in_f32")
    return V


def wino_out(M, shape, bias=None, res=None, want_sum=False, want_act=True, cfg=(6, 4)):
    """M [ts*ts, C, N*T] -> (s | None, ELU(s) | None) as [N,C,H,W], s = A^T M A + bias[c] (+ res)."""
    _need_cuda(M, bias, res)
    ts, ms = cfg
    N, Cc, H, W = shape
    assert M.dtype == torch.float32 and M.is_contiguous() and tuple(M.shape) == (ts * ts, Cc, N * (H // ms) * (W // ms))
    assert res is None or (res.is_contiguous() and tuple(res.shape) == tuple(shape) and res.dtype == torch.float32)
    s_out = torch.empty(shape, dtype=torch.float32, device=M.device) if want_sum else None
    a_out = torch.empty(shape, dtype=torch.float32, device=M.device) if want_act else None
    _check(load().bs_wino_out_f32(_ptr(M), _ptr(bias), _ptr(res), _ptr(s_out), _ptr(a_out), N, Cc, H, W, ts, ms,
                                  _stream()), "bs_wino_out_f32")
    return s_out, a_out


def conv3_wino_supported(cin, h, w):
    """Shapes bs_conv3_wino_f32 takes: a wavefront covers whole images (T = (h/4)*(w/4) divides 64) and the Cin planes
    of those images plus four activation tiles fit the 160 KB of LDS."""
    if h % 4 or w % 4 or h < 4 or w < 4:
        return False
    t = (h // 4) * (w // 4)
    if t > 64 or 64 % t:
        return False
    img, lp = 64 // t, (h + 4) * (w + 4)
    return (img * cin * lp + 4 + 4 * (img * lp + 4)) * 4 <= 160 * 1024


def conv3_wino(x, w, bias, act=3, want_act=True, ts_out=6):
    """Input conv of a stack fused with the first transform: x [N,Cin,H,W], w [C,Cin,3,3] -> (h [N,C,H,W] | None,
    V [ts_out^2, C, N*T]) with h = ELU(conv3x3(x) + bias) (act & 1) and V = B^T ELU(h) B (act & 2); see
    include/bitswap_hip.h, bs_conv3_wino_f32."""
    _need_cuda(x, w, bias)
    assert x.dtype == w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous() and x.dim() == 4
    N, Cin, H, W = x.shape
    Cc = w.shape[0]
    assert tuple(w.shape) == (Cc, Cin, 3, 3)
    T = (H // 4) * (W // 4)
    h = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device) if want_act else None
    V = torch.empty((ts_out * ts_out, Cc, N * T), dtype=torch.float32, device=x.device)
    _check(load().bs_conv3_wino_f32(_ptr(x), _ptr(w), _ptr(bias), int(act), _ptr(h), _ptr(V), ts_out, N, Cin, Cc, H, W,
                                    _stream()), "bs_conv3_wino_f32")
    return h, V


def wino_gemm_supported(U, V):
    """bs_wino_gemm_f32 takes float32, contiguous operands with Cin % 16 == 0 and cols % 4 == 0."""
    return (U.dtype == V.dtype == torch.float32 and U.dim() == V.dim() == 3 and U.is_contiguous() and V.is_contiguous()
            and U.shape[0] == V.shape[0] and U.shape[2] == V.shape[1] and U.shape[2] % 16 == 0 and V.shape[2] % 4 == 0)


def wino_gemm(U, V, out=None):
    """M [T, Cout, cols] = U [T, Cout, Cin] x V [T, Cin, cols] on the matrix cores, float32 with float32 accumulation
    in one fixed order per output element: bitwise independent of cols (include/bitswap_hip.h, bs_wino_gemm_f32)."""
    _need_cuda(U, V, out)
    assert wino_gemm_supported(U, V)
    T, Cout, Cin = U.shape
    cols = V.shape[2]
    M = out if out is not None else torch.empty((T, Cout, cols), dtype=torch.float32, device=V.device)
    assert M.is_contiguous() and tuple(M.shape) == (T, Cout, cols) and M.dtype == torch.float32
    _check(load().bs_wino_gemm_f32(_ptr(U), _ptr(V), _ptr(M), T, Cout, Cin, cols, _stream()), "bs_wino_gemm_f32")
    return M


def split_bf16x3(U):
    """U float32 -> limbs [3, *U.shape] bfloat16 with U == limbs[0] + limbs[1] + limbs[2] exactly (round-to-nearest split:
    3 x 8 significand bits cover the 24 of float32)."""
    assert U.dtype == torch.float32
    x0 = U.to(torch.bfloat16)
    r1 = U - x0.float()
    x1 = r1.to(torch.bfloat16)
    r2 = r1 - x1.float()
    x2 = r2.to(torch.bfloat16)
    return torch.stack([x0, x1, x2], 0).contiguous()


class FragsBf16x3:
    """The weight operand of bs_wino_gemm_bf16x3: the pre-tiled limb fragments together with the logical shape they were cut
    from (round 4 hung the shape on the tensor as an ad-hoc attribute, which any .to() / .clone() silently dropped)."""

    __slots__ = ("frags", "T", "Cout", "Cin")

    def __init__(self, frags, T, Cout, Cin):
        want = (T, (Cout + 31) // 32, Cin // 16, 3, 64, 8)
        if frags.dtype != torch.bfloat16 or not frags.is_contiguous() or tuple(frags.shape) != want:
            raise BitswapHipError(f"bf16x3 fragments must be a contiguous bfloat16 tensor of shape {want}, got {tuple(frags.shape)}")
        self.frags, self.T, self.Cout, self.Cin = frags, int(T), int(Cout), int(Cin)

    def to(self, *a, **k):
        return FragsBf16x3(self.frags.to(*a, **k).contiguous(), self.T, self.Cout, self.Cin)


def frags_bf16x3(U):
    """The weight operand bs_wino_gemm_bf16x3 takes: U [T, Cout, Cin] float32 split into its three bfloat16 limbs and
    pre-tiled as MFMA A fragments, [T, ceil(Cout/32), Cin/16, 3, 64, 8] -- lane l = 32 (k // 8) + row of a 32-row tile holds 8
    consecutive k of a 16-deep block (include/bitswap_hip.h).  Rows beyond Cout are zero.  Done once per model (Model.fuse()).
    -> FragsBf16x3 (fragments + the shape they stand for)."""
    assert U.dtype == torch.float32 and U.dim() == 3 and U.shape[2] % 16 == 0
    T, Cout, Cin = U.shape
    pad = (-Cout) % 32
    if pad:
        U = torch.cat([U, U.new_zeros((T, pad, Cin))], 1)
    L = split_bf16x3(U)                                              # [3, T, R, Cin]
    R = Cout + pad
    L = L.view(3, T, R // 32, 32, Cin // 16, 2, 8)                  # limb, t, row tile, row, k block, k half, k
    L = L.permute(1, 2, 4, 0, 5, 3, 6).contiguous()                 # t, row tile, k block, limb, k half, row, k
    return FragsBf16x3(L.view(T, R // 32, Cin // 16, 3, 64, 8), T, Cout, Cin)


def wino_gemm_bf16x3(Uf, V, nprod=6, out=None):
    """M [T, Cout, cols] = U x V with U given as frags_bf16x3(U) and V [T, Cin, cols] float32 split in the kernel: nprod limb
    products per k block on the bf16 matrix cores, float32 accumulation, one fixed order per output
    (include/bitswap_hip.h, bs_wino_gemm_bf16x3).  Opt-in arithmetic."""
    if not isinstance(Uf, FragsBf16x3):
        raise BitswapHipError("wino_gemm_bf16x3 takes the FragsBf16x3 that frags_bf16x3() returns")
    _need_cuda(Uf.frags, V, out)
    T, Cout, Cin = Uf.T, Uf.Cout, Uf.Cin
    assert V.dtype == torch.float32 and V.dim() == 3 and V.is_contiguous()
    if V.shape[0] != T or V.shape[1] != Cin or V.shape[2] % 4:
        raise BitswapHipError(f"V {tuple(V.shape)} does not go with fragments of a [{T}, {Cout}, {Cin}] operand (columns: multiple of 4)")
    cols = V.shape[2]
    M = out if out is not None else torch.empty((T, Cout, cols), dtype=torch.float32, device=V.device)
    assert M.is_contiguous() and tuple(M.shape) == (T, Cout, cols) and M.dtype == torch.float32
    _check(load().bs_wino_gemm_bf16x3(_ptr(Uf.frags), _ptr(V), _ptr(M), T, Cout, Cin, cols, int(nprod), _stream()), "bs_wino_gemm_bf16x3")
    if SELFCHECK_BF16X3 is not None:      # diagnostics (tools/bf16x3_repro.py): the same launch again, compared on the device
        M2 = torch.empty_like(M)
        _check(load().bs_wino_gemm_bf16x3(_ptr(Uf.frags), _ptr(V), _ptr(M2), T, Cout, Cin, cols, int(nprod), _stream()), "bs_wino_gemm_bf16x3")
        d = M != M2
        acc = SELFCHECK_BF16X3.setdefault((T, Cout, Cin, cols), {
            "launches": 0, "differing": torch.zeros((), dtype=torch.int64, device=M.device),
            "cols": torch.zeros(cols, dtype=torch.bool, device=M.device), "rows": torch.zeros(Cout, dtype=torch.bool, device=M.device),
            "t": torch.zeros(T, dtype=torch.bool, device=M.device), "elements": torch.zeros((), dtype=torch.int64, device=M.device)})
        acc["launches"] += 1
        acc["differing"] += d.any()
        acc["elements"] += d.sum()
        acc["cols"] |= d.any(0).any(0)
        acc["rows"] |= d.any(0).any(1)
        acc["t"] |= d.any(1).any(1)
    return M


SELFCHECK_BF16X3 = None     # set to {} to make every bs_wino_gemm_bf16x3 launch run twice and record where the two results differ


def small_k_gemm(U, V):
    """M [T, Cout, cols] = U [T, Cout, Cin] x V [T, Cin, cols], float32, small Cin (the stacks' input convs in the
    Winograd domain): a dedicated kernel that costs the write of M."""
    _need_cuda(U, V)
    assert U.dtype == V.dtype == torch.float32 and U.is_contiguous() and V.is_contiguous()
    T, Cout, Cin = U.shape
    assert V.shape[0] == T and V.shape[1] == Cin
    cols = V.shape[2]
    M = torch.empty((T, Cout, cols), dtype=torch.float32, device=V.device)
    _check(load().bs_small_k_gemm_f32(_ptr(U), _ptr(V), _ptr(M), T, Cout, Cin, cols, _stream()), "bs_small_k_gemm_f32")
    return M


def wino_fused(src, shape, ts_in=0, bias=None, res=None, act=True, want_sum=False, want_act=False, ts_out=0):
    """One pass between two Winograd-domain GEMMs (include/bitswap_hip.h, bs_wino_fused_f32).
    src: x [N,C,H,W] (ts_in = 0) or M [ts_in^2, C, N*T]; shape = (N,C,H,W).  act: False/0, True/1 (ELU), or 3 (ELU for
    act_out, ELU of that again for V).
    -> (sum | None, act | None, V | None)."""
    _need_cuda(src, bias, res)
    N, Cc, H, W = shape
    T = (H // 4) * (W // 4)
    assert src.dtype == torch.float32 and src.is_contiguous()
    assert tuple(src.shape) == (tuple(shape) if ts_in == 0 else (ts_in * ts_in, Cc, N * T))
    assert res is None or (res.is_contiguous() and tuple(res.shape) == tuple(shape) and res.dtype == torch.float32)
    s_out = torch.empty(shape, dtype=torch.float32, device=src.device) if want_sum else None
    a_out = torch.empty(shape, dtype=torch.float32, device=src.device) if want_act else None
    V = torch.empty((ts_out * ts_out, Cc, N * T), dtype=torch.float32, device=src.device) if ts_out else None
    _check(load().bs_wino_fused_f32(_ptr(src), ts_in, _ptr(bias), _ptr(res), int(act), _ptr(s_out), _ptr(a_out),
                                    _ptr(V), ts_out, N, Cc, H, W, _stream()), "bs_wino_fused_f32")
    return s_out, a_out, V


# ---- diagnostics: run a kernel wrapper twice and record on the device whether (and for which image) the two results differ ---
SELFCHECK = None     # set to {} (tools/bf16x3_repro.py): wino_fused / conv3_wino / wino_gemm / head_params launches are doubled


def _selfchecked(name, fn, image_of):
    """image_of(tensor, args) -> per-element image index or None; wraps fn so that with SELFCHECK set every call runs twice."""
    def wrapped(*a, **k):
        out = fn(*a, **k)
        if SELFCHECK is None:
            return out
        out2 = fn(*a, **k)
        o1 = out if isinstance(out, (tuple, list)) else (out,)
        o2 = out2 if isinstance(out2, (tuple, list)) else (out2,)
        acc = SELFCHECK.setdefault(name, {"calls": 0, "differing": torch.zeros((), dtype=torch.int64, device="cuda"),
                                          "images": torch.zeros(4096, dtype=torch.bool, device="cuda")})
        acc["calls"] += 1
        for x, y in zip(o1, o2):
            if x is None or not torch.is_tensor(x):
                continue
            d = x != y
            acc["differing"] += d.any()
            idx = image_of(x, a, k)
            if idx is not None:
                n = int(idx[1])
                acc["images"][:n] |= d.reshape(idx[0]).any(dim=tuple(i for i in range(len(idx[0])) if i != idx[2]))
        return out
    return wrapped


def _img_fused(x, a, k):
    N, C, H, W = a[1]
    T = (H // 4) * (W // 4)
    if x.dim() == 4:                      # [N, C, H, W]
        return ((N, C * H * W), N, 0)
    return ((x.shape[0] * x.shape[1], N, T), N, 1)     # V / M [ts^2, C, N*T]


def _img_conv3(x, a, k):
    N = a[0].shape[0]
    if x.dim() == 4:
        return ((N, x.numel() // N), N, 0)
    T = x.shape[2] // N
    return ((x.shape[0] * x.shape[1], N, T), N, 1)


def _img_gemm(x, a, k):
    cols = x.shape[2]
    if cols % 16:
        return None
    return ((x.shape[0] * x.shape[1], cols // 16, 16), cols // 16, 1)


def _img_head(x, a, k):
    return ((x.shape[0], x.numel() // x.shape[0]), x.shape[0], 0)


wino_fused = _selfchecked("wino_fused", wino_fused, _img_fused)
conv3_wino = _selfchecked("conv3_wino", conv3_wino, _img_conv3)
wino_gemm = _selfchecked("wino_gemm", wino_gemm, _img_gemm)
head_params = _selfchecked("head_params", head_params, _img_head)
