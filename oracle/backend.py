"""Oracle-backed stand-in for bitswap_amd.codec.HipBackend -- TEST INFRASTRUCTURE ONLY.

Lets tests/ and bench.py's cpu_baseline leg run the package's schedule code (codec.py) on the
CPU with the C restatement of the reference doing the arithmetic.  Never constructed by the
product: bitswap_amd imports nothing from oracle/.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import oracle as O


class _Tables:
    """Lazy 'cdf rows' handle: the oracle rebuilds each row inside layer_pop/layer_push."""

    def __init__(self, endpoints, mu, scale, quantbits, shared=False, step=None):
        self.e, self.mu, self.scale, self.q, self.shared, self.step = endpoints, mu, scale, quantbits, shared, step

    def __getitem__(self, i):
        return _Tables(self.e, self.mu[i:i + 1], self.scale[i:i + 1], self.q, shared=True, step=self.step)


class OracleState:
    def __init__(self, states, cap):
        self.stacks = [O.Stack(s, cap=cap) for s in states]
        self.B = len(states)
        self.rc = np.zeros(self.B, dtype=np.int32)

    def prefix(self, k):
        """First k chains, sharing the stacks and the status array (mirror of RansState.prefix)."""
        if k == self.B:
            return self
        v = object.__new__(OracleState)
        v.stacks, v.B, v.rc = self.stacks[:k], int(k), self.rc[:k]
        if getattr(self, "min_len", None) is not None:
            v.min_len = self.min_len[:k]
        return v

    @property
    def len(self):
        return torch.tensor([int(s.len[0]) for s in self.stacks], dtype=torch.int32)

    @property
    def status(self):
        return torch.from_numpy(self.rc)

    def to_lists(self):
        return [s.tolist() for s in self.stacks]


class OracleBackend:
    name = "oracle"

    def __init__(self, mode=O.MODE_DET, threads=1):
        self.mode, self.threads = mode, threads
        self.device = torch.device("cpu")
        self.pool = ThreadPoolExecutor(threads) if threads > 1 else None

    def _map(self, fn, n):
        if self.pool is None:
            return [fn(b) for b in range(n)]
        return list(self.pool.map(fn, range(n)))

    @staticmethod
    def _np(t):
        return np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t)

    def new_state(self, states, cap):
        return OracleState(states, cap)

    def table_buffer(self, B, D, K):
        return None

    @staticmethod
    def uniform_step(endpoints, tol_ulps=8.0):
        """Independent restatement of the product's decision rule (bitswap_amd.bins.uniform_step, tests compare the
        two): bin width h = (e[K-2] - e[0]) / (K - 2) per row if every endpoint lies within tol_ulps units in the last
        place of e[0] + j*h -- uniform-width bins, discretization.py:105-118 -- else None."""
        e = np.asarray(endpoints.detach().cpu().numpy() if torch.is_tensor(endpoints) else endpoints, dtype=np.float64)
        if e.ndim != 2 or e.shape[1] < 3:
            return None
        n = e.shape[1] - 1
        with np.errstate(all="ignore"):
            h = (e[:, -1] - e[:, 0]) / np.float64(n)
            if not np.all(np.isfinite(h)) or not np.all(h > 0):
                return None
            dev = np.abs(e - (e[:, :1] + np.arange(n + 1, dtype=np.float64)[None] * h[:, None]))
            lim = tol_ulps * np.spacing(np.maximum(np.abs(e[:, 0]), np.abs(e[:, -1])))
        return np.ascontiguousarray(h) if np.all(dev <= lim[:, None]) else None

    def bin_step(self, endpoints):
        """Same decision as HipBackend.bin_step; only the deterministic mode has the uniform-bin specs 2 / 3 -- the libm /
        torch modes restate the reference formula and ignore them."""
        if self.mode != O.MODE_DET or endpoints.shape[1] + 1 < 256:
            return None
        return self.uniform_step(endpoints)

    # CDF specification of tables of uniform-width bins (2 or 3); the codec sets it from its cdf_spec, like HipBackend's
    cdf_spec = 4      # (a codec sets it from its own cdf_spec; 4 = the product's default, bitswap_amd/meta.py)

    def _mode(self, step):
        if step is not None and self.mode == O.MODE_DET:
            return {2: O.MODE_DET2, 3: O.MODE_DET3, 4: O.MODE_DET4}[self.cdf_spec]
        return self.mode

    @staticmethod
    def _torch_cdf_rows(e, mu, scale, bits, q):
        """MODE_TORCH: the reference's lines verbatim -- logistic_cdf (utils/torch/rand.py:67-68), pmf assembly
        (mnist_compress.py:183-185) -- evaluated by this torch build, then the oracle's integer tables.  Reproduces
        the reference's words exactly wherever the reference ran on the same torch CPU build."""
        e, mu, scale = torch.from_numpy(e), torch.from_numpy(mu), torch.from_numpy(scale)
        cdfs = torch.sigmoid((e.t() - mu) / scale).t()
        pmfs = cdfs[:, 1:] - cdfs[:, :-1]
        pmfs = torch.cat((cdfs[:, 0].unsqueeze(1), pmfs, 1. - cdfs[:, -1].unsqueeze(1)), dim=1)
        _, cdf, rc = O.tables(pmfs.numpy(), bits, q)
        return cdf, rc

    def _pop1(self, stack, t, i, bits):
        if self.mode == O.MODE_TORCH:
            cdf, rc = self._torch_cdf_rows(t.e, t.mu[i], t.scale[i], bits, t.q)
            if rc:
                return np.zeros(t.e.shape[0], dtype=np.int32), rc
            return O.pop(stack, cdf, bits)
        return O.layer_pop(stack, t.e, t.mu[i], t.scale[i], bits, t.q, self._mode(t.step), t.step)

    def _push1(self, stack, e, mu, scale, sym, bits, q, step):
        if self.mode == O.MODE_TORCH:
            cdf, rc = self._torch_cdf_rows(e, mu, scale, bits, q)
            return rc or O.push(stack, cdf, sym, bits)
        return O.layer_push(stack, e, mu, scale, sym, bits, q, self._mode(step), step)

    def shared_table(self, endpoints, mu, scale, quantbits, bits, step=None):
        return self.tables(endpoints, mu, scale, quantbits, bits, step=step)[0]

    def tables(self, endpoints, mu, scale, quantbits, bits, out=None, step=None, status=None):
        return _Tables(self._np(endpoints).astype(np.float64), self._np(mu).astype(np.float64),
                       self._np(scale).astype(np.float64), quantbits,
                       step=None if step is None else self._np(step).astype(np.float64))

    def pop(self, state, t, K, bits, centres=None):
        def one(b):
            if state.rc[b]:
                return np.zeros(t.e.shape[0], dtype=np.int32)
            i = 0 if t.shared else b
            sym, rc = self._pop1(state.stacks[b], t, i, bits)
            state.rc[b] = rc
            if rc:
                sym[:] = 0   # sticky failure (e.g. too few initial bits); reported by check()
            return sym
        sym = torch.from_numpy(np.stack(self._map(one, state.B)))
        z = self.centres(centres, sym) if centres is not None else None
        return sym, z

    def push_params(self, state, endpoints, mu, scale, sym, quantbits, bits, step=None):
        e, mu, scale = self._np(endpoints).astype(np.float64), self._np(mu).astype(np.float64), self._np(scale).astype(np.float64)
        sym = self._np(sym).astype(np.int32)
        step = None if step is None else self._np(step).astype(np.float64)

        def one(b):
            if not state.rc[b]:
                state.rc[b] = self._push1(state.stacks[b], e, mu[b], scale[b], sym[b], bits, quantbits, step)
        self._map(one, state.B)

    def push_table(self, state, t, sym, K, bits):
        sym = self._np(sym).astype(np.int32)

        def one(b):
            if not state.rc[b]:
                i = 0 if t.shared else b
                state.rc[b] = self._push1(state.stacks[b], t.e, t.mu[i], t.scale[i], sym[b], bits, t.q, t.step)
        self._map(one, state.B)

    def centres(self, centres, sym):
        idx = torch.as_tensor(sym).long()
        rows = torch.arange(idx.shape[1]).unsqueeze(0).expand_as(idx)
        return centres[rows, idx].float()

    def check(self, state, what):
        if state.rc.any():
            b = int(np.nonzero(state.rc)[0][0])
            raise RuntimeError(f"{what}: chain {b}: oracle status {int(state.rc[b])}")


# ---- the opt-in 64-state stream format (include/bitswap_hip.h, BS_FORMAT_WAVE64) restated on the oracle's primitives ----
NSTATES = 64


def split_state(s, nstates=NSTATES):
    """Restates bitswap_amd.hip.split_state: initial words dealt round-robin onto the states, last one as the head."""
    words = list(s[:-1]) + [s[-1] >> 32]
    return [words[j::nstates][:-1] + [words[j::nstates][-1] << 32] for j in range(nstates)]


class OracleState64:
    def __init__(self, states, cap):
        nested = [s if isinstance(s[0], (list, tuple)) else split_state(s) for s in states]
        self.stacks = [[O.Stack(sub, cap=cap) for sub in ch] for ch in nested]
        self.B = len(states)
        self.rc = np.zeros(self.B, dtype=np.int32)

    def prefix(self, k):
        if k == self.B:
            return self
        v = object.__new__(OracleState64)
        v.stacks, v.B, v.rc = self.stacks[:k], int(k), self.rc[:k]
        if getattr(self, "min_len", None) is not None:
            v.min_len = self.min_len[:k]
        return v

    @property
    def len64(self):
        return torch.tensor([[int(s.len[0]) for s in ch] for ch in self.stacks], dtype=torch.int32)

    @property
    def len(self):
        return self.len64.sum(1, dtype=torch.int32)

    @property
    def status(self):
        return torch.from_numpy(self.rc)

    def to_lists(self):
        return [[s.tolist() for s in ch] for ch in self.stacks]


class Oracle64Backend(OracleBackend):
    """Symbol d of every coding operation is coded on state d % 64 of its chain, with the reference's arithmetic and
    order inside that state: state j sees the sub-sequence d = j, j + 64, ... -- exactly what the oracle's single-state
    layer_pop / layer_push (ANS.decode / ANS.encode, mnist_compress.py:49-68) do when handed every 64th row."""

    name = "oracle-wave64"

    def new_state(self, states, cap):
        return OracleState64(states, (int(cap) + NSTATES - 1) // NSTATES + 64)

    def pop(self, state, t, K, bits, centres=None):
        D = t.e.shape[0]

        def one(b):
            sym = np.zeros(D, dtype=np.int32)
            if state.rc[b]:
                return sym
            i = 0 if t.shared else b
            for j in range(min(NSTATES, D)):
                sl = slice(j, None, NSTATES)
                step = None if t.step is None else np.ascontiguousarray(t.step[sl])
                s, rc = O.layer_pop(state.stacks[b][j], np.ascontiguousarray(t.e[sl]), np.ascontiguousarray(t.mu[i][sl]),
                                    np.ascontiguousarray(t.scale[i][sl]), bits, t.q, self._mode(t.step), step)
                if rc:
                    state.rc[b] = rc
                    return np.zeros(D, dtype=np.int32)
                sym[sl] = s
            return sym
        sym = torch.from_numpy(np.stack(self._map(one, state.B)))
        z = self.centres(centres, sym) if centres is not None else None
        return sym, z

    def _push(self, state, b, e, mu, scale, sym, bits, q, step):
        for j in range(min(NSTATES, e.shape[0])):
            sl = slice(j, None, NSTATES)
            stp = None if step is None else np.ascontiguousarray(step[sl])
            rc = O.layer_push(state.stacks[b][j], np.ascontiguousarray(e[sl]), np.ascontiguousarray(mu[sl]),
                              np.ascontiguousarray(scale[sl]), np.ascontiguousarray(sym[sl]), bits, q, self._mode(step), stp)
            if rc:
                state.rc[b] = rc
                return

    def push_params(self, state, endpoints, mu, scale, sym, quantbits, bits, step=None):
        e, mu, scale = self._np(endpoints).astype(np.float64), self._np(mu).astype(np.float64), self._np(scale).astype(np.float64)
        sym = self._np(sym).astype(np.int32)
        step = None if step is None else self._np(step).astype(np.float64)

        def one(b):
            if not state.rc[b]:
                self._push(state, b, e, mu[b], scale[b], sym[b], bits, quantbits, step)
        self._map(one, state.B)

    def push_table(self, state, t, sym, K, bits):
        sym = self._np(sym).astype(np.int32)

        def one(b):
            if not state.rc[b]:
                i = 0 if t.shared else b
                self._push(state, b, t.e, t.mu[i], t.scale[i], sym[b], bits, t.q, t.step)
        self._map(one, state.B)
