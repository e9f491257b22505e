"""Batched Bit-Swap / BB-ANS coding schedules over B independent chains held in HBM.

This replaces the per-image Python loops of the reference's `compress()` functions
(sender mnist_compress.py:164-263, receiver :277-358; Bit-Swap :176-205 / :293-319, BB-ANS
:206-243 / :321-354, prior :245-251 / :284-291).  One "block step" codes one 32x32 block of every
chain in lock-step: the conv stacks see a [B, ...] batch, the table kernels see B*D rows, the rANS
kernels one chain per wavefront.  Nothing in a step synchronises with the host; word counts
for the bit accounting are snapshotted on the device.

The arithmetic back-end is an object with the interface of `HipBackend` below.  The product
constructs only `HipBackend` (HIP kernels through the C ABI, no fallback).  tests/ and bench.py's
cpu_baseline leg may pass the oracle-backed stand-in from oracle/backend.py to run the very same
schedule code on the CPU.
"""
import collections
import contextlib
import os

import numpy as np
import torch

from . import hip
from .rand import Bins, ImageBins


class HipBackend:
    """HIP kernels behind include/bitswap_hip.h.  All tensors live on one HIP device."""

    name = "hip"

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise hip.BitswapHipError("HipBackend needs a HIP device: bitswap_amd has no CPU coding path")
        hip.load()

    def new_state(self, states, cap):
        return hip.RansState.from_lists(states, cap=cap, device=self.device)

    # cdf rows are an internal hand-off between two of our kernels (never part of a stream: both forms pop the same
    # symbols).  Whole integer rows in the wave-native layout whenever the pop kernel has it (K = 256..2048; the reference's
    # linear rows otherwise) -- until a launch's rows add up to gigabytes: from `pivot_min_bytes` on, rows of uniform-width
    # bins (CDF spec 2) travel as 64 cumulative values and the pop kernel rebuilds the one group of bins it needs
    # (BS_LAYOUT_PIVOT, 512 B per row instead of 4 (K + 64)).  At 400 chains x 2048 dims the row-reading pop kernel sits at
    # the HBM roof (3.57 GB in 0.57 ms) and nothing overlaps with it; the rebuilding one takes 1.0 ms of instruction issue on
    # 400 SIMDs and leaves HBM to the other chain group's kernels.  With few chains the rows are no HBM problem and the
    # shorter pop wins (100 chains: 27.9 vs 36.7 ms per step, profiles/r03p).  BITSWAP_PIVOT=0: whole rows always.
    pivot = os.environ.get("BITSWAP_PIVOT", "1") == "1"
    pivot_min_bytes = int(float(os.environ.get("BITSWAP_PIVOT_MIN_BYTES", "2e9")))
    # CDF specification of tables of uniform-width bins (2 or 3, include/bitswap_hip.h); the codec sets it from its cdf_spec
    cdf_spec = hip.UNIFORM_CDF_SPEC

    def _sp(self, step):
        return None if step is None else self.cdf_spec

    def table_layout(self, K, uniform=False, D=64, B=1):
        if (uniform and self.pivot and hip.pivot_supported(K, D)
                and int(B) * int(D) * 4 * (K + 64) >= self.pivot_min_bytes):
            return hip.LAYOUT_PIVOT
        return hip.LAYOUT_WAVE if hip.wave_supported(K) else hip.LAYOUT_LINEAR

    def table_buffer(self, B, D, K, uniform=False):
        layout = self.table_layout(K, uniform, D, B)
        ld = hip.PIVOT_LD if layout == hip.LAYOUT_PIVOT else hip.wave_ld(K) if layout == hip.LAYOUT_WAVE else hip.aligned_ld(K)
        t = torch.empty((B, D, ld), dtype=torch.int32, device=self.device)
        t.bs_layout = layout
        return t

    def bin_step(self, endpoints):
        """Per-row bin width on the device if the rows are uniform-width bins the CDF spec 2 kernels take
        (K >= 256), else None (CDF spec 1)."""
        from .bins import uniform_step
        K = endpoints.shape[1] + 1
        h = uniform_step(endpoints) if K >= 256 else None
        return None if h is None else torch.from_numpy(h).to(self.device)

    def tables(self, endpoints, mu, scale, quantbits, bits, out=None, step=None, status=None):
        K = endpoints.shape[1] + 1
        layout = self.table_layout(K, step is not None, mu.shape[1], mu.shape[0])
        if out is not None:                              # a reusable buffer decides (a prefix of a larger batch keeps its layout)
            want = hip.PIVOT_LD if layout == hip.LAYOUT_PIVOT else hip.wave_ld(K) if layout == hip.LAYOUT_WAVE else hip.aligned_ld(K)
            if out.shape[-1] != want:
                if step is not None and out.shape[-1] == hip.PIVOT_LD:
                    layout = hip.LAYOUT_PIVOT
                elif out.shape[-1] == hip.wave_ld(K) and hip.wave_supported(K):
                    layout = hip.LAYOUT_WAVE
                else:
                    out = None
        return hip.logistic_tables(endpoints, mu, scale, bits, quantbits, out=out, layout=layout, step=step, status=status,
                                   spec=self._sp(step))

    def shared_table(self, endpoints, mu, scale, quantbits, bits, step=None):
        """One table row set [D, ld] shared by every chain (the prior)."""
        t = self.tables(endpoints, mu, scale, quantbits, bits, step=step)
        p = t[0]
        p.bs_layout = t.bs_layout
        return p

    def pop(self, state, cdf, K, bits, centres=None):
        return hip.rans_pop(state, cdf, K, bits, centres=centres)

    def push_params(self, state, endpoints, mu, scale, sym, quantbits, bits, step=None):
        f, c = hip.logistic_fc(endpoints, mu, scale, sym, state.status, bits, quantbits, step=step, spec=self._sp(step))
        hip.rans_push(state, f, c, bits)

    # push in two halves, for schedules that evaluate the (f, c) of a layer early and code it later (forked block step)
    def push_prepare(self, state, endpoints, mu, scale, sym, quantbits, bits, step=None):
        return hip.logistic_fc(endpoints, mu, scale, sym, state.status, bits, quantbits, step=step, spec=self._sp(step))

    def push_commit(self, state, token, bits):
        hip.rans_push(state, token[0], token[1], bits)

    def push_table(self, state, cdf, sym, K, bits):
        hip.rans_push_table(state, cdf, sym, K, bits)

    def centres(self, centres, sym):
        return hip.gather_centres(centres, sym)

    def check(self, state, what):
        state.check(what)


@contextlib.contextmanager
def deterministic_convs():
    """The decoder must reproduce the encoder's (mu, scale) bit for bit: MIOpen has to pick the same, deterministic
    algorithm on both sides (the reference sets the flags process-wide, mnist_compress.py:98-99).  Here they hold for
    the conv stacks of the codec only and the caller's settings come back afterwards."""
    be = torch.backends.cudnn
    prev = (be.deterministic, be.benchmark)
    be.deterministic, be.benchmark = True, False
    try:
        yield
    finally:
        be.deterministic, be.benchmark = prev


class _LazyTable:
    """What Hip64Backend.tables() hands to its pop(): the arguments of the fused kernel, not a table."""

    def __init__(self, endpoints, mu, scale, quantbits, step):
        self.endpoints, self.mu, self.scale, self.quantbits, self.step = endpoints, mu, scale, quantbits, step


class Hip64Backend(HipBackend):
    """The opt-in 64-state stream format (BS_FORMAT_WAVE64, include/bitswap_hip.h): every chain owns 64 rANS states and
    each coding operation is ONE launch that builds the integer table rows in registers and codes on them
    (bs_layer_pop64 / bs_layer_push64) -- no cdf rows in HBM, no serial kernels to split off: GroupedCodec gives every
    chain group ONE stream and lets the groups overlap each other (HBM-bound transforms and float64 coding kernels of
    one group under the matrix-core GEMMs of the other: +10 % at 800 chains, profiles/r02y).  Streams are not the
    reference's (64 heads per chain instead of one)."""

    name = "hip-wave64"

    def new_state(self, states, cap):
        """cap: word capacity the caller wants for a whole chain; a state gets its share of the head-room on top of
        the longest state present (trimmed containers come back with uneven states)."""
        nested = [s if isinstance(s[0], (list, tuple)) else hip.split_state(s) for s in states]
        total = max(sum(len(sub) - 1 for sub in ch) for ch in nested)
        longest = max(len(sub) - 1 for ch in nested for sub in ch)
        cap64 = longest + max(int(cap) - total, 0) // hip.NSTATES + 64
        return hip.RansState64.from_lists(nested, cap=cap64, device=self.device)

    def table_buffer(self, B, D, K, uniform=False):
        return None

    def tables(self, endpoints, mu, scale, quantbits, bits, out=None, step=None, status=None):
        return _LazyTable(endpoints, mu, scale, quantbits, step)

    def shared_table(self, endpoints, mu, scale, quantbits, bits, step=None):
        return _LazyTable(endpoints, mu[0].contiguous(), scale[0].contiguous(), quantbits, step)

    def pop(self, state, t, K, bits, centres=None):
        return hip.layer_pop64(state, t.endpoints, t.mu, t.scale, bits, t.quantbits, centres=centres, step=t.step, spec=self._sp(t.step))

    def push_params(self, state, endpoints, mu, scale, sym, quantbits, bits, step=None):
        hip.layer_push64(state, endpoints, mu, scale, sym, bits, quantbits, step=step, spec=self._sp(step))

    def push_prepare(self, state, endpoints, mu, scale, sym, quantbits, bits, step=None):
        return (endpoints, mu, scale, sym, quantbits, step)      # table + push are ONE launch here: nothing to do early

    def push_commit(self, state, token, bits):
        endpoints, mu, scale, sym, quantbits, step = token
        hip.layer_push64(state, endpoints, mu, scale, sym, bits, quantbits, step=step, spec=self._sp(step))

    def push_table(self, state, t, sym, K, bits):
        hip.layer_push64(state, t.endpoints, t.mu, t.scale, sym, bits, t.quantbits, step=t.step, spec=self._sp(t.step))


def initial_states(nchains, nwords=10000, seed=100):
    """The reference's stream initialisation (mnist_compress.py:94,158-159): numpy seeded once with
    100, then 10000 'random' uint32 words per experiment, the last shifted up to form the head."""
    np.random.seed(seed)
    out = []
    for _ in range(nchains):
        s = list(map(int, np.random.randint(low=1 << 16, high=(1 << 32) - 1, size=nwords, dtype=np.uint32)))
        s[-1] = s[-1] << 32
        out.append(s)
    return out


def reference_draws(ntest, experiments, ndatapoints, nwords=10000, seed=100):
    """What the reference's dataset scripts hold after their numpy draws, IN THEIR ORDER (mnist_compress.py:94,133-137,158
    and the cifar / imagenet siblings): `np.random.seed(100)`, then `np.random.choice(len(test_set), size=(experiments,
    ndatapoints), replace=False)` -- always drawn there: the `os.path.exists("bitstreams/<ds>/indices")` guard never hits
    because `np.save` appends ".npy" -- and only then, experiment after experiment, the 10000 initial words.  The words of
    experiment i therefore depend on the size of the test set (the permutation behind `choice` consumes the generator):
    10000 for MNIST / CIFAR-10, 50000 for ImageNet 32x32.  -> (randindices [experiments, ndatapoints], initial states).
    `initial_states()` (seed, then the words at once) stays the fallback for shapes the reference sequence cannot serve:
    fewer test images than experiments x ndatapoints, where its `choice(..., replace=False)` raises."""
    np.random.seed(seed)
    randindices = np.random.choice(ntest, size=(experiments, ndatapoints), replace=False)
    out = []
    for _ in range(experiments):
        s = list(map(int, np.random.randint(low=1 << 16, high=(1 << 32) - 1, size=nwords, dtype=np.uint32)))
        s[-1] = s[-1] << 32
        out.append(s)
    return randindices, out


class Timeline:
    """Optional per-category device timing with events on the launch stream (used by bench.py)."""

    def __init__(self, enabled=False):
        self.enabled = enabled
        self.spans = {}

    def span(self, key):
        return _Span(self, key)

    def totals(self):
        """-> {key: (seconds, count)}; synchronises."""
        if self.enabled:
            torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) * 1e-3, len(v)) for k, v in self.spans.items()}

    def reset(self):
        self.spans = {}


# BITSWAP_ROCTX=1: every coding operation (tables_z, pop_z, fc_z, push_x, net, ...) is bracketed by a roctx range, so a
# `rocprofv3 --marker-trace` timeline shows the schedule of SURVEY.md 8a#10 by name (torch.cuda.nvtx IS roctx on ROCm)
_ROCTX = os.environ.get("BITSWAP_ROCTX", "0") == "1"


class _Span:
    def __init__(self, tl, key):
        self.tl, self.key = tl, key

    def __enter__(self):
        if _ROCTX:
            try:
                torch.cuda.nvtx.range_push("bitswap:" + self.key)
            except Exception:
                pass
        if self.tl.enabled:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.tl.enabled:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.tl.spans.setdefault(self.key, []).append((self.a, b))
        if _ROCTX:
            try:
                torch.cuda.nvtx.range_pop()
            except Exception:
                pass


class _StepGraph:
    """One lock-step block step (sender or receiver) of a fixed set of chains as a hipGraph.

    With few chains per GPU (config 4 on 8 GPUs: 13 images per rank; a single demo image) a block step is ~600 short
    launches and the host cannot enqueue them as fast as the GPU retires them.  The step is a fixed sequence of
    launches on fixed buffers -- the rANS state is updated in place, every temporary comes from the graph's private
    pool -- so it is captured once and replayed per block: one hipGraphLaunch instead of ~600 launches.  The pixel
    block of the step is copied into a static tensor first (sender); the decoded block is read from one (receiver)."""

    def __init__(self, codec, state, sender):
        self.graph = torch.cuda.CUDAGraph()
        self.x = torch.zeros((state.B, codec.X), dtype=torch.int32, device=codec.device)
        self.out = None
        tl, codec.tl = codec.tl, Timeline(False)          # no timing events inside a capture
        try:
            with torch.cuda.graph(self.graph):
                if sender:
                    codec.encode_block(state, self.x)
                else:
                    self.out = codec.decode_block(state)
        finally:
            codec.tl = tl

    def replay(self, x=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.out


class BitSwapCodec:
    """Sender and receiver for B chains.

    model       bitswap_amd.model.Model (eval mode, on the backend's device); used in compress mode
    zendpoints  [nz, Z, K-1] float64, zcentres [nz, Z, K] float64 (discretize(), bins.py)
    bitswap     True: Bit-Swap schedule; False: BB-ANS (all pops, then all pushes)
    """

    def __init__(self, model, zendpoints, zcentres, quantbits=10, bitswap=True, ansbits=31, backend=None,
                 timeline=None, cdf_spec=None):
        self.backend = backend if backend is not None else HipBackend(zendpoints.device)
        self.model = model
        self.nz = model.nz
        self.q, self.K, self.bits = quantbits, 1 << quantbits, ansbits
        self.bitswap = bool(bitswap)
        self.Z, self.X = model.zdim_flat, model.xdim
        dev = zendpoints.device
        self.device = dev
        assert zendpoints.shape == (self.nz, self.Z, self.K - 1) and zcentres.shape == (self.nz, self.Z, self.K)
        self.zend = [zendpoints[i].contiguous() for i in range(self.nz)]
        self.zcen = [zcentres[i].contiguous() for i in range(self.nz)]
        xb = ImageBins(torch.float64, dev, self.X)
        self.xend, self.xcen = xb.endpoints(), xb.centres()   # expanded views, row stride 0
        # deterministic CDF specification per table (include/bitswap_hip.h): spec `cdf_spec` (default meta.DEFAULT_CDF_SPEC -- 4
        # since round 6: one reciprocal per block of 8 bins + a Newton correction per quotient; 3: round 5; 2: rounds 3-4) on every
        # set of uniform-width bins (all latent layers but the top one, and the pixels), spec 1 elsewhere.  Which tables are
        # uniform is a function of the bins alone, so a receiver built from the same bins and the same cdf_spec makes the same
        # choice; cdf_spec=1 forces spec 1 everywhere (round 1/2 streams).
        from .meta import CDF_SPECS, DEFAULT_CDF_SPEC
        cdf_spec = DEFAULT_CDF_SPEC if cdf_spec is None else int(cdf_spec)
        assert cdf_spec in CDF_SPECS
        self.cdf_spec = cdf_spec
        if cdf_spec >= 2:
            self.backend.cdf_spec = cdf_spec
        none = lambda e: None
        stepper = getattr(self.backend, "bin_step", none) if cdf_spec >= 2 else none
        self.zstep = [stepper(e) for e in self.zend]
        self.xstep = stepper(self.xend)
        self.tl = timeline or Timeline(False)
        self._cdf_bufs = {}
        # hipGraph replay of the block step: "auto" = when the chains are few enough for the step to be launch-bound
        # (single stream only; a failed capture falls back to eager launches for good)
        self.use_graphs = "auto"
        self.graph_max_chains = int(os.environ.get("BITSWAP_GRAPH_MAX_CHAINS", "128"))
        self._graphs = collections.OrderedDict()      # (state, direction) -> _StepGraph | None, the newest few
        self._graph_cap = 4
        self._graph_failures = 0
        self.graph_captures = 0                      # block steps captured so far (tests, diagnostics)
        # forked block step (round 4): with few chains the step is a latency chain, not a throughput problem, and half of that
        # chain is not a dependence at all -- generate(i)(z_i) and infer(i+1)(z_i) both need z_i only (sender), likewise
        # infer(i)(y) and generate(i-1)(y) on the receiver; only pop -> push -> pop on the stack is ordered
        # (mnist_compress.py:176-205, 293-319).  "auto": fork when the chains are few enough for the step to be
        # latency-bound (the same bound as graph replay); "1" / "0": always / never.  Scheduling only: same words.
        self.fork = os.environ.get("BITSWAP_FORK", "auto")
        self.fork_max_chains = int(os.environ.get("BITSWAP_FORK_MAX_CHAINS", "128"))
        self._aux = None                              # the second stream of the forked step
        self.forked_steps = 0                         # block steps enqueued (or captured) in the forked order (tests)
        # optional stream split (GroupedCodec): convs + table kernels on `bulk`, the serial rANS kernels
        # on `serial`; None = everything on the caller's current stream
        self.bulk = self.serial = None
        self._ev_serial = None
        # where the table kernels run when the streams are split: "bulk" (behind the convs that produce their
        # inputs) or "serial" (in front of the rANS kernel that consumes their output, next to the OTHER
        # groups' convs)
        self.tables_on = "bulk"
        # the prior p(z_L) = Logistic(0,1) table does not depend on the image: build it once
        # (the reference rebuilds it for every image, mnist_compress.py:246-251)
        one = torch.ones((1, self.Z), dtype=torch.float32, device=dev)
        self.prior_cdf = self.backend.shared_table(self.zend[-1], torch.zeros_like(one), one, self.q, self.bits,
                                                     step=self.zstep[-1])
        model.compress(True)

    # ------------------------------------------------------------------------------------------
    def new_states(self, nchains, nblocks, nwords=10000, seed=100, states=None):
        states = states if states is not None else initial_states(nchains, nwords, seed)
        # capacity: 32 bits/dim per block covers any sane model (raw pixels cost 8); a chain that
        # still outgrows it is flagged BS_ST_OVERFLOW by the kernels, never silently corrupted
        cap = max(len(s) for s in states) + nblocks * (self.X + 64) + 4 * self.Z
        return self.backend.new_state(states, cap)

    def _cdf(self, B, D, K, uniform=False):
        """Reusable cdf-row buffer per table shape (the largest, [B, Z, K+64] u32, is 1.8 GB at B=200,
        Z=2048, K=1024: resident for the whole run instead of re-allocated per layer).  A smaller chain
        count (ragged runs: chains drop out) takes a prefix of the buffer made for the largest one."""
        key = (D, K, bool(uniform))
        buf = self._cdf_bufs.get(key)
        if buf is None or buf.shape[0] < B:
            if buf is not None and self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                # replacing the buffer would free memory other captured steps point into, in the middle of a capture:
                # refuse (the capture fails, the codec falls back to eager launches); callers that know the largest
                # chain count reserve the buffers first (_reserve_tables)
                raise RuntimeError("cdf-row buffer too small inside a hipGraph capture")
            if buf is not None and self._graphs:
                torch.cuda.synchronize()   # no replay in flight while its graph is destroyed
                self._graphs.clear()       # captured block steps hold pointers into the buffer that is going away
            try:
                buf = self.backend.table_buffer(B, D, K, uniform)
            except TypeError:                        # a backend without the hand-off choice (oracle stand-in)
                buf = self.backend.table_buffer(B, D, K)
            self._cdf_bufs[key] = buf
        return buf if buf is None or buf.shape[0] == B else buf[:B]

    def _reserve_tables(self, B):
        """Size the cdf-row buffers for B chains now, so that no later, larger call replaces them under a captured step
        (ragged receivers start with the longest chain alone and grow)."""
        for zs in {s is not None for s in self.zstep}:
            self._cdf(B, self.Z, self.K, zs)
        self._cdf(B, self.X, 256, self.xstep is not None)

    # ---- stream split helpers ---------------------------------------------------------------------
    def _on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def _tables_stream(self, inputs):
        """Stream of the table kernels; when it is the serial stream, order it after the convs first."""
        if self.serial is None or self.tables_on == "bulk":
            return self.bulk
        self._serial_waits_bulk()
        self._share(inputs, self.serial)
        return self.serial

    def _serial_waits_bulk(self):
        if self.serial is not None:
            self.serial.wait_stream(self.bulk)

    def _bulk_waits_serial(self):
        """Before a conv stack that consumes symbols popped on the serial stream."""
        if self.serial is not None and self._ev_serial is not None:
            self.bulk.wait_event(self._ev_serial)

    def _mark_serial(self):
        if self.serial is not None:
            self._ev_serial = self.serial.record_event()

    def _share(self, tensors, stream):
        """Tell the caching allocator that `tensors` are also used on `stream`."""
        if stream is not None:
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(stream)

    def _spec_guard(self, step):
        """A backend object carries ONE CDF spec for its uniform-bin tables (tables / push_params / the 64-state launches read
        it); the codec's own copy decides the (f, c) of the split push and the fingerprint.  They are set together in
        __init__ -- but a second codec built on the same backend object with another spec moves the backend's: refuse to code
        rather than write tables of one spec and (f, c) of another under a fingerprint that looks valid."""
        if step is not None and getattr(self.backend, "cdf_spec", self.cdf_spec) != self.cdf_spec:
            raise RuntimeError(f"this codec codes with CDF spec {self.cdf_spec} but its backend object was re-configured to spec "
                               f"{self.backend.cdf_spec} (by another BitSwapCodec built on the same backend): give every codec "
                               "of another spec a backend object of its own")

    def _pop_layer(self, state, endpoints, centres, mu, scale, quantbits, K, key, step=None):
        self._spec_guard(step)
        ts = self._tables_stream((mu, scale))
        with self._on(ts), self.tl.span("tables_" + key):
            cdf = self.backend.tables(endpoints, mu, scale, quantbits, self.bits,
                                      out=self._cdf(mu.shape[0], mu.shape[1], K, step is not None), step=step,
                                      status=state.status)
        self._serial_waits_bulk()
        self._share((mu, scale), self.serial)    # the pivot hand-off's pop kernel reads them again, on the serial stream
        with self._on(self.serial):
            with self.tl.span("pop_" + key):
                out = self.backend.pop(state, cdf, K, self.bits, centres=centres)
            self._track_min(state)
            self._share(out, self.bulk)
        self._mark_serial()
        return out

    @staticmethod
    def _track_min(state):
        """Fewest words each chain ever held (the demo container trims the untouched initial
        words, demo_compress.py:133,159-160).  Enabled by giving the state a `min_len` tensor."""
        ml = getattr(state, "min_len", None)
        if ml is not None:
            cur = getattr(state, "len64", None)          # 64-state format: every state has its own low-water mark
            cur = state.len if cur is None or ml.dim() == 1 else cur
            torch.minimum(ml, cur.to(ml.device), out=ml)

    def _push_layer(self, state, endpoints, mu, scale, sym, quantbits, key, step=None):
        self._spec_guard(step)
        if self.serial is None or not isinstance(self.backend, HipBackend) or isinstance(self.backend, Hip64Backend):
            with self.tl.span("push_" + key):
                self.backend.push_params(state, endpoints, mu, scale, sym, quantbits, self.bits, step=step)
            return
        ts = self._tables_stream((mu, scale, sym))
        with self._on(ts), self.tl.span("fc_" + key):
            f, c = hip.logistic_fc(endpoints, mu, scale, sym, state.status, self.bits, quantbits, step=step,
                                   spec=None if step is None else self.cdf_spec)
            self._share((f, c), self.serial)
        self._serial_waits_bulk()
        with self._on(self.serial), self.tl.span("push_" + key):
            hip.rans_push(state, f, c, self.bits)

    def _net(self, fn, given):
        self._bulk_waits_serial()
        with self._on(self.bulk), self.tl.span("net"), torch.no_grad(), deterministic_convs():
            mu, scale = fn(given)
            mu, scale = mu.contiguous(), scale.contiguous()
        return mu, scale

    # ------------------------------------------------------------------------------------------
    def encode_block(self, state, x, rest_len=None):
        """Sender, one block per chain.  x [B, X] integer pixels.  If `rest_len` is a tensor it
        receives the word count right after the first bits-back pop(s) (restbits, :191-193,225-227)."""
        if self._fork_ok(state):
            return self._encode_forked(state, x, rest_len)
        for _ in self.encode_steps(state, x, rest_len):
            pass

    def decode_block(self, state):
        """Receiver, one block per chain (exact mirror).  Returns x [B, X] int32."""
        if self._fork_ok(state):
            return self._decode_forked(state)
        out = None
        for out in self.decode_steps(state):
            pass
        return out

    # ---- the forked block step ------------------------------------------------------------------------
    def _fork_ok(self, state):
        if self.fork == "0" or self.serial is not None or self.bulk is not None or not isinstance(self.backend, HipBackend):
            return False
        # (Round 5, first half: with the opt-in bf16x3 conv arithmetic the forked step was not taken -- 2-7 % of the forked runs
        # decoded one chain wrong.  Cause found in visits v-A: a compiler-packed v_pk_add_f32 of k_wino_fused lost its
        # result in lanes 48..63 beside the bf16 MFMA wavefronts; net_epilogue.hip is built without the SLP vectorizer since,
        # 0 failures in 1,500 forked runs against 22 in 800 with the packed build on the same box -- the gate is gone.)
        return self.fork == "1" or state.B <= self.fork_max_chains

    def _aux_stream(self):
        if self._aux is None:
            self._aux = torch.cuda.Stream(device=self.device)
        return self._aux

    def _encode_forked(self, state, x, rest_len=None):
        """encode_block with the dependence structure of the schedule spelled out on two streams.  S (the caller's stream)
        carries what the NEXT pop waits for -- infer(i), its table, the pop; A carries what only the push needs --
        generate(i), its (f, c), the push.  The stack is touched in the reference's order, pop_i -> push_i -> pop_(i+1), by
        stream waits; the critical path of a latent layer is pop + max(infer + table, generate + fc + push) instead of
        their sum.  BB-ANS (:206-243): every generate(i) hangs off its pop and runs under the remaining inference chain; the
        pushes follow the last pop.  Tensors that cross from S to A (z, symbols, x) stay referenced (`keep`) until S has
        waited for A at the end of the step: released earlier, the caching allocator would hand their memory to the next S
        kernel -- infer(i+2) starts right behind pop(i+1), while push(i+1) on A has yet to read the symbols of pop(i);
        nothing crosses from A to S.  Inside a hipGraph capture A joins the capture at its first wait and is joined back
        before the step ends."""
        m, nz, be = self.model, self.nz, self.backend
        S, A = torch.cuda.current_stream(self.device), self._aux_stream()
        self.forked_steps += 1
        x = x.to(self.device, torch.int32).contiguous()
        given = be.centres(self.xcen, x)
        keep = [x]
        if self.bitswap:
            zsym = None
            for zi in range(nz):
                mu, sc = self._net(m.infer(zi), given)
                with self.tl.span("tables_z"):
                    cdf = be.tables(self.zend[zi], mu, sc, self.q, self.bits, out=self._cdf(mu.shape[0], mu.shape[1], self.K, self.zstep[zi] is not None),
                                    step=self.zstep[zi], status=state.status)
                if zi:
                    S.wait_stream(A)                                  # push_(zi-1) has left the stack
                with self.tl.span("pop_z"):
                    zsymtop, z = be.pop(state, cdf, self.K, self.bits, centres=self.zcen[zi])
                keep += [zsymtop, z]
                self._track_min(state)
                if rest_len is not None and zi == 0:
                    self._snap(rest_len, state)
                A.wait_stream(S)
                with torch.cuda.stream(A):
                    mu, sc = self._net(m.generate(zi), z)
                    if zi == 0:
                        self._push_layer(state, self.xend, mu, sc, x, 8, "x", self.xstep)
                    else:
                        self._push_layer(state, self.zend[zi - 1], mu, sc, zsym, self.q, "z", self.zstep[zi - 1])
                zsym, given = zsymtop, z
        else:
            syms, zs, pending = [], [], []
            for zi in range(nz):
                mu, sc = self._net(m.infer(zi), given)
                with self.tl.span("tables_z"):
                    cdf = be.tables(self.zend[zi], mu, sc, self.q, self.bits, out=self._cdf(mu.shape[0], mu.shape[1], self.K, self.zstep[zi] is not None),
                                    step=self.zstep[zi], status=state.status)
                with self.tl.span("pop_z"):
                    s, z = be.pop(state, cdf, self.K, self.bits, centres=self.zcen[zi])
                self._track_min(state)
                syms.append(s)
                zs.append(z)
                given = z
                if zi == nz - 1 and rest_len is not None:
                    self._snap(rest_len, state)
                A.wait_stream(S)
                with torch.cuda.stream(A):                            # generate(zi) under the rest of the inference chain
                    mu, sc = self._net(m.generate(zi), z)
                    with self.tl.span("fc_x" if zi == 0 else "fc_z"):
                        if zi == 0:
                            pending.append(be.push_prepare(state, self.xend, mu, sc, x, 8, self.bits, step=self.xstep))
                        else:
                            pending.append(be.push_prepare(state, self.zend[zi - 1], mu, sc, syms[zi - 1], self.q, self.bits,
                                                           step=self.zstep[zi - 1]))
            with torch.cuda.stream(A):                                # A has waited for the last pop
                for zi, tok in enumerate(pending):
                    with self.tl.span("push_x" if zi == 0 else "push_z"):
                        be.push_commit(state, tok, self.bits)
            zsymtop = syms[-1]
        S.wait_stream(A)
        with self.tl.span("push_prior"):
            be.push_table(state, self.prior_cdf, zsymtop, self.K, self.bits)
        del keep                                                      # S is behind everything A read

    def _decode_forked(self, state):
        """decode_block in the forked order (mirror of _encode_forked, :293-354): S carries generate(i), its table and the
        pop; A carries infer(i), its (f, c) and the push."""
        m, nz, be = self.model, self.nz, self.backend
        S, A = torch.cuda.current_stream(self.device), self._aux_stream()
        self.forked_steps += 1
        with self.tl.span("pop_prior"):
            zsymtop, z = be.pop(state, self.prior_cdf, self.K, self.bits, centres=self.zcen[-1])
        self._track_min(state)
        keep = [zsymtop, z]                                           # see _encode_forked: alive until S has waited for A

        def pop_under(zi, mu, sc):
            if zi == 0:
                ends, cens, q, K, key, step = self.xend, self.xcen, 8, 256, "x", self.xstep
            else:
                ends, cens, q, K, key, step = self.zend[zi - 1], self.zcen[zi - 1], self.q, self.K, "z", self.zstep[zi - 1]
            with self.tl.span("tables_" + key):
                cdf = be.tables(ends, mu, sc, q, self.bits, out=self._cdf(mu.shape[0], mu.shape[1], K, step is not None),
                                step=step, status=state.status)
            return cdf, K, cens, key

        if self.bitswap:
            for k, zi in enumerate(reversed(range(nz))):
                mu, sc = self._net(m.generate(zi), z)
                cdf, K, cens, key = pop_under(zi, mu, sc)
                if k:
                    S.wait_stream(A)                                  # the previous layer's push has left the stack
                with self.tl.span("pop_" + key):
                    sym, given = be.pop(state, cdf, K, self.bits, centres=cens)
                keep += [sym, given]
                self._track_min(state)
                A.wait_stream(S)
                with torch.cuda.stream(A):
                    mu, sc = self._net(m.infer(zi), given)
                    self._push_layer(state, self.zend[zi], mu, sc, zsymtop, self.q, "z", self.zstep[zi])
                zsymtop, z = sym, given
            S.wait_stream(A)
            del keep
            return zsymtop
        syms, cens_l, pending = [zsymtop], [z], []
        for k, zi in enumerate(reversed(range(nz))):
            mu, sc = self._net(m.generate(zi), cens_l[-1])
            cdf, K, cens, key = pop_under(zi, mu, sc)
            with self.tl.span("pop_" + key):
                s, c = be.pop(state, cdf, K, self.bits, centres=cens)
            self._track_min(state)
            syms.append(s)
            cens_l.append(c)
            A.wait_stream(S)
            with torch.cuda.stream(A):                                # infer(zi) under the rest of the generative chain
                mu, sc = self._net(m.infer(zi), c)
                with self.tl.span("fc_z"):
                    pending.append(be.push_prepare(state, self.zend[zi], mu, sc, syms[k], self.q, self.bits, step=self.zstep[zi]))
        with torch.cuda.stream(A):                                    # A has waited for the last pop
            for tok in pending:
                with self.tl.span("push_z"):
                    be.push_commit(state, tok, self.bits)
        S.wait_stream(A)
        return syms[-1]

    def _graph_ok(self, state):
        if not self.use_graphs or self.serial is not None or self.bulk is not None or not isinstance(self.backend, HipBackend):
            return False
        return self.use_graphs is True or state.B <= self.graph_max_chains

    def _graphed(self, state, sender):
        """The captured step for this state (same tensors on every replay), or None if capture is not possible.
        A graph pins its state (a stack of up to GBs) and a private allocator pool: the cache is LRU-bounded and
        release_graphs() drops a state's graphs when its run ends."""
        key = (id(state.head), state.B, sender)
        if key in self._graphs:
            self._graphs.move_to_end(key)            # least recently USED goes first
            return self._graphs[key]
        try:
            torch.cuda.synchronize()
            g = _StepGraph(self, state, sender)
            g._keep = state                          # the graph holds raw pointers into the state's tensors
            self.graph_captures += 1
        except Exception as e:                       # e.g. a library call that cannot be captured on this stack
            import warnings
            self._graph_failures += 1
            warnings.warn(f"hipGraph capture of the block step failed ({e!r}); running this state eagerly"
                          + ("; giving up on graph replay for this codec" if self._graph_failures >= 2 else ""))
            torch.cuda.synchronize()
            g = None
            if self._graph_failures >= 2:            # one failure may be that state's shape; two are the stack
                self.use_graphs = False
        self._graphs[key] = g
        while len(self._graphs) > self._graph_cap:
            torch.cuda.synchronize()
            self._graphs.popitem(last=False)
        return g

    def release_graphs(self, state=None):
        """Drop the captured block steps of `state` (and of its prefix views), or all of them: their private memory pools
        and the references that keep finished states alive go with them.  compress()/decompress() and the ragged
        drivers call it when their run ends; a long-lived codec (a server, cli.compress looping over nz) therefore holds
        graphs only while a run is in flight."""
        if not self._graphs:
            return
        base = None if state is None else state.stack.data_ptr()
        dead = [k for k, g in self._graphs.items()
                if state is None or g is None or g._keep.stack.data_ptr() == base]
        if dead:
            torch.cuda.synchronize()
            for k in dead:
                del self._graphs[k]

    def _graph_room(self, n):
        """Let up to n graphs coexist (ragged runs: one per distinct number of active chains and direction)."""
        self._graph_cap = max(4, min(int(n), 64))

    def prepare_graphs(self, state):
        """Capture both block-step graphs for `state` now (nothing is executed), e.g. before a timed region.  The
        eager path must have run once in each direction before (library warm-up, buffers)."""
        if self._graph_ok(state):
            self._graphed(state, True)
            self._graphed(state, False)

    def encode_block_fast(self, state, x, first=False):
        """encode_block, replayed from a hipGraph when that pays (never for the block that records restbits)."""
        g = self._graphed(state, True) if (not first and self._graph_ok(state)) else None
        if g is None:
            self.encode_block(state, x)
        else:
            g.replay(x)

    def decode_block_fast(self, state):
        g = self._graphed(state, False) if self._graph_ok(state) else None
        return self.decode_block(state) if g is None else g.replay().clone()

    def _snap(self, dst, state):
        with self._on(self.serial):
            dst.copy_(state.len)

    def encode_steps(self, state, x, rest_len=None):
        """Generator form of encode_block: yields after every coding operation has been enqueued, so a
        scheduler (GroupedCodec) can interleave the enqueue order of several chain groups."""
        m, nz = self.model, self.nz
        with self._on(self.bulk):
            x = x.to(self.device, torch.int32).contiguous()
            given = self.backend.centres(self.xcen, x)            # xcentres[xrange, x] -> float32
        if self.bitswap:
            zsym = None
            for zi in range(nz):
                mu, sc = self._net(m.infer(zi), given)
                zsymtop, z = self._pop_layer(state, self.zend[zi], self.zcen[zi], mu, sc, self.q, self.K, "z", self.zstep[zi])
                if rest_len is not None and zi == 0:
                    self._snap(rest_len, state)
                yield
                mu, sc = self._net(m.generate(zi), z)
                if zi == 0:
                    self._push_layer(state, self.xend, mu, sc, x, 8, "x", self.xstep)
                else:
                    self._push_layer(state, self.zend[zi - 1], mu, sc, zsym, self.q, "z", self.zstep[zi - 1])
                yield
                zsym, given = zsymtop, z
        else:
            syms, zs = [], []
            for zi in range(nz):
                mu, sc = self._net(m.infer(zi), given)
                s, z = self._pop_layer(state, self.zend[zi], self.zcen[zi], mu, sc, self.q, self.K, "z", self.zstep[zi])
                syms.append(s)
                zs.append(z)
                given = z
                yield
            if rest_len is not None:
                self._snap(rest_len, state)
            for zi in range(nz):
                mu, sc = self._net(m.generate(zi), zs[zi])
                if zi == 0:
                    self._push_layer(state, self.xend, mu, sc, x, 8, "x", self.xstep)
                else:
                    self._push_layer(state, self.zend[zi - 1], mu, sc, syms[zi - 1], self.q, "z", self.zstep[zi - 1])
                yield
            zsymtop = syms[-1]
        with self._on(self.serial), self.tl.span("push_prior"):
            self.backend.push_table(state, self.prior_cdf, zsymtop, self.K, self.bits)
        yield

    def decode_steps(self, state):
        """Generator form of decode_block; the last yielded value is x [B, X] int32."""
        m, nz = self.model, self.nz
        with self._on(self.serial):
            with self.tl.span("pop_prior"):
                # prior table [Z, ld] is shared by every chain (chain stride 0)
                zsymtop, z = self.backend.pop(state, self.prior_cdf, self.K, self.bits, centres=self.zcen[-1])
            self._track_min(state)
            self._share((zsymtop, z), self.bulk)
        self._mark_serial()
        yield None
        if self.bitswap:
            for zi in reversed(range(nz)):
                mu, sc = self._net(m.generate(zi), z)
                if zi == 0:
                    sym, given = self._pop_layer(state, self.xend, self.xcen, mu, sc, 8, 256, "x", self.xstep)
                else:
                    sym, given = self._pop_layer(state, self.zend[zi - 1], self.zcen[zi - 1], mu, sc, self.q,
                                                 self.K, "z", self.zstep[zi - 1])
                yield None
                mu, sc = self._net(m.infer(zi), given)
                self._push_layer(state, self.zend[zi], mu, sc, zsymtop, self.q, "z", self.zstep[zi])
                yield None
                zsymtop, z = sym, given
            yield zsymtop
            return
        syms, cens = [zsymtop], [z]
        for zi in reversed(range(nz)):
            mu, sc = self._net(m.generate(zi), cens[-1])
            if zi == 0:
                s, c = self._pop_layer(state, self.xend, self.xcen, mu, sc, 8, 256, "x", self.xstep)
            else:
                s, c = self._pop_layer(state, self.zend[zi - 1], self.zcen[zi - 1], mu, sc, self.q, self.K, "z", self.zstep[zi - 1])
            syms.append(s)
            cens.append(c)
            yield None
        # syms = [z_L, ..., z_1, x]; push z_L .. z_1 back under q(z_i | z_{i-1} or x)
        for k, zi in enumerate(reversed(range(nz))):
            mu, sc = self._net(m.infer(zi), cens[k + 1])
            self._push_layer(state, self.zend[zi], mu, sc, syms[k], self.q, "z", self.zstep[zi])
            yield None
        yield syms[-1]

    # ------------------------------------------------------------------------------------------
    def compress(self, images, state=None, nwords=10000, seed=100):
        """images [B, nblocks, X] integers -> (state, metrics).  metrics follow the reference's bit
        accounting (mnist_compress.py:253-261): nets, cma, total as float arrays [B, nblocks]."""
        images = torch.as_tensor(images)
        B, n, X = images.shape
        assert X == self.X
        if state is None:
            state = self.new_states(B, n, nwords, seed)
        init_len = state.len.clone()
        rest_len = torch.zeros_like(state.len)
        lens = torch.zeros((n, B), dtype=torch.int32, device=state.len.device)
        for xi in range(n):
            if xi == 0:
                self.encode_block(state, images[:, xi], rest_len)      # eager: records restbits, warms every library up
            else:
                self.encode_block_fast(state, images[:, xi])
            self._snap(lens[xi], state)
        self.backend.check(state, "compress")
        self.release_graphs(state)
        lens, init_len, rest_len = lens.cpu().numpy().T.astype(np.int64), init_len.cpu().numpy(), rest_len.cpu().numpy()
        added = (lens - init_len[:, None]) * 32                      # totaladdedbits (:254)
        total = (lens - rest_len[:, None] + 1) * 32                  # totalbits (:255): len(restbits)-1 = rest words
        cumnet = added / X
        nets = np.diff(np.concatenate([np.zeros((B, 1)), cumnet], axis=1), axis=1)   # (:258)
        cma = total / (X * np.arange(1, n + 1)[None, :])                             # (:260)
        return state, dict(nets=nets, cma=cma, total=total.astype(np.float64), rest_len=rest_len, init_len=init_len)

    # ---- chains of different lengths (one image = one chain, imagenetcrop_compress.py:279-300) ---------
    def stage_ragged(self, chains):
        """chains: list of integer tensors [n_i, X] -> (x [B, nmax, X] int32 on the device, rows sorted by decreasing length
        and zero-padded; order; ns): what compress_ragged codes.  Callers that time the coding stage first."""
        n = [int(c.shape[0]) for c in chains]
        assert min(n) >= 1
        order = sorted(range(len(chains)), key=lambda i: (-n[i], i))
        ns = [n[i] for i in order]
        x = torch.zeros((len(chains), ns[0], self.X), dtype=torch.int32, device=self.device)
        for k, i in enumerate(order):
            x[k, : ns[k]] = torch.as_tensor(chains[i]).to(self.device, torch.int32)
        return x, order, ns

    def compress_ragged(self, chains, state=None, nwords=10000, seed=100, same_init=True, staged=None):
        """chains: list of integer tensors [n_i, X].  All chains run in lock-step; with the chains sorted
        by decreasing length the active set at block xi is a PREFIX of the batch, so the kernels simply see
        fewer chains as the short ones finish (no masking, no padding work).  Returns (state, order,
        metrics) where state/metrics rows follow `order` (indices into `chains`, longest first).
        same_init: every chain starts from the same initial words, like the crop script (:249,122).
        staged: the result of stage_ragged(chains), if the caller made it ahead of time."""
        x, order, ns = staged if staged is not None else self.stage_ragged(chains)
        B, nmax = len(ns), ns[0]
        if state is None:
            one = initial_states(1, nwords, seed)[0]
            states = [list(one) for _ in range(B)] if same_init else initial_states(B, nwords, seed)
            state = self.new_states(B, nmax, states=states)
        init_len = state.len.clone()
        rest_len = torch.zeros_like(state.len)
        active = [sum(1 for m in ns if m > xi) for xi in range(nmax)]
        # blocks coded with k chains active: a graph per k that is worth one.  Captures cost tens of ms each, so only for
        # a handful of chains with long runs (a single demo image: 8.0 -> 5.4 ms per block in the 64-state format); with
        # many images the active count changes every few blocks and eager launches win (profiles/r02o_crop_runs.txt)
        runs = collections.Counter(active)
        worth = {k for k, v in runs.items() if v >= 8} if B <= 16 else set()
        self._graph_room(2 * len(worth))
        for xi in range(nmax):
            k = active[xi]
            if xi == 0:
                self.encode_block(state.prefix(k), x[:k, xi], rest_len)
            elif k in worth:
                self.encode_block_fast(state.prefix(k), x[:k, xi])
            else:
                self.encode_block(state.prefix(k), x[:k, xi])
        self.backend.check(state, "compress_ragged")
        self.release_graphs(state)
        lens, init_len, rest_len = state.len.cpu().numpy().astype(np.int64), init_len.cpu().numpy(), rest_len.cpu().numpy()
        nsa = np.array(ns, dtype=np.int64)
        total = (lens - rest_len + 1) * 32                             # totalbits of the whole chain (:255)
        return state, order, dict(total=total.astype(np.float64), cma=total / (self.X * nsa),
                                  net=(lens - init_len) * 32 / (self.X * nsa), nblocks=nsa, rest_len=rest_len)

    def decompress_ragged(self, state, nblocks):
        """Receiver for compress_ragged: nblocks[k] blocks for chain k (non-increasing).  -> list of
        [n_k, X] int32 tensors; state is unwound in place."""
        ns = [int(v) for v in nblocks]
        assert all(a >= b for a, b in zip(ns, ns[1:])), "chains must be sorted by decreasing length"
        nmax = ns[0]
        out = [[None] * m for m in ns]
        active = [sum(1 for m in ns if m > xi) for xi in range(nmax)]
        runs = collections.Counter(active)
        self._reserve_tables(len(ns))
        worth = {k for k, v in runs.items() if v >= 8} if len(ns) <= 16 else set()
        self._graph_room(2 * len(worth))
        for xi in reversed(range(nmax)):
            k = active[xi]
            eager = xi == nmax - 1 or k not in worth        # the first receiver step warms the libraries up
            xb = self.decode_block(state.prefix(k)) if eager else self.decode_block_fast(state.prefix(k))
            for c in range(k):
                out[c][xi] = xb[c]
        self.backend.check(state, "decompress_ragged")
        self.release_graphs(state)
        return [torch.stack(o, dim=0) for o in out]

    def decompress(self, state, nblocks):
        """-> images [B, nblocks, X] int32 (blocks in original order); state is unwound in place."""
        out = [None] * nblocks
        for xi in reversed(range(nblocks)):
            # the first step runs eagerly (library warm-up before a capture), the rest replay the graph when it pays
            out[xi] = self.decode_block(state) if xi == nblocks - 1 else self.decode_block_fast(state)
        self.backend.check(state, "decompress")
        self.release_graphs(state)
        return torch.stack(out, dim=1)


class GroupedCodec:
    """Software pipelining across chain groups.

    The serial rANS kernels keep one wavefront busy per chain -- a few percent of an MI355X at 100
    chains -- while the conv stacks and the table kernels want the whole chip.  The chains are split
    into G groups, each on its own HIP stream (default since round 3; BITSWAP_GROUP_STREAMS=0 gives every group a bulk
    stream for convs and table kernels plus a serial stream for pop/push, the round-2 arrangement).  The enqueue order is
    interleaved at coding-operation granularity, so while group A pops, group B's convs are already executing, and the
    HBM-bound transform passes and the float64 table kernels of one group fill in under the matrix-core GEMMs of the other.
    Chains never interact: results are identical to coding each group on its own.
    """

    def __init__(self, model, zendpoints, zcentres, groups=2, **kw):
        self.codecs = [BitSwapCodec(model, zendpoints, zcentres, **kw) for _ in range(groups)]
        self.device = zendpoints.device
        self.X, self.Z, self.K = self.codecs[0].X, self.codecs[0].Z, self.codecs[0].K
        self.bulk = None
        self.group_streams = None
        gs = os.environ.get("BITSWAP_GROUP_STREAMS", "auto")
        if int(os.environ.get("BITSWAP_SERIAL_CUS", "0")):
            gs = "0"
        be = self.codecs[0].backend
        one_stream = gs == "1" or (gs == "auto" and (isinstance(be, Hip64Backend) or getattr(be, "pivot", False)))
        if groups > 1 and one_stream:
            # ONE stream per group, the groups overlap each other.  64-state format: the coding kernels are fused, there is
            # nothing serial to split off.  Reference format: since the pop kernel of big batches no longer streams whole
            # rows from HBM (BS_LAYOUT_PIVOT) it is cheap to keep in line, and two in-order streams beat the bulk + serial
            # split (800 chains: 152.5 vs 157.9 ms per step, profiles/r03r; round 2, with the row-reading pop: 164.8 vs 164.2)
            self.group_streams = [torch.cuda.Stream(device=self.device) for _ in self.codecs]
        elif groups > 1:
            # BITSWAP_SERIAL_CUS=n (round 5 experiment, VERDICT r4 #1c; scheduling only): the serial streams -- the pop / push
            # kernels, one wavefront per chain -- restricted to n compute units (mask bits 0 .. n-1: n / 8 per XCD) and the bulk
            # streams to the others, so that the coder wavefronts do not sit on the SIMDs of the GEMMs and table kernels.
            # Takes the bulk + serial arrangement (BITSWAP_GROUP_STREAMS=0); set BITSWAP_GEMM_CUS to the bulk CU count as well.
            ncu = int(os.environ.get("BITSWAP_SERIAL_CUS", "0"))
            total_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            self._masked = []

            def stream(first, n):
                if not ncu:
                    return torch.cuda.Stream(device=self.device)
                m = hip.MaskedStream(first, n, self.device)
                self._masked.append(m)
                return m.stream
            self.bulk = stream(ncu, total_cus - ncu)
            own = os.environ.get("BITSWAP_BULK_PER_GROUP", "1") == "1"
            for c in self.codecs:
                c.bulk = stream(ncu, total_cus - ncu) if own else self.bulk
                c.serial = stream(0, ncu)

    def split(self, n):
        g = len(self.codecs)
        base, extra = divmod(n, g)
        sizes = [base + (1 if i < extra else 0) for i in range(g)]
        offs = np.cumsum([0] + sizes)
        return [slice(int(offs[i]), int(offs[i + 1])) for i in range(g)]

    def new_states(self, nchains, nblocks, nwords=10000, seed=100, states=None):
        states = states if states is not None else initial_states(nchains, nwords, seed)
        return [c.new_states(sl.stop - sl.start, nblocks, states=states[sl])
                for c, sl in zip(self.codecs, self.split(nchains))]

    def _streams(self):
        if self.group_streams is not None:
            return self.group_streams
        return [] if self.bulk is None else list({id(t): t for t in [c.bulk for c in self.codecs] + [c.serial for c in self.codecs]}.values())

    def _fork(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams():
            s.wait_stream(cur)
        for c in self.codecs:
            c._ev_serial = None

    def _join(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams():
            cur.wait_stream(s)

    def _graphs_per_group(self, states):
        return (self.group_streams is not None and os.environ.get("BITSWAP_GROUP_GRAPHS", "1") == "1"
                and all(c._graph_ok(st) for c, st in zip(self.codecs, states)))

    def _round_robin(self, gens, skew=1):
        """Advance the generators in turn; group g starts g*skew operations late so that one group's
        serial kernel coincides with another group's conv stack."""
        live = list(range(len(gens)))
        last = [None] * len(gens)
        tick = 0
        ctx = [torch.cuda.stream(t) for t in self.group_streams] if self.group_streams is not None else None
        while live:
            for g in list(live):
                if tick >= g * skew:
                    try:
                        with (ctx[g] if ctx is not None else contextlib.nullcontext()):
                            last[g] = next(gens[g])
                    except StopIteration:
                        live.remove(g)
            tick += 1
        return last

    def encode_blocks(self, states, images, rest_lens=None):
        """images [B, n, X]: n block steps of every group, enqueue order interleaved per coding op."""
        B, n, _ = images.shape
        sls = self.split(B)
        if len(self.codecs) == 1:        # one stream: block after block, replayed from a hipGraph when launch-bound
            c, st = self.codecs[0], states[0]
            for xi in range(n):
                if rest_lens is not None and xi == 0:
                    c.encode_block(st, images[:, xi], rest_lens[0])
                else:
                    c.encode_block_fast(st, images[:, xi], first=(xi == 0 and not c._graphs))
            return
        if self._graphs_per_group(states):
            # few chains per group: every group's block step replayed from its own hipGraph on its own stream -- one
            # launch per group and block instead of ~600 each (100 chains in 4 groups enqueued operation by operation are
            # host-bound: 41.5 ms per step against 29.0 in one group, profiles/r03K)
            self._fork()
            for xi in range(n):
                for g, (c, st) in enumerate(zip(self.codecs, states)):
                    with torch.cuda.stream(self.group_streams[g]):
                        if rest_lens is not None and xi == 0:
                            c.encode_block(st, images[sls[g], xi], rest_lens[g])
                        else:
                            c.encode_block_fast(st, images[sls[g], xi], first=(xi == 0 and not c._graphs))
            self._join()
            return
        self._fork()

        def chain_of_blocks(g):
            c, st = self.codecs[g], states[g]
            for xi in range(n):
                yield from c.encode_steps(st, images[sls[g], xi],
                                          rest_lens[g] if (rest_lens is not None and xi == 0) else None)
        self._round_robin([chain_of_blocks(g) for g in range(len(self.codecs))])
        self._join()

    def decode_blocks(self, states, n):
        """-> [B, n, X] int32, blocks in original order."""
        outs = [[None] * n for _ in self.codecs]
        if len(self.codecs) == 1:
            c, st = self.codecs[0], states[0]
            for xi in reversed(range(n)):
                warm = any(k[2] is False for k in c._graphs)        # the first receiver step ever runs eagerly
                outs[0][xi] = c.decode_block_fast(st) if warm or not c._graph_ok(st) else c.decode_block(st)
                if not warm and c._graph_ok(st):
                    c._graphed(st, False)
            return torch.stack(outs[0], dim=1)
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if self._graphs_per_group(states):
            self._fork()
            for xi in reversed(range(n)):
                for g, (c, st) in enumerate(zip(self.codecs, states)):
                    with torch.cuda.stream(self.group_streams[g]):
                        warm = any(k[2] is False for k in c._graphs)        # the first receiver step ever runs eagerly
                        x = c.decode_block_fast(st) if warm else c.decode_block(st)
                        if not warm:
                            c._graphed(st, False)
                        x.record_stream(main)
                        outs[g][xi] = x
            self._join()
            return torch.cat([torch.stack(o, dim=1) for o in outs], dim=0)
        self._fork()

        def chain_of_blocks(g):
            c, st = self.codecs[g], states[g]
            for xi in reversed(range(n)):
                x = None
                for x in c.decode_steps(st):
                    yield
                if x is not None and x.is_cuda:
                    x.record_stream(main)
                outs[g][xi] = x
        self._round_robin([chain_of_blocks(g) for g in range(len(self.codecs))])
        self._join()
        return torch.cat([torch.stack(o, dim=1) for o in outs], dim=0)

    def check(self, states, what="grouped"):
        for c, st in zip(self.codecs, states):
            c.backend.check(st, what)

    @staticmethod
    def to_lists(states):
        return [s for st in states for s in st.to_lists()]
