"""GPU: the HIP kernels (through the C ABI) against the oracle and the reference-generated
fixtures.  Integer work must be bit-exact; the float64 CDF must be bit-exact against the
oracle's deterministic mode and within the stated budget of torch's sigmoid."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import chain_tables, linear_to_wave, reference_init_state, words_to_state

pytestmark = pytest.mark.gpu
DEV = "cuda"


def hip():
    from bitswap_amd import hip as h
    return h


def u32(t):
    return t.cpu().numpy().view(np.uint32)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def test_selftest_wave_primitives():
    assert hip().selftest() == 0


def test_sigmoid_bit_exact_vs_oracle():
    rng = np.random.RandomState(0)
    t = np.concatenate([rng.uniform(-45, 45, 200000), rng.uniform(-1, 1, 50000), rng.uniform(-800, 800, 5000),
                        [0.0, -0.0, 700.0, -700.0, 1e6, -1e6, 1e-300, -1e-300, 1e12, -1e12, 1e300, -1e300,
                         708.0, -708.0, 745.2, -745.2, 3.0e9, -3.0e9]])
    got = hip().sigmoid_f64(dev(t)).cpu().numpy()
    want = O.det_sigmoid(t)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("name", ["ztop", "zuni", "x"])
@pytest.mark.parametrize("aligned", [False, True])
def test_table_rows_bit_exact_vs_reference(golden, name, aligned):
    g = golden("tables_rans.npz")
    q = int(g[f"{name}_quantbits"])
    K = g[f"{name}_pmf_f64"].shape[1]
    ld = hip().aligned_ld(K) if aligned else K + 1
    f, cdf, st = hip().table_rows(dev(g[f"{name}_pmf_f64"]), 31, q, ld=ld)
    assert int(st.abs().max()) == 0
    assert np.array_equal(u32(f), g[f"{name}_f"])
    assert np.array_equal(u32(cdf)[:, : K + 1], g[f"{name}_cdf"])


def test_table_rows_generic_k_and_ties(golden):
    g = golden("tables_rans.npz")
    f, cdf, st = hip().table_rows(dev(g["tie_pmf_f64"]), 31, 4)
    assert np.array_equal(u32(f), g["tie_f"]) and np.array_equal(u32(cdf), g["tie_cdf"])
    # ragged K values through the generic kernel, against the oracle
    rng = np.random.RandomState(5)
    for K in (1, 2, 63, 64, 65, 200, 1000):
        p = rng.dirichlet(np.ones(K) * 0.3, size=7)
        fo, co, rc = O.tables(p, 31, 4)
        f, cdf, st = hip().table_rows(dev(p), 31, 4)
        assert np.array_equal(u32(f), fo) and np.array_equal(u32(cdf), co)


@pytest.mark.parametrize("name", ["ztop", "zuni", "x"])
@pytest.mark.parametrize("ptype", [torch.float32, torch.float64])
def test_logistic_tables_vs_oracle_and_reference(golden, name, ptype):
    g = golden("tables_rans.npz")
    q = int(g[f"{name}_quantbits"])
    e, mu, sc = g[f"{name}_endpoints"], g[f"{name}_mu"], g[f"{name}_scale"]   # mu/scale are f32-representable
    K = e.shape[1] + 1
    B = 5
    rng = np.random.RandomState(1)
    mus = np.stack([mu] + [(mu + rng.randn(*mu.shape) * 0.1).astype(np.float32).astype(np.float64) for _ in range(B - 1)])
    scs = np.stack([sc] + [np.clip(sc * rng.uniform(0.8, 1.2, sc.shape), 1e-3, None).astype(np.float32).astype(np.float64)
                           for _ in range(B - 1)])
    if ptype == torch.float32:   # the kernel converts float32 parameters to float64 exactly
        mus, scs = mus.astype(np.float32).astype(np.float64), scs.astype(np.float32).astype(np.float64)
    exact_inputs = np.array_equal(mus[0], mu) and np.array_equal(scs[0], sc)
    for ld in (K + 1, hip().aligned_ld(K)):
        cdf = hip().logistic_tables(dev(e), dev(mus, ptype), dev(scs, ptype), 31, q, ld=ld)
        got = u32(cdf)[:, :, : K + 1]
        for b in range(B):
            pmf = O.logistic_pmf(e, mus[b], scs[b], O.MODE_DET)
            _, want, rc = O.tables(pmf, 31, q)
            assert np.array_equal(got[b], want), (name, b, ld)
        # chain 0 uses the fixture's parameters: compare with the reference's own integer table
        ref = g[f"{name}_cdf"].astype(np.int64)
        if exact_inputs:
            mism = (np.diff(got[0].astype(np.int64), axis=1) != np.diff(ref, axis=1))
            assert mism.mean() <= 2e-6


def test_logistic_tables_shared_endpoint_row():
    """ImageBins / top-layer bins are expanded views (row stride 0), rand.py:134-153."""
    e1 = ((np.arange(1, 256) - 127.5) / 127.5) - 1. / 255.
    D, B = 70, 3
    rng = np.random.RandomState(2)
    mu = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    sc = rng.uniform(0.003, 0.5, (B, D)).astype(np.float32)
    et = dev(e1).unsqueeze(0).expand(D, -1)
    cdf = u32(hip().logistic_tables(et, dev(mu), dev(sc), 31, 8))
    full = np.broadcast_to(e1[None], (D, 255))
    for b in range(B):
        _, want, _ = O.tables(O.logistic_pmf(full, mu[b].astype(np.float64), sc[b].astype(np.float64), O.MODE_DET), 31, 8)
        assert np.array_equal(cdf[b][:, :257], want)


@pytest.mark.parametrize("K", [64, 128, 256, 512, 1024, 2048])
def test_logistic_fc_matches_tables(K):
    q = int(np.log2(K))
    rng = np.random.RandomState(K)
    D, B = 37, 6
    lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
    e = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
    mu = rng.randn(B, D).astype(np.float32)
    sc = rng.uniform(0.1, 1, (B, D)).astype(np.float32)
    sym = rng.randint(0, K, (B, D)).astype(np.int32)
    sym[0, :4] = [0, K - 1, 1, K - 2]
    status = torch.zeros(B, dtype=torch.int32, device=DEV)
    f, c = hip().logistic_fc(dev(e), dev(mu), dev(sc), dev(sym), status, 31, q)
    cdf = u32(hip().logistic_tables(dev(e), dev(mu), dev(sc), 31, q))
    f, c = u32(f), u32(c)
    for b in range(B):
        _, want, _ = O.tables(O.logistic_pmf(e, mu[b].astype(np.float64), sc[b].astype(np.float64), O.MODE_DET), 31, q)
        assert np.array_equal(cdf[b][:, : K + 1], want)
        rows = np.arange(D)
        assert np.array_equal(c[b], want[rows, sym[b]])
        assert np.array_equal(f[b], want[rows, sym[b] + 1] - want[rows, sym[b]])
    assert int(status.abs().max()) == 0
    # out-of-range symbol is reported, not executed
    bad = sym.copy()
    bad[2, 5] = K
    hip().logistic_fc(dev(e), dev(mu), dev(sc), dev(bad), status, 31, q)
    assert status.cpu().tolist() == [0, 0, hip().ST_BADSYMBOL, 0, 0, 0]


def wave_table(h, rows, K, bits=31):
    """Reference cdf rows as a BS_LAYOUT_WAVE device tensor (whole 64-row chunks, identity rows appended)."""
    t = dev(linear_to_wave(rows, K, bits).view(np.int32))
    t.bs_layout = h.LAYOUT_WAVE
    return t


@pytest.mark.parametrize("name", ["ztop", "zuni", "x"])
@pytest.mark.parametrize("form", ["linear", "aligned", "wave"])
def test_rans_words_bit_exact_vs_reference(golden, name, form):
    """ANS.decode / ANS.encode word streams of the reference against every table form the kernels take: the
    reference's rows (ld = K+1), 16-byte rows, and BS_LAYOUT_WAVE -- the form the production pop kernel
    (k_rans_pop_wave) and the prior push read."""
    h = hip()
    g = golden("tables_rans.npz")
    K = g[f"{name}_cdf"].shape[1] - 1
    D = g[f"{name}_cdf"].shape[0]
    B = 3  # the same chain three times + per-chain tables
    if form == "wave":
        cdf = wave_table(h, g[f"{name}_cdf"], K)
        Dp = cdf.shape[0]
        per_chain = cdf.unsqueeze(0).expand(B, -1, -1).contiguous()
        per_chain.bs_layout = h.LAYOUT_WAVE
    else:
        ld = h.aligned_ld(K) if form == "aligned" else K + 1
        tab = np.zeros((D, ld), dtype=np.uint32)
        tab[:, : K + 1] = g[f"{name}_cdf"]
        cdf, Dp = dev(tab.view(np.int32)), D
        per_chain = cdf.unsqueeze(0).expand(B, -1, -1).contiguous()

    def padded(sym):   # symbols of the appended identity rows: 0
        out = np.zeros((B, Dp), dtype=np.int32)
        out[:, :D] = sym
        return dev(out)
    s0 = words_to_state(g[f"{name}_state0"])
    st = h.RansState.from_lists([s0] * B, cap=len(s0) + 2 * Dp + 8, device=DEV)
    sym, _ = h.rans_pop(st, per_chain, K)
    st.check()
    for b in range(B):
        assert np.array_equal(sym[b, :D].cpu().numpy(), g[f"{name}_pop_sym"])
        assert int(sym[b, D:].abs().sum()) == 0
    assert st.to_lists() == [words_to_state(g[f"{name}_state_after_pop"])] * B
    h.rans_push_table(st, cdf, sym, K)               # shared table, chain_stride 0
    st.check()
    assert st.to_lists() == [s0] * B
    h.rans_push_table(st, per_chain, padded(g[f"{name}_push_sym"]), K)
    st.check()
    assert st.to_lists() == [words_to_state(g[f"{name}_state_after_push"])] * B


def test_rans_push_fc_equals_push_table(golden):
    h = hip()
    g = golden("tables_rans.npz")
    cdf_np = g["zuni_cdf"]
    D, K = cdf_np.shape[0], cdf_np.shape[1] - 1
    sym = g["zuni_push_sym"]
    s0 = words_to_state(g["zuni_state0"])
    st = h.RansState.from_lists([s0], cap=len(s0) + D + 8, device=DEV)
    rows = np.arange(D)
    f = (cdf_np[rows, sym + 1] - cdf_np[rows, sym]).astype(np.uint32).view(np.int32)[None]
    c = cdf_np[rows, sym].astype(np.uint32).view(np.int32)[None]
    h.rans_push(st, dev(f), dev(c))
    st.check()
    o = O.Stack(s0)
    assert O.push(o, cdf_np, sym) == O.OK
    assert st.to_lists()[0] == o.tolist()


def test_push_division_corner_cases():
    """ANS.encode's `head // f, head % f` (mnist_compress.py:55) over frequencies the float-reciprocal
    estimate finds hardest: f = 1, f = 2^31 - K + 1 (one bin holds everything), powers of two and their
    neighbours, with heads spread over [2^32, 2^64).  Word streams must equal the oracle's bigint-free
    64-bit restatement, which is pinned to the reference's Python ints by the golden fixtures."""
    h = hip()
    rng = np.random.RandomState(42)
    K, D, B = 4, 6000, 4
    total = 1 << 31
    pats = [[1, 1, 1], [total - 3, 1, 1], [1, total - 3, 1], [1 << 30, (1 << 30) - 2, 1], [(1 << 30) - 1, (1 << 30) - 1, 1],
            [(1 << 16) + 1, (1 << 16) - 1, 12345], [3, 5, 7], [total // 3, total // 3, total // 3], [1 << 29, (3 << 29) - 2, 1]]
    rows = np.zeros((D, K + 1), dtype=np.int64)
    for d in range(D):
        f3 = pats[rng.randint(len(pats))]
        f = list(f3) + [total - sum(f3)]
        perm = rng.permutation(4)
        rows[d, 1:] = np.cumsum(np.array(f)[perm])
    assert np.all(np.diff(rows, axis=1) >= 1) and np.all(rows[:, -1] == total)
    sym = rng.randint(0, K, (B, D))
    sym[1] = np.argmax(np.diff(rows, axis=1), axis=1)          # always the huge bin: no words leave for a long time
    sym[2] = np.argmin(np.diff(rows, axis=1), axis=1)          # always the rarest bin: a word per symbol
    states = [reference_init_state(50, seed=3 + b) for b in range(B)]
    states[3][-1] = (1 << 64) - 12345                          # head near the top of its range
    st = h.RansState.from_lists(states, cap=50 + D + 8, device=DEV)
    r = np.arange(D)
    f = np.stack([(rows[r, sym[b] + 1] - rows[r, sym[b]]) for b in range(B)]).astype(np.uint32).view(np.int32)
    c = np.stack([rows[r, sym[b]] for b in range(B)]).astype(np.uint32).view(np.int32)
    h.rans_push(st, dev(f), dev(c))
    st.check()
    got = st.to_lists()
    for b in range(B):
        o = O.Stack(states[b])
        assert O.push(o, rows, sym[b]) == O.OK
        assert got[b] == o.tolist(), b
        # the reference's own arithmetic on Python ints (mnist_compress.py:49-56), verbatim
        x = list(states[b])
        for d in range(D):
            cs, fs = int(rows[d, sym[b, d]]), int(rows[d, sym[b, d] + 1] - rows[d, sym[b, d]])
            if x[-1] >= (((1 << 32) >> 31) << 32) * fs:   # lbound = 2^32: head >= 2^33 * f
                x.append(x[-1] >> 32)
                x[-2] = x[-2] & ((1 << 32) - 1)
            x[-1] = ((x[-1] // fs) << 31) + (x[-1] % fs) + cs
        assert got[b] == x, b
    # and back: popping the linear table returns the symbols and the initial states
    tab = np.zeros((D, h.aligned_ld(K)), dtype=np.uint32)
    tab[:, : K + 1] = rows
    back, _ = h.rans_pop(st, dev(tab.view(np.int32)), K, B=B)
    st.check()
    assert np.array_equal(back.cpu().numpy(), sym) and st.to_lists() == states


@pytest.mark.parametrize("bits", [16, 24, 28, 30])
def test_other_ans_precisions_vs_oracle(bits):
    """The ABI takes any precision 1..31 (the reference hard-codes 31, mnist_compress.py:76): tables, pop and
    push at 16/24 bits (generic division path) and 28/30 bits (systolic fast path) against the oracle, whose
    tables and rANS arithmetic follow ANS.__init__/encode/decode with `bits` as a parameter."""
    h = hip()
    rng = np.random.RandomState(bits)
    K, D, q = 256, 192, 8
    p = rng.dirichlet(np.full(K, 0.3), size=D)
    f_o, cdf_o, rc = O.tables(p, bits, q)
    assert rc == O.OK
    f, cdf, st_rows = h.table_rows(dev(p), bits, q, ld=h.aligned_ld(K))
    assert int(st_rows.abs().max()) == 0 and np.array_equal(u32(cdf)[:, : K + 1], cdf_o) and np.array_equal(u32(f), f_o)
    states = [reference_init_state(600, seed=bits + b) for b in range(3)]
    st = h.RansState.from_lists(states, cap=600 + D + 8, device=DEV)
    sym, _ = h.rans_pop(st, cdf, K, bits=bits, B=3)
    st.check()
    want = []
    for b in range(3):
        o = O.Stack(states[b])
        s_o, rc = O.pop(o, cdf_o, bits)
        assert rc == O.OK and np.array_equal(sym[b].cpu().numpy(), s_o)
        want.append(o.tolist())
    assert st.to_lists() == want
    data = rng.randint(0, K, (3, D)).astype(np.int32)
    h.rans_push_table(st, cdf, dev(data), K, bits=bits)
    st.check()
    for b in range(3):
        o = O.Stack(want[b])
        assert O.push(o, cdf_o, data[b], bits) == O.OK
        assert st.to_lists()[b] == o.tolist()
    # (f, c) flavour of the same push
    st2 = h.RansState.from_lists(want, cap=600 + D + 8, device=DEV)
    r = np.arange(D)
    fa = np.stack([cdf_o[r, data[b] + 1] - cdf_o[r, data[b]] for b in range(3)]).astype(np.uint32).view(np.int32)
    ca = np.stack([cdf_o[r, data[b]] for b in range(3)]).astype(np.uint32).view(np.int32)
    h.rans_push(st2, dev(fa), dev(ca), bits=bits)
    st2.check()
    assert st2.to_lists() == st.to_lists()


@pytest.mark.parametrize("bits", [16, 24, 28])
@pytest.mark.parametrize("layout", ["linear", "wave"])
def test_other_precisions_vs_reference_fixture(golden, bits, layout):
    """HIP tables / pop / push at 16, 24, 28 bits against word streams produced by the reference's ANS class;
    `wave`: the same rows in BS_LAYOUT_WAVE through k_rans_pop_wave."""
    h = hip()
    g = golden("rans_bits.npz")
    q, K = int(g["quantbits"]), g["pmf_f64"].shape[1]
    f, cdf, st_rows = h.table_rows(dev(g["pmf_f64"]), bits, q, ld=h.aligned_ld(K))
    assert int(st_rows.abs().max()) == 0
    assert np.array_equal(u32(f), g[f"b{bits}_f"]) and np.array_equal(u32(cdf)[:, : K + 1], g[f"b{bits}_cdf"])
    if layout == "wave":
        assert cdf.shape[0] % 64 == 0
        cdf = wave_table(h, g[f"b{bits}_cdf"], K, bits)
    s0 = words_to_state(g[f"b{bits}_state0"])
    st = h.RansState.from_lists([s0, s0], cap=len(s0) + 2 * cdf.shape[0] + 8, device=DEV)
    sym, _ = h.rans_pop(st, cdf, K, bits=bits, B=2)
    st.check()
    assert np.array_equal(sym[1].cpu().numpy(), g[f"b{bits}_pop_sym"])
    assert st.to_lists() == [words_to_state(g[f"b{bits}_state_after_pop"])] * 2
    h.rans_push_table(st, cdf, dev(np.tile(g[f"b{bits}_push_sym"], (2, 1))), K, bits=bits)
    st.check()
    assert st.to_lists() == [words_to_state(g[f"b{bits}_state_after_push"])] * 2


def test_status_codes():
    h = hip()
    K, D = 256, 300
    p = np.full((D, K), 1.0 / K)
    _, cdf, _ = h.table_rows(dev(p), 31, 8, ld=h.aligned_ld(K))
    # underflow: a head and two words cannot feed 300 8-bit symbols
    st = h.RansState.from_lists([[7, 9, 5 << 32], reference_init_state(400)], cap=1024, device=DEV)
    sym, _ = h.rans_pop(st, cdf, K, B=2)
    assert st.status.cpu().tolist() == [h.ST_UNDERFLOW, 0]
    with pytest.raises(h.BitswapHipError):
        st.check()
    # sticky: the failed chain is skipped, the healthy one keeps working
    before = st.to_lists()
    h.rans_push_table(st, cdf, sym, K)
    after = st.to_lists()
    assert after[0] == before[0] and after[1] == reference_init_state(400)
    # overflow: capacity too small for the pushed words
    st = h.RansState.from_lists([reference_init_state(50)], cap=60, device=DEV)
    s = torch.zeros((1, D), dtype=torch.int32, device=DEV)
    h.rans_push_table(st, cdf, s, K)
    assert st.status.cpu().tolist() == [h.ST_OVERFLOW]
    # bad symbol
    st = h.RansState.from_lists([reference_init_state(50)], cap=600, device=DEV)
    s[0, 3] = K
    h.rans_push_table(st, cdf, s, K)
    assert st.status.cpu().tolist() == [h.ST_BADSYMBOL]


def test_ans_class_drop_in(golden):
    """The reference call pattern, verbatim: ANS(pmfs, bits, quantbits).decode(state) / .encode(state, sym)."""
    from bitswap_amd.ans import ANS
    g = golden("tables_rans.npz")
    for name in ("ztop", "x"):
        pmfs = dev(g[f"{name}_pmf_f64"])
        q = int(g[f"{name}_quantbits"])
        a = ANS(pmfs, bits=31, quantbits=q)
        assert np.array_equal(a.pmfs, g[f"{name}_f"].astype(np.int64))
        assert np.array_equal(a.cdfs, g[f"{name}_cdf"].astype(np.int64))
        state = words_to_state(g[f"{name}_state0"])
        same = state
        state, sym = a.decode(state)
        assert state is same and sym.dtype == torch.int64 and sym.is_cuda
        assert np.array_equal(sym.cpu().numpy(), g[f"{name}_pop_sym"])
        assert state == words_to_state(g[f"{name}_state_after_pop"])
        state = ANS(pmfs, bits=31, quantbits=q).encode(state, sym)
        assert state == words_to_state(g[f"{name}_state0"])
        state = a.encode(state, [int(s) for s in g[f"{name}_push_sym"]])
        assert state == words_to_state(g[f"{name}_state_after_push"])
    a = ANS(dev(np.full((40, 16), 1 / 16)), 31, 4)
    with pytest.raises(IndexError):
        a.decode([3 << 32])


@pytest.mark.parametrize("chain", ["chain_mnist_small_bitswap", "chain_mnist_small_bbans", "chain_rgb4_small_bitswap",
                                   "chain_rgb4_small_bbans"])
@pytest.mark.parametrize("layout", ["linear", "wave"])
@pytest.mark.parametrize("spec", [1, 2, 3, 4])
def test_chain_replay_matches_reference_words(golden, chain, layout, spec):
    """Teacher-forced replay of the reference sender through the HIP kernels: same popped symbols,
    same per-operation state, same final word stream as the reference's Python run -- for the production kernel
    pair (layout wave: k_logistic<.., M_WAVE> + k_rans_pop_wave; spec 2: the uniform-bin CDF on every table that
    qualifies) as well as for the reference's linear rows / CDF spec 1."""
    from bitswap_amd.bins import uniform_step
    h = hip()
    g = golden(chain + ".npz")
    zend, xend, zcen = chain_tables(g)
    zend_d = [dev(z) for z in zend]
    xend_d = dev(xend[0]).unsqueeze(0).expand(xend.shape[0], -1)
    steps = {}
    for tab, e in list(enumerate(zend)) + [(-1, xend)]:
        hs = uniform_step(e) if (spec >= 2 and e.shape[1] + 1 >= 256) else None
        steps[tab] = None if hs is None else dev(hs)
    if spec >= 2:
        assert steps[-1] is not None and steps[0] is not None and steps[len(zend) - 1] is None
    lay = h.LAYOUT_WAVE if layout == "wave" else h.LAYOUT_LINEAR
    sp = lambda stp: None if stp is None else spec
    B = 2
    s0 = reference_init_state()
    st = h.RansState.from_lists([s0] * B, cap=40000, device=DEV)
    for i, (kind, tab, q) in enumerate(zip(g["op_kind"], g["op_table"], g["op_q"])):
        e = xend_d if tab < 0 else zend_d[tab]
        K = e.shape[1] + 1
        mu = dev(np.tile(g[f"op{i}_mu"], (B, 1)))
        sc = dev(np.tile(g[f"op{i}_scale"], (B, 1)))
        if kind == 0:
            cdf = h.logistic_tables(e, mu, sc, 31, int(q), layout=lay, step=steps[int(tab)], status=st.status, spec=sp(steps[int(tab)]))
            sym, z = h.rans_pop(st, cdf, K, centres=dev(zcen[tab]))
            assert np.array_equal(sym[1].cpu().numpy(), g[f"op{i}_sym"])
            want_z = zcen[tab][np.arange(zcen.shape[1]), g[f"op{i}_sym"]].astype(np.float32)
            assert np.array_equal(z[0].cpu().numpy(), want_z)
        else:
            sym = dev(np.tile(g[f"op{i}_sym"].astype(np.int32), (B, 1)))
            f, c = h.logistic_fc(e, mu, sc, sym, st.status, 31, int(q), step=steps[int(tab)], spec=sp(steps[int(tab)]))
            h.rans_push(st, f, c)
        assert (st.len.cpu() + 1).tolist() == [int(g["op_nwords"][i])] * B
        assert int(st.head[0].cpu().numpy().view(np.uint64)) == int(g["op_head"][i])
    st.check()
    assert st.to_lists() == [words_to_state(g["sent_words"])] * B


@pytest.mark.parametrize("data,sched", [("mnist", "bitswap"), ("mnist", "bbans"), ("cifar", "bitswap"), ("cifar", "bbans"),
                                        ("imagenet", "bitswap"), ("imagenet", "bbans")])
@pytest.mark.parametrize("spec", [1, 2, 3, 4])
def test_divergence_horizon_on_the_gpu(golden, sched, spec, data):
    """How far the HIP kernels follow the reference's OWN word stream: BASELINE configs[0] at its real width,
    one chain of 100 blocks written by the reference's sender on CPU (tests/golden/chain_mnist_full_*.npz; torch.sigmoid
    tables), replayed teacher-forced -- the reference's (mu, scale) and pushed symbols in -- through k_logistic +
    k_rans_pop_wave / the systolic push.  After every operation the head and word count must be the ORACLE's (same CDF spec,
    replayed on the CPU beside it): bit-exact HIP == oracle along 500 operations and ~50,000 words, also past the point
    where both have left the reference.  And the distance to the reference is the recorded one (tests/test_oracle.py::HORIZON):
    specs 1 and 2 keep the reference's state through all 100 blocks (spec 2's Bit-Swap stream with one word off by one),
    spec 3 leaves it at operation 165 = block 33.  Two chains side by side: both see the same thing.
    data = cifar: BASELINE configs[1], the headline's model at its real width -- 6 blocks = 102 operations of 2048 x 1024-bin rows
    (tests/golden/chain_cifar_full_*.npz): HIP == oracle word for word all the way, and the reference is left inside the first
    block by every spec (one table entry in 34 M: test_oracle.py::HORIZON_CIFAR)."""
    from bitswap_amd.bins import uniform_step
    from test_oracle import HORIZONS, full_chain_ops
    h = hip()
    g = golden(f"chain_{data}_full_{sched}.npz")
    zend, xend, zcen = chain_tables(g)
    zend_d = [dev(z) for z in zend]
    xend_d = dev(xend[0]).unsqueeze(0).expand(xend.shape[0], -1)
    steps_np = {tab: (uniform_step(e) if (spec >= 2 and e.shape[1] + 1 >= 256) else None)
                for tab, e in list(enumerate(zend)) + [(-1, xend)]}
    steps = {tab: None if v is None else dev(v) for tab, v in steps_np.items()}
    B = 2
    st = h.RansState.from_lists([reference_init_state()] * B, cap=80000, device=DEV)
    ost = O.Stack(reference_init_state(), cap=80000)
    first = None
    lens, heads = [], []
    for i, (kind, tab, q, mu, sc, sym) in enumerate(full_chain_ops(g)):
        e = xend_d if tab < 0 else zend_d[tab]
        K = e.shape[1] + 1
        mu_d, sc_d = dev(np.tile(mu, (B, 1))), dev(np.tile(sc, (B, 1)))
        sp = None if steps[tab] is None else spec
        mode = O.MODE_OF_SPEC[spec] if steps[tab] is not None else O.MODE_DET
        e_np = xend if tab < 0 else zend[tab]
        if kind == 0:
            cdf = h.logistic_tables(e, mu_d, sc_d, 31, q, layout=h.LAYOUT_WAVE, step=steps[tab], status=st.status, spec=sp)
            got, _ = h.rans_pop(st, cdf, K)
            osym, rc = O.layer_pop(ost, e_np, mu.astype(np.float64), sc.astype(np.float64), 31, q, mode, steps_np[tab])
        else:
            f, c = h.logistic_fc(e, mu_d, sc_d, dev(np.tile(sym, (B, 1))), st.status, 31, q, step=steps[tab], spec=sp)
            h.rans_push(st, f, c)
            rc = O.layer_push(ost, e_np, mu.astype(np.float64), sc.astype(np.float64), sym, 31, q, mode, steps_np[tab])
        assert rc == O.OK
        lens.append(st.len.clone())
        heads.append(st.head.clone())
        if first is None and (int(ost.len[0]) + 1 != int(g["op_nwords"][i]) or int(ost.head[0]) != int(g["op_head"][i])):
            first = i
        if i % 25 == 24 or i == len(g["op_kind"]) - 1:       # HIP == oracle after every operation (compared in batches)
            nw = torch.stack(lens).cpu().numpy()
            hd = torch.stack(heads).cpu().numpy().view(np.uint64)
            lens, heads = [], []
            assert (nw[-1] == int(ost.len[0])).all() and (hd[-1] == np.uint64(int(ost.head[0]))).all(), i
    st.check()
    got = st.to_lists()
    assert got == [ost.tolist()] * B                          # every word, not just the heads
    want_first, want_ndiff = HORIZONS[data][(sched, spec)]
    assert first == want_first
    if want_ndiff is not None:
        ref = words_to_state(g["sent_words"])
        assert len(got[0]) == len(ref) and sum(x != y for x, y in zip(got[0], ref)) == want_ndiff


@pytest.mark.parametrize("data,sched", [("mnist", "bitswap"), ("mnist", "bbans"), ("cifar", "bitswap"), ("cifar", "bbans"),
                                        ("imagenet", "bitswap"), ("imagenet", "bbans")])
@pytest.mark.parametrize("spec", [1, 2, 3, 4])
def test_bits_per_dim_of_the_full_width_reference_chain_on_the_gpu(golden, sched, spec, data):
    """bits/dim <= 1e-4 pinned on BASELINE configs[0] at full width, 100 blocks, per CDF spec.  The ideal code
    length of the reference's own 500 operations' symbols (tests/golden/chain_mnist_full_*.npz, written by the reference's
    sender: mnist_compress.py:176-251) under the frequencies the HIP table kernel builds from the teacher-forced (mu, scale)
    against the same under the reference's torch.sigmoid tables -- per operation and in total -- and the total against the
    fixture's `nets` (mnist_compress.py:253-261).  The frequencies must also be the oracle's, symbol for symbol.
    data = cifar: the same on BASELINE configs[1] at its real width (the bench headline's model), 6 blocks = 102 operations."""
    from bitswap_amd.bins import uniform_step
    from test_oracle import check_rate_against_reference, ideal_bits_of_full_chain, torch_table_bits
    h = hip()
    g = golden(f"chain_{data}_full_{sched}.npz")
    zend, xend, _ = chain_tables(g)
    zend_d = [dev(z) for z in zend]
    xend_d = dev(xend[0]).unsqueeze(0).expand(xend.shape[0], -1)
    steps_np = {tab: (uniform_step(e) if (spec >= 2 and e.shape[1] + 1 >= 256) else None)
                for tab, e in list(enumerate(zend)) + [(-1, xend)]}
    steps = {tab: None if v is None else dev(v) for tab, v in steps_np.items()}
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    nchecked = [0]

    def freqs(tab, q, mu, sc, sym):
        e = xend_d if tab < 0 else zend_d[tab]
        sp = None if steps[tab] is None else spec
        f, _ = h.logistic_fc(e, dev(mu[None]), dev(sc[None]), dev(sym[None]), status, 31, q, step=steps[tab], spec=sp)
        f = f.cpu().numpy().view(np.uint32)[0]
        if nchecked[0] < 40:                                 # HIP == oracle on the first blocks' operations (the rest: horizon test)
            mode = O.MODE_OF_SPEC[spec] if steps_np[tab] is not None else O.MODE_DET
            e_np = np.ascontiguousarray(xend if tab < 0 else zend[tab])
            fo, _, rc = O.tables(O.logistic_pmf(e_np, mu.astype(np.float64), sc.astype(np.float64), mode, steps_np[tab]), 31, q)
            assert rc == O.OK and np.array_equal(f, fo[np.arange(len(sym)), sym])
            nchecked[0] += 1
        return f
    got = ideal_bits_of_full_chain(g, freqs)
    assert int(status.item()) == 0
    check_rate_against_reference(g, got, torch_table_bits(g), f"HIP spec {spec} {data} {sched}")


@pytest.mark.parametrize("layout", ["linear", "wave"])
@pytest.mark.parametrize("spec", [1, 2, 3, 4])
def test_full_size_round_trip_property(layout, spec):
    """BASELINE config sizes (D=2048, K=1024 latents; D=3072, K=256 pixels), 64 chains: bits-back
    pop followed by push of the same symbols restores every state exactly; pushing then popping
    returns the pushed symbols.  Size-independent property, no oracle needed.  All four (layout, CDF spec)
    kernel combinations; the production pair is (wave, 2)."""
    from bitswap_amd.bins import uniform_step
    h = hip()
    rng = np.random.RandomState(9)
    lay = h.LAYOUT_WAVE if layout == "wave" else h.LAYOUT_LINEAR
    for (D, K, q) in ((2048, 1024, 10), (3072, 256, 8)):
        B = 64
        lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
        e_np = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
        e = dev(e_np)
        step = dev(uniform_step(e_np)) if spec >= 2 else None
        sp = None if step is None else spec
        mu = dev(rng.randn(B, D).astype(np.float32))
        sc = dev(rng.uniform(0.1, 1.0, (B, D)).astype(np.float32))
        states = [reference_init_state(3000, seed=100 + b) for b in range(B)]
        st = h.RansState.from_lists(states, cap=3000 + D + 64, device=DEV)
        cdf = h.logistic_tables(e, mu, sc, 31, q, layout=lay, step=step, status=st.status, spec=sp)
        sym, _ = h.rans_pop(st, cdf, K)
        f, c = h.logistic_fc(e, mu, sc, sym, st.status, 31, q, step=step, spec=sp)
        h.rans_push(st, f, c)
        st.check()
        assert st.to_lists() == states
        data = dev(rng.randint(0, K, (B, D)).astype(np.int32))
        f, c = h.logistic_fc(e, mu, sc, data, st.status, 31, q, step=step, spec=sp)
        h.rans_push(st, f, c)
        back, _ = h.rans_pop(st, cdf, K)
        st.check()
        assert torch.equal(back, data)
        assert st.to_lists() == states
        # every cdf row is a valid table
        rows = u32(cdf[:4])
        t = (unpermute_wave(rows, K)[0] if layout == "wave" else rows[:, :, :K]).astype(np.int64)
        assert np.all(t[:, :, 0] == 0) and np.all(t[:, :, -1] < 1 << 31) and np.all(np.diff(t, axis=2) >= 1)


def unpermute_wave(rows, K):
    """BS_LAYOUT_WAVE row [K+64] -> (c_0..c_{K-1}, pivots) following include/bitswap_hip.h."""
    j = np.arange(K)
    off = ((j // 256) * 64 + j % 64) * 4 + (j // 64) % 4
    return rows[..., off], rows[..., K:K + 64]


@pytest.mark.parametrize("K", [256, 512, 1024, 2048])
def test_wave_layout_tables_and_pop(K):
    """The wave-native hand-off format: same integers as the linear rows, permuted; popping from it
    gives the same symbols and the same state words as popping from linear rows."""
    h = hip()
    q = int(np.log2(K))
    rng = np.random.RandomState(K + 1)
    D, B = 128, 5
    lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
    e = dev(np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)]))
    mu = dev(rng.randn(B, D).astype(np.float32))
    sc = dev(rng.uniform(0.1, 1, (B, D)).astype(np.float32))
    lin = h.logistic_tables(e, mu, sc, 31, q)
    wav = h.logistic_tables(e, mu, sc, 31, q, layout=h.LAYOUT_WAVE)
    assert wav.shape[-1] == K + 64 and wav.bs_layout == h.LAYOUT_WAVE
    c, piv = unpermute_wave(u32(wav), K)
    assert np.array_equal(c, u32(lin)[:, :, :K])
    nr = K // 64
    assert np.array_equal(piv[..., :nr], u32(lin)[:, :, 0:K:64])          # word r = c_{64r}
    assert np.all(piv[..., nr] == 1 << 31) and np.all(piv[..., nr + 1:] == 0xffffffff)
    states = [reference_init_state(4000, seed=7 + b) for b in range(B)]
    s1 = h.RansState.from_lists(states, cap=6000, device=DEV)
    s2 = h.RansState.from_lists(states, cap=6000, device=DEV)
    cen = dev(rng.randn(D, K))
    sym1, z1 = h.rans_pop(s1, lin, K, centres=cen)
    sym2, z2 = h.rans_pop(s2, wav, K, centres=cen)
    s1.check(); s2.check()
    assert torch.equal(sym1, sym2) and torch.equal(z1, z2)
    assert s1.to_lists() == s2.to_lists()
    # push back through the wave table (shared-table form with chain 0's rows, and per-chain form)
    h.rans_push_table(s2, wav, sym2, K)
    s2.check()
    assert s2.to_lists() == states
    one = wav[0]
    one.bs_layout = h.LAYOUT_WAVE
    s3 = h.RansState.from_lists([states[0]] * 2, cap=6000, device=DEV)
    sym3, _ = h.rans_pop(s3, one, K)
    assert torch.equal(sym3[0], sym3[1])
    h.rans_push_table(s3, one, sym3, K)
    s3.check()
    assert s3.to_lists() == [states[0]] * 2


@pytest.mark.parametrize("K", [256, 512, 1024, 2048])
@pytest.mark.parametrize("ptype", [torch.float32, torch.float64])
@pytest.mark.parametrize("spec", [2, 3, 4])
def test_logistic_spec2_bit_exact_vs_oracle(K, ptype, spec):
    """CDF specs 2 and 3 (uniform bins) on the GPU -- decode flavour in both layouts and encode flavour -- against the
    oracle's C restatements (oracle/bitswap_oracle.c::det2_row_cdf / det3_row_cdf), bit for bit; spec 3 on rows that take
    the batch inversion AND on peaked rows that fall back to spec 2's arithmetic ((K/64) h / scale >= 8); 6 chains so that a wavefront
    walks several chains with the same residual registers, scales down to the model's minimum and saturated rows."""
    from bitswap_amd.bins import uniform_step
    h = hip()
    q = int(np.log2(K))
    rng = np.random.RandomState(K + 7)
    D, B = 41, 6
    lo = rng.uniform(-9, -2, D).astype(np.float16).astype(np.float64)
    hi = rng.uniform(2, 9, D).astype(np.float16).astype(np.float64)
    e = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
    step = uniform_step(e)
    assert step is not None
    mu = (rng.randn(B, D) * 0.8).astype(np.float32)
    sc = rng.uniform(0.1, 1.0, (B, D)).astype(np.float32)
    sc[0, :8] = 0.1
    mu[1, 0], mu[1, 1] = 30.0, -30.0
    sc[2, :6] = (np.float32(K // 64) * step[:6] / np.array([7.0, 7.9, 8.1, 9.0, 40.0, 600.0])).astype(np.float32)   # either side of spec 3's switch
    mu[3, :6], sc[3, :6] = 12.0, 0.1                                        # lower groups beyond the anchor clamp at 41
    sym = rng.randint(0, K, (B, D)).astype(np.int32)
    sym[0, :4] = [0, K - 1, 1, K - 2]
    status = torch.zeros(B, dtype=torch.int32, device=DEV)
    lin = u32(h.logistic_tables(dev(e), dev(mu, ptype), dev(sc, ptype), 31, q, step=dev(step), status=status, spec=spec))
    wav = u32(h.logistic_tables(dev(e), dev(mu, ptype), dev(sc, ptype), 31, q, layout=h.LAYOUT_WAVE, step=dev(step),
                                status=status, spec=spec))
    f, c = h.logistic_fc(dev(e), dev(mu, ptype), dev(sc, ptype), dev(sym), status, 31, q, step=dev(step), spec=spec)
    f, c = u32(f), u32(c)
    assert int(status.abs().max()) == 0
    cw, piv = unpermute_wave(wav, K)
    rows = np.arange(D)
    for b in range(B):
        pmf = O.logistic_pmf(e, mu[b].astype(np.float64), sc[b].astype(np.float64), O.MODE_OF_SPEC[spec], step)
        _, want, rc = O.tables(pmf, 31, q)
        assert rc == O.OK
        assert np.array_equal(lin[b][:, : K + 1], want), b
        assert np.array_equal(cw[b], want[:, :K]), b
        assert np.array_equal(piv[b][:, : K // 64], want[:, 0:K:64])
        assert np.array_equal(c[b], want[rows, sym[b]])
        assert np.array_equal(f[b], want[rows, sym[b] + 1] - want[rows, sym[b]])
    # K below 256 has no spec 2
    e64 = np.stack([np.linspace(-4, 4, 65)[1:-1]] * 3)
    with pytest.raises(h.BitswapHipError):
        h.logistic_tables(dev(e64), dev(mu[:, :3]), dev(sc[:, :3]), 31, 6, step=dev(uniform_step(e64)), spec=spec)


@pytest.mark.parametrize("spec", [1, 2, 3, 4])
def test_degenerate_parameters_are_flagged(spec):
    """NaN / Inf / non-positive (mu, scale) from a broken checkpoint: the table kernels set BS_ST_BADTABLE for the
    chain (the reference would trip over its assert at mnist_compress.py:47 or code garbage), later kernels skip it,
    the healthy chains are untouched."""
    from bitswap_amd.bins import uniform_step
    h = hip()
    K, D, B = 256, 64, 5
    e = np.stack([np.linspace(-4, 4, K + 1)[1:-1]] * D)
    step = dev(uniform_step(e)) if spec >= 2 else None
    sp = None if step is None else spec
    mu = np.zeros((B, D), dtype=np.float32)
    sc = np.full((B, D), 0.5, dtype=np.float32)
    mu[1, 3] = np.nan
    sc[2, 60] = 0.0
    sc[3, 7] = np.inf
    states = [reference_init_state(800, seed=b) for b in range(B)]
    st = h.RansState.from_lists(states, cap=2000, device=DEV)
    cdf = h.logistic_tables(dev(e), dev(mu), dev(sc), 31, 8, layout=h.LAYOUT_WAVE, step=step, status=st.status, spec=sp)
    assert st.status.cpu().tolist() == [0, h.ST_BADTABLE, h.ST_BADTABLE, h.ST_BADTABLE, 0]
    sym, _ = h.rans_pop(st, cdf, K)
    got = st.to_lists()
    assert got[1:4] == states[1:4] and got[0] != states[0] and got[4] != states[4]
    st2 = h.RansState.from_lists(states, cap=2000, device=DEV)
    h.logistic_fc(dev(e), dev(mu), dev(sc), sym, st2.status, 31, 8, step=step, spec=sp)
    assert st2.status.cpu().tolist() == [0, h.ST_BADTABLE, h.ST_BADTABLE, h.ST_BADTABLE, 0]


@pytest.mark.parametrize("K,D,ptype,shared_row", [(1024, 192, np.float32, False), (256, 128, np.float32, True),
                                                   (512, 64, np.float64, False), (2048, 64, np.float32, False),
                                                   (1024, 2048, np.float32, False),
                                                   (256, 15872, np.float32, True)])     # D = BS_POP_PIVOT_MAX_D: the LDS limit
@pytest.mark.parametrize("spec", [2, 3, 4])
def test_pivot_handoff_pops_the_same_symbols_as_whole_rows(K, D, ptype, shared_row, spec):
    """BS_LAYOUT_PIVOT (64 cumulative values per row; bs_rans_pop_pivot rebuilds the symbol's group of bins with the table
    kernel's arithmetic) against BS_LAYOUT_WAVE (the whole integer row in HBM) and against the oracle: same symbols, same
    words, same centres -- peaked and flat rows, remnant bumps inside and outside the popped group, the first and the last
    bin, float32 and float64 parameters, the pixel form (one endpoint row shared by every dim), and the bench's own row
    count; a chain flagged by the table kernel is skipped by both."""
    from bitswap_amd.bins import uniform_step
    h = hip()
    assert h.pivot_supported(K, D) and not h.pivot_supported(K, h.PIVOT_MAX_D + 64)
    q = int(np.log2(K))
    rng = np.random.RandomState(K + D)
    B = 6
    if shared_row:
        e = np.stack([np.linspace(-1.0, 1.0, K + 1)[1:-1]] * D)
        cen = np.stack([np.linspace(-1.0, 1.0, K)] * D)
    else:
        lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
        e = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
        cen = np.stack([np.linspace(a, b, K) for a, b in zip(lo, hi)])
    step = dev(uniform_step(e))
    mu = (rng.randn(B, D) * (0.4 if shared_row else 1.5)).astype(ptype)
    sc = rng.uniform(0.004 if shared_row else 0.05, 1.0, (B, D)).astype(ptype)
    sc[0, :] = 0.004 if shared_row else 0.02            # a chain of peaked rows: most bins f = 1
    mu[1, :] = -50.0                                    # all mass in the first bin
    mu[2, :] = 50.0                                     # ... in the last one
    hh = uniform_step(e)
    sc[3, :8] = ((K // 64) * hh[:8] / np.array([7.0, 7.9, 8.1, 9.0, 20.0, 40.0, 100.0, 300.0])).astype(ptype)   # either side of spec 3's switch to spec 2's arithmetic
    e_d = dev(e[:1]).expand(D, -1) if shared_row else dev(e)
    c_d = dev(cen)
    states = [reference_init_state(3000 + 12 * D // 10, seed=b) for b in range(B)]
    res = {}
    for layout in (h.LAYOUT_WAVE, h.LAYOUT_PIVOT):
        st = h.RansState.from_lists(states, cap=8000 + 2 * D, device=DEV)
        tab = h.logistic_tables(e_d, dev(mu), dev(sc), 31, q, layout=layout, step=step, status=st.status, spec=spec)
        assert tab.shape[-1] == (h.PIVOT_LD if layout == h.LAYOUT_PIVOT else K + 64)
        sym, z = h.rans_pop(st, tab, K, 31, centres=c_d)
        st.check()
        res[layout] = (sym.cpu().numpy(), z.cpu().numpy(), st.to_lists())
    a, b = res[h.LAYOUT_WAVE], res[h.LAYOUT_PIVOT]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert (a[0][1] == 0).all() and (a[0][2] >= K - 2).mean() > 0.9      # the mass sits where the mean is
    # ... and the oracle (the same CDF spec) agrees
    for bb in (0, 2, 3):
        ost = O.Stack(states[bb], cap=8000 + 2 * D)
        osym, rc = O.layer_pop(ost, e, mu[bb].astype(np.float64), sc[bb].astype(np.float64), 31, q, O.MODE_OF_SPEC[spec])
        assert rc == O.OK and np.array_equal(osym, b[0][bb]) and ost.tolist() == b[2][bb]
    # a degenerate chain: flagged by the table kernel, skipped by the pop, the others untouched by it
    sc2 = sc.copy()
    sc2[4, 7] = 0.0
    st = h.RansState.from_lists(states, cap=8000 + 2 * D, device=DEV)
    tab = h.logistic_tables(e_d, dev(mu), dev(sc2), 31, q, layout=h.LAYOUT_PIVOT, step=step, status=st.status, spec=spec)
    sym, _ = h.rans_pop(st, tab, K, 31)
    assert st.status.cpu().tolist() == [0, 0, 0, 0, h.ST_BADTABLE, 0] and st.to_lists()[4] == states[4]
    assert np.array_equal(sym.cpu().numpy()[[0, 1, 2, 3, 5]], b[0][[0, 1, 2, 3, 5]])


@pytest.mark.parametrize("spec", [2, 3, 4])
def test_cdf_spec2_domain_is_flagged(spec):
    """a row whose anchors leave the +-700 domain of det_exp (scale tiny against the bin width -- reachable
    through the C ABI, never by the reference's models) is outside CDF spec 2: every flavour flags BS_ST_BADTABLE for the
    chain and leaves its state alone, as the oracle does (tests/test_oracle.py::test_cdf_spec2_domain_is_enforced); the
    same parameters are fine under spec 1 (one clamped sigmoid per endpoint: monotone)."""
    from bitswap_amd.bins import uniform_step
    from bitswap_amd.codec import Hip64Backend
    h = hip()
    K, D, B = 256, 64, 3
    e = np.stack([np.linspace(-4, 4, K + 1)[1:-1]] * D)
    step = dev(uniform_step(e))
    mu = np.zeros((B, D), dtype=np.float32)
    sc = np.full((B, D), 0.5, dtype=np.float32)
    sc[1, 5] = 1e-4
    states = [reference_init_state(800, seed=b) for b in range(B)]
    st = h.RansState.from_lists(states, cap=2000, device=DEV)
    cdf = h.logistic_tables(dev(e), dev(mu), dev(sc), 31, 8, layout=h.LAYOUT_WAVE, step=step, status=st.status, spec=spec)
    assert st.status.cpu().tolist() == [0, h.ST_BADTABLE, 0]
    sym, _ = h.rans_pop(st, cdf, K)
    got = st.to_lists()
    assert got[1] == states[1] and got[0] != states[0] and got[2] != states[2]
    st2 = h.RansState.from_lists(states, cap=2000, device=DEV)
    h.logistic_fc(dev(e), dev(mu), dev(sc), sym, st2.status, 31, 8, step=step, spec=spec)
    assert st2.status.cpu().tolist() == [0, h.ST_BADTABLE, 0]
    st3 = h.RansState.from_lists(states, cap=2000, device=DEV)                       # spec 1: a well-formed table
    cdf1 = h.logistic_tables(dev(e), dev(mu), dev(sc), 31, 8, layout=h.LAYOUT_WAVE, status=st3.status)
    assert st3.status.cpu().tolist() == [0, 0, 0]
    s1, _ = h.rans_pop(st3, cdf1, K)
    st3.check()
    assert 0 <= int(s1.min()) and int(s1.max()) < K
    s64 = h.RansState64.from_lists(states, cap=64, device=DEV)                      # the fused 64-state kernels too
    h.layer_pop64(s64, dev(e), dev(mu), dev(sc), 31, 8, step=step, spec=spec)
    assert s64.status.cpu().tolist() == [0, h.ST_BADTABLE, 0]


@pytest.mark.parametrize("K", [256, 512, 1024])
@pytest.mark.parametrize("spec", [1, 2, 3, 4])
def test_layer64_kernels_vs_oracle(K, spec):
    """bs_layer_pop64 / bs_layer_push64 (64 states per chain, table row + rANS step in one launch) against the
    oracle: state j of a chain codes dims j, j + 64, ... with the single-state arithmetic (ANS.decode / ANS.encode,
    mnist_compress.py:49-68).  D not a multiple of 64 (ragged residues), 7 chains (partial chain group), float32 and
    float64 parameters; popped symbols, all 64 x B states, centres, the shared-row (prior) form, and pop-after-push."""
    from bitswap_amd.bins import uniform_step
    from oracle.backend import split_state
    h = hip()
    q = int(np.log2(K))
    rng = np.random.RandomState(K + spec)
    D, B = 200, 7
    lo, hi = rng.uniform(-8, -2, D), rng.uniform(2, 8, D)
    e = np.stack([np.linspace(a, b, K + 1)[1:-1] for a, b in zip(lo, hi)])
    step_np = uniform_step(e) if spec >= 2 else None
    step = None if step_np is None else dev(step_np)
    sp = None if step is None else spec
    mode = O.MODE_OF_SPEC[spec]
    mu = (rng.randn(B, D) * 0.6).astype(np.float32)
    sc = rng.uniform(0.1, 1.0, (B, D)).astype(np.float32)
    cen = rng.randn(D, K)
    states = [reference_init_state(1600, seed=31 + b) for b in range(B)]

    def oracle_pop(stacks, mu_b, sc_b):
        sym = np.zeros(D, dtype=np.int32)
        for j in range(64):
            sl = slice(j, None, 64)
            s, rc = O.layer_pop(stacks[j], e[sl], mu_b[sl].astype(np.float64), sc_b[sl].astype(np.float64), 31, q, mode,
                                None if step_np is None else step_np[sl])
            assert rc == O.OK
            sym[sl] = s
        return sym

    for ptype in (torch.float32, torch.float64):
        st = h.RansState64.from_lists(states, cap=400, device=DEV)
        assert st.to_lists() == [split_state(s) for s in states]
        sym, z = h.layer_pop64(st, dev(e), dev(mu, ptype), dev(sc, ptype), 31, q, centres=dev(cen), step=step, spec=sp)
        st.check()
        got = st.to_lists()
        for b in range(B):
            stacks = [O.Stack(sub) for sub in split_state(states[b])]
            want = oracle_pop(stacks, mu[b], sc[b])
            assert np.array_equal(sym[b].cpu().numpy(), want), b
            assert got[b] == [s_.tolist() for s_ in stacks], b
            assert np.array_equal(z[b].cpu().numpy(), cen[np.arange(D), want].astype(np.float32))
        h.layer_push64(st, dev(e), dev(mu, ptype), dev(sc, ptype), sym, 31, q, step=step, spec=sp)   # bits back: restores every state
        st.check()
        assert st.to_lists() == [split_state(s) for s in states]
    # fresh symbols: push, compare with the oracle, pop them back
    data = rng.randint(0, K, (B, D)).astype(np.int32)
    st = h.RansState64.from_lists(states, cap=400, device=DEV)
    h.layer_push64(st, dev(e), dev(mu), dev(sc), dev(data), 31, q, step=step, spec=sp)
    st.check()
    got = st.to_lists()
    for b in range(B):
        stacks = [O.Stack(sub) for sub in split_state(states[b])]
        for j in range(64):
            sl = slice(j, None, 64)
            assert O.layer_push(stacks[j], e[sl], mu[b][sl].astype(np.float64), sc[b][sl].astype(np.float64),
                                np.ascontiguousarray(data[b][sl]), 31, q, mode, None if step_np is None else step_np[sl]) == O.OK
        assert got[b] == [s_.tolist() for s_ in stacks], b
    back, _ = h.layer_pop64(st, dev(e), dev(mu), dev(sc), 31, q, step=step, spec=sp)
    st.check()
    assert torch.equal(back.cpu(), torch.from_numpy(data)) and st.to_lists() == [split_state(s) for s in states]
    # one row set shared by all chains (the prior): same as expanding it
    st_a = h.RansState64.from_lists(states, cap=400, device=DEV)
    st_b = h.RansState64.from_lists(states, cap=400, device=DEV)
    sa, _ = h.layer_pop64(st_a, dev(e), dev(mu[0]), dev(sc[0]), 31, q, step=step, spec=sp)
    sb, _ = h.layer_pop64(st_b, dev(e), dev(np.tile(mu[:1], (B, 1))), dev(np.tile(sc[:1], (B, 1))), 31, q, step=step, spec=sp)
    assert torch.equal(sa, sb) and st_a.to_lists() == st_b.to_lists()


def test_layer64_status_codes():
    h = hip()
    K, D, B = 256, 128, 3
    e = np.stack([np.linspace(-4, 4, K + 1)[1:-1]] * D)
    mu, sc = np.zeros((B, D), dtype=np.float32), np.full((B, D), 0.5, dtype=np.float32)
    # chain 0: two words per state cannot feed two 8-bit symbols per state forever -> underflow after a few pops
    states = [reference_init_state(200, seed=1), reference_init_state(3000, seed=2), reference_init_state(3000, seed=3)]
    st = h.RansState64.from_lists(states, cap=100, device=DEV)
    for _ in range(8):
        sym, _ = h.layer_pop64(st, dev(e), dev(mu), dev(sc), 31, 8)
    assert st.status.cpu().tolist() == [h.ST_UNDERFLOW, 0, 0]
    with pytest.raises(h.BitswapHipError):
        st.check()
    # bad symbol / degenerate parameters / overflow are reported per chain, healthy chains keep going
    st = h.RansState64.from_lists(states[1:], cap=60, device=DEV)
    bad = np.zeros((2, D), dtype=np.int32)
    bad[0, 77] = K
    h.layer_push64(st, dev(e), dev(mu[:2]), dev(sc[:2]), dev(bad), 31, 8)
    assert st.status.cpu().tolist() == [h.ST_BADSYMBOL, 0]
    st.status.zero_()
    sc2 = sc[:2].copy()
    sc2[1, 5] = np.nan
    h.layer_push64(st, dev(e), dev(mu[:2]), dev(sc2), dev(np.zeros((2, D), dtype=np.int32)), 31, 8)
    assert st.status.cpu().tolist() == [0, h.ST_BADTABLE]
    st.status.zero_()
    rare = np.full((2, D), K - 1, dtype=np.int32)           # far tail: ~30 bits per symbol, a word per push
    for _ in range(40):
        h.layer_push64(st, dev(e), dev(mu[:2]), dev(sc[:2]), dev(rare), 31, 8)
    assert st.status.cpu().tolist() == [h.ST_OVERFLOW, h.ST_OVERFLOW]
