"""GPU: the batched codec end to end on the HIP kernels (through the C ABI)."""
import numpy as np
import pytest
import torch

import oracle as O
from oracle.backend import OracleBackend
from bitswap_amd import cli, container, tiling, workload
from bitswap_amd.codec import BitSwapCodec, initial_states
from conftest import chain_tables, load_golden_model, reference_init_state, words_to_state

pytestmark = pytest.mark.gpu
DEV = "cuda"


def record_nets(codec):
    rec, orig = [], codec._net

    def wrapped(fn, given):
        out = orig(fn, given)
        rec.append(out)
        return out
    codec._net = wrapped
    return rec, orig


@pytest.mark.parametrize("bitswap", [1, 0])
@pytest.mark.parametrize("name,q", [("mnist2", 10), ("cifar8", 8)])
def test_round_trip_and_oracle_word_parity(name, q, bitswap):
    """Sender on the GPU, then (a) the oracle replays the same schedule on the CPU with the GPU's conv
    outputs: the word streams must be identical; (b) the GPU receiver returns the images and unwinds
    every chain to its initial state."""
    model, zend, zcen = workload.build(name, DEV, quantbits=q, small=16)
    B, n = 5, 2
    images = workload.synthetic_blocks(B * n, model.xs, seed=3).view(B, n, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=q, bitswap=bool(bitswap))
    rec, plain_net = record_nets(codec)
    state, met = codec.compress(images.to(DEV))
    sent = state.to_lists()
    assert np.all(met["total"][:, -1] > 0)

    it = iter(rec)
    oc = BitSwapCodec(model, zend.cpu(), zcen.cpu(), quantbits=q, bitswap=bool(bitswap),
                      backend=OracleBackend(O.MODE_DET, threads=4))
    oc._net = lambda fn, given: tuple(t.cpu() for t in next(it))
    ostate, omet = oc.compress(images)
    assert ostate.to_lists() == sent
    assert np.array_equal(omet["cma"], met["cma"]) and np.array_equal(omet["rest_len"], met["rest_len"])

    codec._net = plain_net
    out = codec.decompress(state, n)
    assert torch.equal(out.cpu(), images)
    assert state.to_lists() == initial_states(B)


def test_sender_is_deterministic_and_chains_independent_with_nn_batch():
    model, zend, zcen = workload.build("mnist2", DEV, quantbits=10, small=16, nn_batch=4)
    B, n = 6, 2
    images = workload.synthetic_blocks(B * n, model.xs, seed=5).view(B, n, -1).to(torch.int32).to(DEV)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10)
    a, _ = codec.compress(images)
    b, _ = codec.compress(images)
    assert a.to_lists() == b.to_lists()
    # a different batch composition (chains 2..4 alone) gives those chains the same streams
    sub, _ = codec.compress(images[2:5], state=codec.new_states(3, n, states=initial_states(B)[2:5]))
    assert sub.to_lists() == a.to_lists()[2:5]


def test_full_width_imagenet_block_step():
    """One lock-step block of the real ImageNet32 nz=4 architecture (reswidth 254, Z=2048, X=3072,
    K=1024): lossless, state restored, bits accounted."""
    model, zend, zcen = workload.build("imagenet4", DEV, quantbits=10)
    B = 8
    images = workload.synthetic_blocks(B, model.xs, seed=9).view(B, 1, -1).to(torch.int32).to(DEV)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10)
    state, met = codec.compress(images)
    assert np.all(met["cma"] > 0)
    out = codec.decompress(state, 1)
    assert torch.equal(out, images) and state.to_lists() == initial_states(B)


def test_bbans_needs_deep_initial_stack():
    """config 5: BB-ANS pops all nz layers before pushing anything; too few initial bits is reported
    per chain (the reference would die with IndexError at mnist_compress.py:66)."""
    from bitswap_amd import hip
    model, zend, zcen = workload.build("cifar8", DEV, quantbits=10, small=16)
    images = workload.synthetic_blocks(2, model.xs, seed=1).view(2, 1, -1).to(torch.int32).to(DEV)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=False)
    with pytest.raises(hip.BitswapHipError, match="underflow"):
        codec.compress(images, nwords=3000)          # 8 x 2048 x ~10 bits > 3000 words
    state, _ = codec.compress(images, nwords=10000)
    out = codec.decompress(state, 1)
    assert torch.equal(out, images)


def test_cli_and_demo_container_on_gpu(tmp_path):
    r = cli.compress(10, 2, 1, 0, dataset="mnist", experiments=4, ndatapoints=2, decompress=True,
                     outdir=str(tmp_path), small=16, verbose=False)
    assert r["cmas"].shape == (4, 2) and np.all(r["total"] > 0)
    setup = cli.crop_setup(0, nz=2, quantbits=10, small=16)
    rng = np.random.RandomState(1)
    blocks, h, w = tiling.extract_blocks(rng.randint(0, 256, (70, 100, 3)).astype(np.uint8))
    (st, min_words, bpd), = cli.compress_images([blocks], quantbits=10, nz=2, setup=setup)
    arr = container.pack(st, min_words, len(blocks), h, w)
    st2, nb, hh, ww = container.unpack(arr)
    out, rest = cli.decompress_image(st2, nb, quantbits=10, nz=2, setup=setup)
    assert np.array_equal(tiling.unextract_blocks(out, hh, ww), tiling.unextract_blocks(blocks, h, w))
    assert rest == reference_init_state()[min_words:]


@pytest.mark.parametrize("ks,small", [(3, False), (5, False), (5, True)])
def test_winograd_convs_match_torch(ks, small):
    """The transform-domain convolution (k_wino_in -> one batched GEMM of 36 products -> k_wino_out) against
    F.conv2d in float64: F(4x4,3x3), F(4x4,5x5) (8x8 tiles, points +-1/2 added) and F(2x2,5x5); fp32 error a few 1e-6 of the
    output range; input-side bias+ELU and output-side bias+residual+ELU fused into the transforms."""
    from bitswap_amd import hip, winograd
    g = torch.Generator().manual_seed(ks)
    n, C = 5, 24
    x = torch.randn((n, C, 16, 16), generator=g).to(DEV)
    w = (torch.randn((C, C, ks, ks), generator=g) / (C * ks * ks) ** 0.5).to(DEV)
    b_in, b_out = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    res = torch.randn((n, C, 16, 16), generator=g).to(DEV)
    ms = winograd.tile_config(ks, small)
    U = winograd.transform_weights(w, ms)
    a = torch.nn.functional.elu(x.double() + b_in.double().view(1, -1, 1, 1))
    want = torch.nn.functional.conv2d(a, w.double(), padding=ks // 2) + b_out.double().view(1, -1, 1, 1) + res.double()
    m = torch.bmm(U, hip.wino_in(x, b_in, True, ms))
    s, act = hip.wino_out(m, tuple(x.shape), b_out, res, True, True, ms)
    scale = float(want.abs().max())
    assert float((s.double() - want).abs().max()) < 3e-5 * scale
    assert float((act.double() - torch.nn.functional.elu(want)).abs().max()) < 3e-5 * scale
    if not small:   # the fused pass (k_wino_fused) must give the same bits as the separate transforms
        ts = ms[0]
        _, _, v = hip.wino_fused(x, tuple(x.shape), 0, b_in, None, True, ts_out=ts)
        assert torch.equal(v, hip.wino_in(x, b_in, True, ms))
        s2, a2, v2 = hip.wino_fused(m, tuple(x.shape), ts, b_out, res, True, want_sum=True, want_act=True, ts_out=ts)
        assert torch.equal(s2, s) and torch.equal(a2, act)
        assert torch.equal(v2, hip.wino_in(act, None, False, ms))
        n7 = x[:3].contiguous()   # a partial image block (3 of 16 images of the workgroup live)
        assert torch.equal(hip.wino_fused(n7, tuple(n7.shape), 0, None, None, False, ts_out=ts)[2],
                           hip.wino_in(n7, None, False, ms))
    # plain conv (no bias, no activation, no residual) and bitwise repeatability
    m2 = torch.bmm(U, hip.wino_in(x, None, False, ms))
    y, _ = hip.wino_out(m2, tuple(x.shape), None, None, True, False, ms)
    y2, _ = hip.wino_out(torch.bmm(U, hip.wino_in(x, None, False, ms)), tuple(x.shape), None, None, True, False, ms)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=ks // 2)
    assert float((y.double() - ref).abs().max()) < 3e-5 * float(ref.abs().max()) and torch.equal(y, y2)


@pytest.mark.parametrize("algo", ["winograd", "gemm5", "miopen"])
@pytest.mark.parametrize("name", ["mnist2", "imagenetcrop4"])
def test_fused_epilogues_match_torch_modules(name, algo):
    """Model.fuse(): one epilogue launch per conv (net_epilogue.hip) against the plain torch modules
    (bias / ELU / residual / scale heads as separate launches).  Same math, float32: agreement to a few
    ulp of the activations, and the fused path is bitwise repeatable."""
    from bitswap_amd import hip
    model, _, _ = workload.build(name, DEV, quantbits=8, small=24)
    assert model.fused
    model.compress(True)
    model.conv_algo, model.gemm_min_batch = algo, 1       # ResNet convs: Winograd-domain GEMM / row GEMMs / MIOpen
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for i in range(model.nz):
            x = torch.randint(0, 256, (7, model.xdim), generator=g).float().to(DEV) if i == 0 else \
                torch.randn((7, model.zdim_flat), generator=g).to(DEV)
            z = torch.randn((7, model.zdim_flat), generator=g).to(DEV)
            for fn, inp in ((model.infer(i), (x - 127.5) / 127.5 if i == 0 else x), (model.generate(i), z)):
                model.fused = True
                mu_f, sc_f = fn(inp)
                mu_f2, sc_f2 = fn(inp)
                model.fused = False
                mu_t, sc_t = fn(inp)
                model.fused = True
                assert torch.equal(mu_f, mu_f2) and torch.equal(sc_f, sc_f2)
                assert mu_f.shape == mu_t.shape and sc_f.shape == sc_t.shape
                assert torch.allclose(mu_f, mu_t, rtol=2e-4, atol=2e-5), float((mu_f - mu_t).abs().max())
                assert torch.allclose(sc_f, sc_t, rtol=2e-4, atol=2e-6), float((sc_f - sc_t).abs().max())
    # the pointwise kernels alone, odd plane size (scalar path) and 16-byte path
    for shape in ((3, 5, 7, 9), (2, 6, 16, 16)):
        x = torch.randn(shape, generator=g).to(DEV)
        b = torch.randn(shape[1], generator=g).to(DEV)
        r = torch.randn(shape, generator=g).to(DEV)
        s, a = hip.bias_residual_elu(x.clone(), b, r, want_sum=True, want_act=True)
        want = x + b.view(1, -1, 1, 1) + r
        assert torch.allclose(s, want, atol=1e-6) and torch.allclose(a, torch.nn.functional.elu(want), atol=1e-6)
        _, a = hip.bias_residual_elu(x.clone(), None, None)
        assert torch.allclose(a, torch.nn.functional.elu(x), atol=1e-6)


@pytest.mark.parametrize("bitswap", [1, 0])
def test_grouped_codec_equals_plain_codec(bitswap):
    """GroupedCodec (chain groups on separate HIP streams, enqueue order interleaved) is a scheduling
    device only: every chain's stream equals the one the plain single-stream codec produces for the same
    group composition, the receiver returns the blocks, and all states unwind."""
    from bitswap_amd.codec import GroupedCodec
    model, zend, zcen = workload.build("cifar8", DEV, quantbits=10, small=16)
    B, n = 6, 3
    images = workload.synthetic_blocks(B * n, model.xs, seed=21).view(B, n, -1).to(torch.int32).to(DEV)
    init = initial_states(B, 12000)
    gc = GroupedCodec(model, zend, zcen, groups=2, quantbits=10, bitswap=bool(bitswap))
    states = gc.new_states(B, n, states=init)
    gc.encode_blocks(states, images)
    torch.cuda.synchronize()
    gc.check(states)
    got = gc.to_lists(states)
    # the same two groups through the plain codec, one after the other on the default stream
    plain = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=bool(bitswap))
    want = []
    for sl in gc.split(B):
        st = plain.new_states(sl.stop - sl.start, n, states=init[sl])
        for xi in range(n):
            plain.encode_block(st, images[sl, xi])
        want += st.to_lists()
    assert got == want
    out = gc.decode_blocks(states, n)
    torch.cuda.synchronize()
    gc.check(states)
    assert torch.equal(out, images) and gc.to_lists(states) == init


def test_config3_shape_many_blocks_lossless():
    """BASELINE configs[2] shape at reduced length: ImageNet32 nz=4 full-width model, 40 chains x 6 blocks
    through the grouped codec; lossless, every state restored, bit accounting monotone."""
    from bitswap_amd.codec import GroupedCodec
    model, zend, zcen = workload.build("imagenet4", DEV, quantbits=10)
    B, n = 40, 6
    images = workload.synthetic_blocks(B * n, model.xs, seed=33).view(B, n, -1).to(torch.int32).to(DEV)
    init = initial_states(B)
    gc = GroupedCodec(model, zend, zcen, groups=2, quantbits=10, bitswap=True)
    states = gc.new_states(B, n, states=init)
    gc.encode_blocks(states, images)
    lens = torch.cat([st.len for st in states]).cpu().numpy()
    assert np.all(lens > 10000 - 1)                     # every chain grew
    out = gc.decode_blocks(states, n)
    torch.cuda.synchronize()
    gc.check(states)
    assert torch.equal(out, images) and gc.to_lists(states) == init


def test_ragged_chains_on_gpu():
    """config 4 shape: images of different sizes = chains of different lengths in one lock-step run on the
    HIP kernels (prefix views of the state tensors); streams equal the chains coded alone (nn_batch keeps the
    convs batch-invariant), the receiver returns every block."""
    model, zend, zcen = workload.build("imagenetcrop4", DEV, quantbits=10, small=16, nn_batch=4)
    lens = [3, 1, 5, 2, 5]
    chains = [workload.synthetic_blocks(n, model.xs, seed=60 + i).to(torch.int32) for i, n in enumerate(lens)]
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    state, order, met = codec.compress_ragged(chains)
    lists = state.to_lists()
    assert met["nblocks"].tolist() == sorted(lens, reverse=True) and np.all(met["cma"] > 0)
    for k, i in enumerate(order[:3]):
        alone, _, _ = codec.compress_ragged([chains[i]])
        assert alone.to_lists()[0] == lists[k]
    out = codec.decompress_ragged(state, met["nblocks"])
    for k, i in enumerate(order):
        assert torch.equal(out[k].cpu(), chains[i])
    assert state.to_lists() == [initial_states(1)[0]] * len(lens)


def test_full_width_batch_invariance_of_a_chain():
    """The crop/demo contract at full model width (reswidth 256, Winograd-domain GEMMs at nn_batch = 32): a chain
    coded in a ragged batch of 40 has exactly the stream it gets when coded alone, i.e. an image compressed in a
    batch decompresses on its own.  (Needs MIOpen pinned to deterministic algorithms: BitSwapCodec sets it.)"""
    torch.backends.cudnn.deterministic = False      # the codec must not rely on the caller for this
    model, zend, zcen = workload.build("imagenetcrop4", DEV, quantbits=10, nn_batch=32)
    lens = [2, 1, 2] + [1] * 37
    chains = [workload.synthetic_blocks(n, model.xs, seed=80 + i).to(torch.int32) for i, n in enumerate(lens)]
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=True)
    state, order, met = codec.compress_ragged(chains)
    lists = state.to_lists()
    for k in (0, 2, 17):
        alone, _, _ = codec.compress_ragged([chains[order[k]]])
        assert alone.to_lists()[0] == lists[k]
    one = codec.new_states(1, 2, states=[lists[0]])
    out = codec.decompress_ragged(one, [2])
    assert torch.equal(out[0].cpu(), chains[order[0]])


def test_sender_and_receiver_in_separate_processes(tmp_path):
    """The receiver is another process: everything the conv stacks compute (MIOpen algorithm choice, BLAS heuristics,
    the Winograd-domain GEMMs at 26 blocks per call) must come out bit-identical there, or the streams do not decode."""
    import subprocess, sys
    from conftest import ROOT
    import os
    f = str(tmp_path / "streams.npz")
    tool = os.path.join(ROOT, "tools", "xproc_codec.py")
    enc = subprocess.run([sys.executable, tool, "enc", f], capture_output=True, text=True, timeout=300)
    assert enc.returncode == 0 and "encoded" in enc.stdout, enc.stderr[-2000:]
    dec = subprocess.run([sys.executable, tool, "dec", f], capture_output=True, text=True, timeout=300)
    assert dec.returncode == 0 and "decoded ok" in dec.stdout, (dec.stdout + dec.stderr)[-2000:]


class _Count:
    """Count calls of a bitswap_amd.hip entry point for the duration of a block (is the Winograd route live?)."""

    def __init__(self, name):
        from bitswap_amd import hip
        self.hip, self.name, self.n = hip, name, 0

    def __enter__(self):
        self.orig = getattr(self.hip, self.name)

        def wrapped(*a, **k):
            self.n += 1
            return self.orig(*a, **k)
        setattr(self.hip, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.hip, self.name, self.orig)


@pytest.mark.parametrize("name,bitswap", [("cifar8", 1), ("imagenet4", 1), ("imagenet4", 0)])
def test_full_width_oracle_word_parity(name, bitswap):
    """BASELINE configs 2, 3 and 5 at FULL model width (reswidth 252 / 254, Z = 2048, X = 3072, K = 1024 / 256) with
    enough chains per call (26 >= gemm_min_batch) that the conv stacks take the route the bench takes -- Winograd-domain
    batched GEMMs -- and the production kernel pair (k_logistic wave layout, CDF spec 2 + k_rans_pop_wave +
    systolic push): the oracle replays the schedule on the CPU with the GPU's conv outputs and must produce the very
    same words (mnist_compress.py:176-251); then the GPU receiver returns the blocks and unwinds every chain."""
    model, zend, zcen = workload.build(name, DEV, quantbits=10)
    B, n = 26, 1
    assert model.fused and model.conv_algo == "winograd" and B >= model.gemm_min_batch
    images = workload.synthetic_blocks(B * n, model.xs, seed=17).view(B, n, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, bitswap=bool(bitswap))
    assert codec.cdf_spec == 2 and all(s is not None for s in codec.zstep[:-1]) and codec.zstep[-1] is None
    from bitswap_amd import hip
    assert codec.backend.table_layout(codec.K) == hip.LAYOUT_WAVE
    rec, plain_net = record_nets(codec)
    with _Count("wino_fused") as wf:
        state, met = codec.compress(images.to(DEV))
    assert wf.n > 0, "the Winograd-domain conv route was not taken"
    sent = state.to_lists()

    it = iter(rec)
    oc = BitSwapCodec(model, zend.cpu(), zcen.cpu(), quantbits=10, bitswap=bool(bitswap),
                      backend=OracleBackend(O.MODE_DET, threads=16))
    oc._net = lambda fn, given: tuple(t.cpu() for t in next(it))
    ostate, omet = oc.compress(images)
    assert ostate.to_lists() == sent
    assert np.array_equal(omet["cma"], met["cma"]) and np.array_equal(omet["rest_len"], met["rest_len"])

    codec._net = plain_net
    out = codec.decompress(state, n)
    assert torch.equal(out.cpu(), images)
    assert state.to_lists() == initial_states(B)


@pytest.mark.parametrize("name", ["cifar8", "imagenet4"])
def test_full_width_winograd_matches_torch_modules(name):
    """The conv route of the bench (fused epilogues + Winograd-domain batched GEMMs at full width, 26 blocks per
    call) against the plain torch modules of the same Model (MIOpen direct convolutions, separate pointwise ops):
    every infer(i) / generate(i) output within 5e-4 of the output range, bitwise repeatable."""
    model, _, _ = workload.build(name, DEV, quantbits=6)
    model.compress(True)
    g = torch.Generator().manual_seed(1)
    N = 26
    worst = 0.0
    with torch.no_grad(), _Count("wino_fused") as wf:
        for i in range(model.nz):
            x = (torch.randint(0, 256, (N, model.xdim), generator=g).float() - 127.5) / 127.5
            zin = torch.randn((N, model.zdim_flat), generator=g)
            for fn, inp in ((model.infer(i), (x if i == 0 else zin).to(DEV)), (model.generate(i), zin.to(DEV))):
                model.fused = True
                mu_f, sc_f = fn(inp)
                mu_f2, sc_f2 = fn(inp)
                model.fused = False
                mu_t, sc_t = fn(inp)
                model.fused = True
                assert torch.equal(mu_f, mu_f2) and torch.equal(sc_f, sc_f2)
                for a, b in ((mu_f, mu_t), (sc_f, sc_t.expand_as(sc_f))):
                    rng = float(b.abs().max()) + 1e-6
                    worst = max(worst, float((a - b).abs().max()) / rng)
    assert wf.n > 0
    print(f"full-width {name}: max |fused - torch| / range = {worst:.2e}")
    assert worst < 5e-4, worst


@pytest.mark.parametrize("sched", ["bitswap", "bbans"])
def test_gpu_bits_per_dim_matches_reference(golden, sched):
    """north_star: bits/dim within 1e-4 of the reference.  The reference's own chain (model weights, bins, images and
    bit accounting produced by the reference code, tests/golden/make_golden.py) coded UNTETHERED on the GPU: our Model
    with fused epilogues and the Winograd-domain convs forced on (gemm_min_batch = 1), the production HIP kernels, CDF
    spec 2.  cma (mnist_compress.py:253-261) must match to 1e-4 at every block, and so must the exact information
    content of the final state (32 bits per stack word + log2 of the head) per dimension; the receiver is lossless."""
    g = golden(f"chain_rgb4_small_{sched}.npz")
    cfg = g["cfg"]
    q, bitswap, nblocks = int(cfg[7]), bool(cfg[8]), int(cfg[9])
    model = load_golden_model(golden("model_rgb4_small.npz"), DEV).fold().fuse()
    model.gemm_min_batch = 1
    zend, _, zcen = chain_tables(g)
    codec = BitSwapCodec(model, torch.from_numpy(zend).to(DEV), torch.from_numpy(zcen).to(DEV), quantbits=q,
                         bitswap=bitswap)
    B = 3
    imgs = torch.from_numpy(g["images"].astype(np.int32)).view(1, nblocks, -1).expand(B, -1, -1).contiguous()
    init = [reference_init_state()] * B
    with _Count("wino_fused") as wf:
        state, met = codec.compress(imgs.to(DEV), state=codec.new_states(B, nblocks, states=init))
    assert wf.n > 0
    X = model.xdim
    for b in range(B):
        assert np.abs(met["cma"][b] - g["cma"]).max() <= 1e-4, (met["cma"][b], g["cma"])
        assert np.abs(met["nets"][b] - g["nets"]).max() <= 1e-4
    ref = words_to_state(g["sent_words"])
    info_ref = 32 * (len(ref) - 1) + np.log2(float(ref[-1]))
    for st in state.to_lists():
        info = 32 * (len(st) - 1) + np.log2(float(st[-1]))
        assert abs(info - info_ref) / (X * nblocks) <= 1e-4
    out = codec.decompress(state, nblocks)
    assert torch.equal(out.cpu(), imgs) and state.to_lists() == init
