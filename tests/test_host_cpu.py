"""CPU: host logic of the package (Model, bins, schedules, bit accounting, CLI artefacts, tiling,
container) with the oracle doing the coding arithmetic.  The reference chain fixtures are
reproduced WITHOUT teacher forcing: our Model + our schedules + oracle == the reference's words."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from oracle.backend import OracleBackend
from bitswap_amd import bins, cli, container, rand, tiling, workload
from bitswap_amd.codec import BitSwapCodec, initial_states
from bitswap_amd.model import Model
from conftest import chain_tables, load_golden_model, reference_init_state, words_to_state


def load_model(g, **kw):
    return load_golden_model(g, **kw)


@pytest.mark.parametrize("fold", [False, True])
def test_model_matches_reference_outputs(golden, fold):
    g = golden("model_mnist_small.npz")
    m = load_model(g)
    if fold:
        m.fold()
    with torch.no_grad():
        mu, sc = m.infer(0)(torch.from_numpy(g["infer0_in"].astype(np.float32)))
        assert np.array_equal(mu.numpy(), g["infer0_mu"]) and np.array_equal(sc.numpy(), g["infer0_scale"])
        z = torch.from_numpy(g["z_in"])
        for i in range(2):
            mu, sc = m.generate(i)(z)
            assert np.array_equal(mu.numpy(), g[f"gen{i}_mu"])
            assert np.array_equal(np.broadcast_to(sc.numpy(), mu.shape), g[f"gen{i}_scale"])
        mu, sc = m.infer(1)(z)
        assert np.array_equal(mu.numpy(), g["infer1_mu"]) and np.array_equal(sc.numpy(), g["infer1_scale"])


def test_crop_model_conditional_gen_std(golden):
    g = golden("model_crop_small.npz")
    m = load_model(g, conditional_gen_std=True)
    with torch.no_grad():
        mu, sc = m.generate(0)(torch.from_numpy(g["z_in"]))
        assert np.array_equal(mu.numpy(), g["gen0_mu"]) and np.array_equal(sc.numpy(), g["gen0_scale"])
        mu, sc = m.infer(0)(torch.from_numpy(g["infer0_in"].astype(np.float32)))
        assert np.array_equal(mu.numpy(), g["infer0_mu"]) and np.array_equal(sc.numpy(), g["infer0_scale"])


def test_model_compress_mode_shapes_and_chunking(golden):
    g = golden("model_mnist_small.npz")
    m = load_model(g).fold()
    m.compress()
    x = (torch.from_numpy(g["infer0_in"].astype(np.float64)).view(5, -1) - 127.5) / 127.5
    with torch.no_grad():
        mu, sc = m.infer(0)(x)                       # batched: float32 [B, Z]
        assert mu.shape == (5, 256) and mu.dtype == torch.float32
        mu1, _ = m.infer(0)(x[0])                    # reference call pattern: 1-D in, 1-D out, input dtype
        assert mu1.shape == (256,) and mu1.dtype == torch.float64
        m.nn_batch = 2                               # fixed-shape micro-batches, zero padded
        mu2, sc2 = m.infer(0)(x)
        assert mu2.shape == mu.shape and torch.allclose(mu2, mu, atol=1e-5)
        mu3, sc3 = m.generate(0)(torch.from_numpy(g["z_in"]).view(5, -1))
        assert mu3.shape == (5, 1024) and sc3.shape == (5, 1024)


def test_bins_match_reference(golden):
    b = golden("bins.npz")
    for q in (10, 8):
        bb = rand.Bins(torch.zeros((1, 1, 8)), torch.ones((1, 1, 8)), q)
        assert np.array_equal(bb.endpoints().numpy()[0, 0, 0], b[f"top_endpoints_q{q}"])
        assert np.array_equal(bb.centres().numpy()[0, 0, 0], b[f"top_centres_q{q}"])
    ib = rand.ImageBins(torch.float64, "cpu", 5)
    assert np.array_equal(ib.endpoints().numpy()[0], b["x_endpoints"])
    assert np.array_equal(ib.centres().numpy()[0], b["x_centres"])
    assert ib.endpoints().stride(0) == 0             # expanded view, like the reference
    s = b["kbins_samples"].reshape(-1, 32)
    e, c = bins.uniform_bins(s.min(0), s.max(0), 6)  # closed form of KBinsDiscretizer(strategy='uniform')
    assert np.array_equal(e, b["kbins_endpoints_q6"]) and np.array_equal(c, b["kbins_centres_q6"])
    te, tc = bins.top_bins(8, 10)
    assert np.array_equal(te[5], b["top_endpoints_q10"]) and np.array_equal(tc[0], b["top_centres_q10"])


def test_uniform_step_decides_the_cdf_spec():
    """bins.uniform_step: the bin width of uniform-width rows (what discretize() makes below the top layer and what
    ImageBins is), None for anything else -- the top layer's equal-mass bins, a perturbed row."""
    rng = np.random.RandomState(0)
    lo, hi = rng.uniform(-9, -2, 7).astype(np.float16).astype(np.float64), rng.uniform(2, 9, 7)
    e, _ = bins.uniform_bins(lo, hi, 10)
    h = bins.uniform_step(torch.from_numpy(e))
    assert h is not None and h.dtype == np.float64 and np.allclose(h, (hi - lo) / 1024, rtol=1e-12)
    assert np.array_equal(h, (e[:, -1] - e[:, 0]) / np.float64(1022))
    te, _ = bins.top_bins(4, 10)
    assert bins.uniform_step(te) is None
    xe = rand.ImageBins(torch.float64, "cpu", 5).endpoints()          # expanded view, row stride 0
    hx = bins.uniform_step(xe)
    assert hx is not None and np.allclose(hx, 2 / 255)
    bad = e.copy()
    bad[3, 500] += 1e-9
    assert bins.uniform_step(bad) is None
    # the oracle decides on its own, and identically
    for tab in (e, te, xe, bad):
        a, b = bins.uniform_step(tab), OracleBackend.uniform_step(tab)
        assert (a is None and b is None) or np.array_equal(a, b)
    # the codec: spec 2 on the uniform layers and the pixels, spec 1 on the top layer; cdf_spec=1 turns it off
    model, zend, zcen = workload.build("cifar8", "cpu", quantbits=8, small=8)
    c2 = BitSwapCodec(model, zend, zcen, quantbits=8, backend=OracleBackend(O.MODE_DET))
    assert [s is not None for s in c2.zstep] == [True] * 7 + [False] and c2.xstep is not None
    c1 = BitSwapCodec(model, zend, zcen, quantbits=8, backend=OracleBackend(O.MODE_DET), cdf_spec=1)
    assert all(s is None for s in c1.zstep) and c1.xstep is None
    c0 = BitSwapCodec(model, zend, zcen, quantbits=8, backend=OracleBackend(O.MODE_LIBM))
    assert all(s is None for s in c0.zstep)            # the libm mode restates the reference formula only


def test_discretize_sampling_and_cache(tmp_path):
    m = Model(xs=(1, 32, 32), nz=3, zchannels=1, nprocessing=1, resdepth=3, reswidth=8).eval()
    data = torch.randint(0, 256, (64, 1, 32, 32), dtype=torch.uint8)
    ze, zc = bins.discretize(3, 6, torch.float64, "cpu", m, "toy", data=data, ppb=2, cache_dir=str(tmp_path))
    assert ze.shape == (3, 256, 63) and zc.shape == (3, 256, 64) and ze.dtype == torch.float64
    assert bool((ze[:, :, 1:] > ze[:, :, :-1]).all())
    assert bool(((zc[:, :, :-1] < ze) & (ze < zc[:, :, 1:])).all())
    assert os.path.exists(tmp_path / "toy_nz3_zendpoints6.pt")          # the reference's cache file names
    ze2, _ = bins.discretize(3, 6, torch.float64, "cpu", m, "toy", cache_dir=str(tmp_path))   # no data needed now
    assert torch.equal(ze, ze2)
    with pytest.raises(FileNotFoundError):
        bins.discretize(3, 7, torch.float64, "cpu", m, "toy", cache_dir=str(tmp_path))


@pytest.mark.parametrize("chain,model_file", [("chain_mnist_small_bitswap", "model_mnist_small"),
                                              ("chain_mnist_small_bbans", "model_mnist_small"),
                                              ("chain_rgb4_small_bitswap", "model_rgb4_small"),
                                              ("chain_rgb4_small_bbans", "model_rgb4_small")])
def test_codec_reproduces_reference_chain(golden, chain, model_file):
    """Our Model + our batched schedule + the oracle = the reference's own sender run
    (mnist_compress.py:164-263): identical word stream and bit accounting, then the receiver
    (:277-358) returns the images and the initial state.  mnist: 1 channel, nz = 2, both schedules;
    rgb4: 3 channels, nz = 4 (layer ordering for zi > 1, 3072-dim pixel op)."""
    g = golden(chain + ".npz")
    cfg = g["cfg"]
    q, bitswap, nblocks = int(cfg[7]), bool(cfg[8]), int(cfg[9])
    m = load_model(golden(model_file + ".npz")).fold()
    zend, _, zcen = chain_tables(g)
    codec = BitSwapCodec(m, torch.from_numpy(zend), torch.from_numpy(zcen), quantbits=q,
                         bitswap=bitswap, backend=OracleBackend(O.MODE_DET))
    imgs = torch.from_numpy(g["images"].astype(np.int32)).view(1, nblocks, -1)
    state, met = codec.compress(imgs)
    assert state.to_lists()[0] == words_to_state(g["sent_words"])
    assert np.allclose(met["cma"][0], g["cma"]) and np.allclose(met["nets"][0], g["nets"])
    assert int(met["rest_len"][0]) + 1 == int(g["restbits_len"])
    out = codec.decompress(state, nblocks)
    assert torch.equal(out, imgs)
    assert state.to_lists()[0] == reference_init_state()


@pytest.mark.parametrize("bitswap", [1, 0])
def test_codec_many_chains_round_trip(bitswap):
    # nn_batch: convs always see [2, ...] micro-batches, so a chain's (mu, scale) bits -- and therefore
    # its stream -- do not depend on which other chains are coded alongside it
    model, zend, zcen = workload.build("cifar8", "cpu", quantbits=6, small=8, nn_batch=2)
    B, n = 3, 2
    images = workload.synthetic_blocks(B * n, model.xs, seed=3).view(B, n, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=6, bitswap=bool(bitswap),
                         backend=OracleBackend(O.MODE_DET, threads=3))
    st, met = codec.compress(images)
    assert met["nets"].shape == (B, n) and np.all(met["total"][:, -1] > 0)
    # chains are independent: chain 1 alone gives the same stream
    st1, _ = codec.compress(images[1:2], state=codec.new_states(1, n, states=[initial_states(B)[1]]))
    assert st1.to_lists()[0] == st.to_lists()[1]
    out = codec.decompress(st, n)
    assert torch.equal(out, images) and st.to_lists() == initial_states(B)


def test_two_codecs_of_different_cdf_spec_on_one_backend_object_are_refused():
    """the backend object carries the CDF spec of its uniform-bin tables, the codec its own copy for the (f, c) of
    the split push and for the fingerprint.  A second codec with another spec on the SAME backend object re-configures the
    backend under the first one: that codec must refuse to code (tables of one spec, (f, c) of another, under a fingerprint
    that looks valid) instead of writing an undecodable stream.  A backend object per codec is fine."""
    model, zend, zcen = workload.build("mnist2", "cpu", quantbits=8, small=8)
    images = workload.synthetic_blocks(2, model.xs, seed=3).view(1, 2, -1).to(torch.int32)
    shared = OracleBackend(O.MODE_DET)
    c3 = BitSwapCodec(model, zend, zcen, quantbits=8, backend=shared, cdf_spec=3)
    assert any(s is not None for s in c3.zstep) or c3.xstep is not None     # (there IS a uniform-bin table at stake)
    st, _ = c3.compress(images)
    assert torch.equal(c3.decompress(st, 2), images)
    c2 = BitSwapCodec(model, zend, zcen, quantbits=8, backend=shared, cdf_spec=2)
    with pytest.raises(RuntimeError, match="CDF spec 3 .* spec 2"):
        c3.compress(images)
    st, _ = c2.compress(images)                                              # the codec the backend now belongs to works
    assert torch.equal(c2.decompress(st, 2), images)
    c3b = BitSwapCodec(model, zend, zcen, quantbits=8, backend=OracleBackend(O.MODE_DET), cdf_spec=3)
    st, _ = c3b.compress(images)
    assert torch.equal(c3b.decompress(st, 2), images)
    st, _ = c2.compress(images)                                              # ... and is not disturbed by c3b's own backend
    assert torch.equal(c2.decompress(st, 2), images)


def test_too_few_initial_bits_is_reported():
    """BB-ANS pops all nz layers first (config 5): 3000 initial words cannot feed 8 x 2048 x 6 bits."""
    model, zend, zcen = workload.build("cifar8", "cpu", quantbits=6, small=8)
    images = workload.synthetic_blocks(1, model.xs, seed=3).view(1, 1, -1).to(torch.int32)
    codec = BitSwapCodec(model, zend, zcen, quantbits=6, bitswap=False, backend=OracleBackend(O.MODE_DET))
    with pytest.raises(RuntimeError):
        codec.compress(images, nwords=2000)


def test_cli_experiment_artefacts(tmp_path):
    r = cli.compress(6, 2, 1, 0, dataset="mnist", experiments=3, ndatapoints=2, decompress=True,
                     outdir=str(tmp_path), backend=OracleBackend(O.MODE_DET), small=8, verbose=False)
    for k in ("nets", "elbos", "cmas", "total"):
        assert r[k].shape == (3, 2)
        assert os.path.exists(tmp_path / "plots" / "mnist2" / f"bitswap_6bits_{k}.npy")   # mnist_compress.py:363-366
    for e in (1, 2, 3):
        st = container.load_state(tmp_path / "bitstreams" / "mnist" / "nz2" / "Bit-Swap" /
                                  f"Bit-Swap_6bits_nz2_experiment{e}")                      # :265-267
        assert isinstance(st, list) and st[-1] >= 1 << 32
    import json
    meta = json.load(open(tmp_path / "bitstreams" / "mnist" / "nz2" / "Bit-Swap" / "stream_meta.json"))
    assert meta["stream_format"] == "reference" and meta["cdf_spec"] == {"z": [1, 1], "x": 4}   # latents K = 64: spec 1; pixels K = 256: the default uniform-bin spec (4)
    assert meta["conv_route"]["chains_per_call"] == 3 and meta["world_size"] == 1
    # net bit rate formula (:254,258): cumulative nets * xdim * ndatapoints = words added * 32
    assert np.all(r["total"] > 0) and np.isfinite(r["elbos"]).all()


def test_initial_words_follow_the_reference_draw_order(golden, tmp_path):
    """mnist_compress.py:94,133-137,158: numpy is seeded once, `choice(len(test_set), (100, 100), replace=False)`
    consumes the generator (the exists() guard never hits: np.save appends .npy), THEN experiment i draws its 10000 initial
    words.  draws.npz holds what that sequence yields for test sets of 10000 (MNIST / CIFAR-10) and 50000 (ImageNet32)
    images; codec.reference_draws must hold the same indices and words, and cli.compress must start its experiments there."""
    from bitswap_amd.codec import reference_draws
    g = golden("draws.npz")
    for ntest in (10000, 50000):
        idx, inits = reference_draws(ntest, 100, 100)
        assert np.array_equal(idx[:3, :8], g[f"n{ntest}_indices"])
        for ei in range(3):
            assert inits[ei][:4] == [int(v) for v in g[f"n{ntest}_e{ei}_first4"]]
            assert inits[ei][-2:] == [int(v) for v in g[f"n{ntest}_e{ei}_last2"]]
    assert initial_states(3)[0][:4] != inits[0][:4]          # the fallback (seed, then the words at once) is NOT that sequence
    # the CLI: 3 experiments x 2 datapoints out of >= 512 synthetic images -> the reference sequence for that shape
    seen = {}
    orig = BitSwapCodec.new_states

    def spy(self, B, n, **kw):
        seen["states"] = [list(s) for s in kw["states"]]
        return orig(self, B, n, **kw)
    BitSwapCodec.new_states = spy
    try:
        cli.compress(6, 2, 1, 0, dataset="mnist", experiments=3, ndatapoints=2, decompress=True, outdir=str(tmp_path),
                     backend=OracleBackend(O.MODE_DET), small=8, verbose=False)
    finally:
        BitSwapCodec.new_states = orig
    idx, want = reference_draws(512, 3, 2)
    assert seen["states"] == want
    assert np.array_equal(np.load(tmp_path / "bitstreams" / "mnist" / "indices.npy"), idx)
    cli.decompress_streams(6, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=OracleBackend(O.MODE_DET), small=8,
                           verbose=False)                  # the receiver finds the same initial words (:358)


def test_tiling_and_container_round_trip():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (70, 100, 3)).astype(np.uint8)
    blocks, h, w = tiling.extract_blocks(img)
    assert blocks.shape == (6, 32, 32, 3) and (h, w) == (64, 96)
    assert np.array_equal(tiling.unextract_blocks(blocks, h, w), img[:h, :w])
    assert np.array_equal(blocks[1], img[0:32, 32:64])            # row-major grid order
    flat = tiling.blocks_to_chw_flat(blocks)
    assert np.array_equal(flat[2].reshape(3, 32, 32)[1], blocks[2][:, :, 1])
    assert np.array_equal(tiling.chw_flat_to_blocks(flat), blocks)
    state = [11, 22, 33, 44, (7 << 32) | 5]
    arr = container.pack(state, 2, 6, h, w)
    assert arr.dtype == np.uint32 and arr.tolist() == [33, 44, 5, 7, 6, 64, 96]   # demo_compress.py:272-283
    st, nb, hh, ww = container.unpack(arr)
    assert (st, nb, hh, ww) == ([33, 44, (7 << 32) | 5], 6, 64, 96)               # demo_decompress.py:221-227


def test_demo_surface_matches_reference_files(golden, tmp_path):
    """The on-disk surface against files the REFERENCE wrote (tests/golden/make_golden.py::make_surface_fixture runs
    the reference's own extract_blocks, demo_compress.compress and demo_decompress.decompress): our tiling gives the
    same blocks; our Model + lock-step schedule + oracle, untethered, produce the SAME container words
    (demo_compress.py:159-160,272-283) and the same .npy bytes; reading the reference's container
    (demo_decompress.py:221-227) and decoding it returns the image; the per-experiment pickle
    (mnist_compress.py:265-267) is byte-identical."""
    import pickle
    g = golden("demo_surface.npz")
    q = int(g["cfg"][7])
    blocks, h, w = tiling.extract_blocks(g["image"])
    assert np.array_equal(blocks, g["blocks"]) and [h, w] == g["hw"].tolist()
    assert np.array_equal(tiling.unextract_blocks(blocks, h, w), g["crop"])
    model = load_golden_model(g, conditional_gen_std=True).fold()
    zend, _, zcen = chain_tables(g)
    setup = (model, torch.from_numpy(zend), torch.from_numpy(zcen), torch.device("cpu"))
    # with OUR deterministic CDF the chain follows the reference's word for word until the first table entry where
    # torch's sigmoid and ours round differently (about 0.1 ppm of the entries: SURVEY 7a); from there on it is a
    # different, equally long chain.  Same start of the kept words, same trailer, a length within sampling noise.
    (state, min_words, bpd), = cli.compress_images([blocks], quantbits=q, nz=model.nz, setup=setup,
                                                   backend=OracleBackend(O.MODE_DET))
    arr = container.pack(state, min_words, len(blocks), h, w)
    ref = g["container"]
    assert arr[-3:].tolist() == ref[-3:].tolist() and abs(len(arr) - len(ref)) <= 64
    same = int(np.argmin(arr[: min(len(arr), len(ref))] == ref[: min(len(arr), len(ref))]))
    assert same >= 1000 or np.array_equal(arr, ref)                # at least the whole first block (905 words)
    # with the reference's formula evaluated by torch itself the words are the reference's, all of them -- provided this
    # torch build rounds its float64 sigmoid like the one that wrote the fixture
    probe = torch.sigmoid(torch.from_numpy(np.random.RandomState(99).uniform(-40, 40, 1 << 16))).numpy()
    if not np.array_equal(probe, g["sigmoid_probe"]):
        pytest.skip("this torch build rounds float64 sigmoid differently from the one that generated the fixture")
    ob = OracleBackend(O.MODE_TORCH)
    (state, min_words, bpd), = cli.compress_images([blocks], quantbits=q, nz=model.nz, setup=setup, backend=ob)
    arr = container.pack(state, min_words, len(blocks), h, w)
    assert arr.dtype == np.uint32 and np.array_equal(arr, g["container"])
    np.save(tmp_path / "img_bitswap", arr)
    assert open(tmp_path / "img_bitswap.npy", "rb").read() == g["container_npy"].tobytes()
    assert abs(bpd - 32 * (len(arr) - 5 + 1) / (len(blocks) * 3072)) < 0.05     # (len(state) - (len(restbits)-1)) words
    # the receiver on the reference's file
    st, nb, hh, ww = container.unpack(np.load(tmp_path / "img_bitswap.npy"))
    assert (nb, hh, ww) == (len(blocks), h, w) and st == words_to_state(g["state_words"])
    out, rest = cli.decompress_image(st, nb, quantbits=q, nz=model.nz, setup=setup, backend=ob)
    assert np.array_equal(tiling.unextract_blocks(out, hh, ww), g["crop"])
    assert rest == words_to_state(g["rest_words"]) == reference_init_state()[min_words:]
    # experiment pickles
    assert pickle.loads(g["state_pickle"].tobytes()) == st
    container.save_state(tmp_path / "exp1", st)
    assert open(tmp_path / "exp1", "rb").read() == g["state_pickle"].tobytes()
    assert container.load_state(tmp_path / "exp1") == st


def test_discretize_replays_reference_sampling(golden):
    """bins.discretize driven with the reference's noise and data order (tests/golden/make_golden.py::
    make_discretize_fixture replays discretization.py:55-83 around the imported reference pieces): on the CPU, where our
    Model is bitwise the reference's, the float16 samples -- hence the per-dimension minima/maxima and all bins -- are
    identical."""
    g = golden("discretize_small.npz")
    cfg = g["cfg"]
    nz, q = int(cfg[1]), int(cfg[7])
    model = load_golden_model(g)
    torch.manual_seed(int(g["noise_seed"]))
    eps = lambda shape: rand.logistic_eps(shape, device="cpu", bound=1e-30)
    ze, zc = bins.discretize(nz, q, torch.float64, "cpu", model, "toy", data=torch.from_numpy(g["images"]), ppb=2,
                             save=False, cache_dir="/nonexistent", eps_fn=eps, order=g["order"])
    K = 1 << q
    for zi in range(nz - 1):
        e, c = bins.uniform_bins(g["z_mins"][zi], g["z_maxs"][zi], q)
        assert np.array_equal(ze[zi].numpy(), e) and np.array_equal(zc[zi].numpy(), c)
    assert np.array_equal(ze[0, 5].numpy(), g["zend_layer0_dim5"]) and np.array_equal(zc[0, 5].numpy(), g["zcen_layer0_dim5"])
    assert np.array_equal(ze[nz - 1, 0].numpy(), g["z_top_endpoints"])
    assert np.array_equal(zc[nz - 1, 3].numpy(), g["z_top_centres"]) and ze.shape == (nz, 512, K - 1)


def test_demo_image_path_with_trimmed_container():
    ob = OracleBackend(O.MODE_DET)
    setup = cli.crop_setup(-1, nz=2, quantbits=6, backend=ob, small=8)
    rng = np.random.RandomState(1)
    a = tiling.extract_blocks(rng.randint(0, 256, (64, 96, 3)).astype(np.uint8))[0]
    b = a[:2]
    res = cli.compress_images([a, b], quantbits=6, nz=2, setup=setup, backend=ob)
    for (st, min_words, bpd), blk in zip(res, (a, b)):
        assert 0 < min_words < 10000 and bpd > 0
        arr = container.pack(st, min_words, len(blk), 64, 96)
        assert len(arr) < 10000                                     # untouched initial words are gone
        st2, nb, _, _ = container.unpack(arr)
        out, rest = cli.decompress_image(st2, nb, quantbits=6, nz=2, setup=setup, backend=ob)
        assert np.array_equal(out, blk)
        assert rest == reference_init_state()[min_words:]           # what is left is the kept tail of the initial stack


def test_ragged_lock_step_equals_chains_coded_alone():
    """compress_ragged: chains of different length run in lock-step and drop out from the back of the
    (length-sorted) batch.  With nn_batch the conv outputs do not depend on the batch composition, so every
    chain's stream equals the one obtained by coding that chain on its own; the ragged receiver returns all
    blocks and unwinds every state."""
    ob = OracleBackend(O.MODE_DET)
    model, zend, zcen = workload.build("imagenetcrop4", "cpu", quantbits=6, small=8, nn_batch=2)
    lens = [2, 4, 1, 3]
    chains = [workload.synthetic_blocks(n, model.xs, seed=40 + i).to(torch.int32) for i, n in enumerate(lens)]
    codec = BitSwapCodec(model, zend, zcen, quantbits=6, bitswap=True, backend=ob)
    state, order, met = codec.compress_ragged(chains)
    assert [lens[i] for i in order] == [4, 3, 2, 1] and met["nblocks"].tolist() == [4, 3, 2, 1]
    lists = state.to_lists()
    for k, i in enumerate(order):
        alone, _, m1 = codec.compress_ragged([chains[i]])
        assert alone.to_lists()[0] == lists[k]
        assert m1["total"][0] == met["total"][k]
    out = codec.decompress_ragged(state, met["nblocks"])
    for k, i in enumerate(order):
        assert torch.equal(out[k], chains[i])
    one = reference_init_state()
    assert state.to_lists() == [one] * 4


@pytest.mark.parametrize("m,r", [(4, 3), (2, 5), (4, 5)])
def test_winograd_matrices_and_conv_identity(m, r):
    """bitswap_amd/winograd.py on the CPU: the Cook-Toom triple (A^T, G, B^T) satisfies the bilinear identity,
    F(4,3) is the familiar Lavin-Gray set, and the whole transform-domain pipeline (V = B^T d B per tile, one
    batched product with U = G w G^T, Y = A^T M A) equals F.conv2d in float64 to 1e-12."""
    from bitswap_amd import winograd as W
    pts = W.POINTS if m + r - 1 == 6 else W.POINTS8
    AT, G, BT = W.cook_toom(m, r, pts)
    n = m + r - 1
    rng = np.random.RandomState(m * 10 + r)
    d, g = rng.randn(n), rng.randn(r)
    want = np.array([sum(d[o + k] * g[k] for k in range(r)) for o in range(m)])
    assert np.abs(AT @ ((G @ g) * (BT @ d)) - want).max() < 1e-12
    if (m, r) == (4, 3):
        assert np.array_equal(BT[0], [4, 0, -5, 0, 1, 0]) and np.array_equal(AT[3], [0, 1, -1, 8, -8, 1])
        assert np.allclose(G[1], [-1 / 6, -1 / 6, -1 / 6])
    # 2-D convolution of a 16x16 image through the transform domain
    torch.manual_seed(r)
    C, N = 6, 3
    x = torch.randn(N, C, 16, 16, dtype=torch.float64)
    w = torch.randn(C, C, r, r, dtype=torch.float64)
    cfg = (n, m)
    U = W.transform_weights(w.float(), cfg).double()                       # [n*n, Cout, Cin] (float32 storage)
    U64 = torch.einsum("ik,ockl,jl->ijoc", torch.from_numpy(G), w, torch.from_numpy(G)).reshape(n * n, C, C)
    assert torch.allclose(U, U64, rtol=1e-6, atol=1e-6)
    p, nt = r // 2, 16 // m
    xp = torch.nn.functional.pad(x, (p, p + m, p, p + m))
    BTt, ATt = torch.from_numpy(BT), torch.from_numpy(AT)
    tiles = torch.stack([torch.stack([xp[:, :, ty * m: ty * m + n, tx * m: tx * m + n] for tx in range(nt)], 2)
                         for ty in range(nt)], 2)                          # [N, C, nt, nt, n, n]
    V = torch.einsum("ik,bctskl,jl->ijcbts", BTt, tiles, BTt).reshape(n * n, C, N * nt * nt)
    M = torch.bmm(U64, V).reshape(n, n, C, N, nt, nt)
    Y = torch.einsum("ik,klcbts,jl->bctisj", ATt, M, ATt).reshape(N, C, 16, 16)
    ref = torch.nn.functional.conv2d(x, w, padding=p)
    assert float((Y - ref).abs().max()) < 1e-10 * float(ref.abs().max()) + 1e-10


@pytest.mark.parametrize("bitswap", [1, 0])
def test_wave64_format_round_trip_on_oracle(bitswap):
    """The opt-in 64-state stream format restated on the oracle (oracle/backend.py::Oracle64Backend): symbol d of every
    operation on state d % 64.  Lossless, every one of the 64 states of every chain unwinds to its share of the
    initial words, chains stay independent, and the rate is the reference format's up to a constant of a few words per
    state (64 heads, 64 word-granular low-water marks), whatever the number of blocks."""
    from oracle.backend import Oracle64Backend, split_state
    model, zend, zcen = workload.build("cifar8", "cpu", quantbits=8, small=8, nn_batch=2)
    B, n = 3, 2
    images = workload.synthetic_blocks(B * n, model.xs, seed=3).view(B, n, -1).to(torch.int32)
    c64 = BitSwapCodec(model, zend, zcen, quantbits=8, bitswap=bool(bitswap), backend=Oracle64Backend(O.MODE_DET, threads=3))
    st, met = c64.compress(images)
    lists = st.to_lists()
    assert len(lists) == B and all(len(ch) == 64 and all(sub[-1] >= 1 << 32 for sub in ch) for ch in lists)
    c1 = BitSwapCodec(model, zend, zcen, quantbits=8, bitswap=bool(bitswap), backend=OracleBackend(O.MODE_DET, threads=3))
    _, met1 = c1.compress(images)
    # 64 heads and 64 word-granular low-water marks instead of one: a few words per state, once per chain
    assert np.all(np.abs(met["total"][:, -1] - met1["total"][:, -1]) <= 64 * 4 * 32 + 2000)
    st1, _ = c64.compress(images[1:2], state=c64.new_states(1, n, states=[initial_states(B)[1]]))
    assert st1.to_lists()[0] == lists[1]
    out = c64.decompress(st, n)
    assert torch.equal(out, images)
    assert st.to_lists() == [split_state(s) for s in initial_states(B)]


def test_cli_experiments_in_wave64_format(tmp_path):
    from oracle.backend import Oracle64Backend
    r = cli.compress(6, 2, 1, 0, dataset="mnist", experiments=2, ndatapoints=2, decompress=True, outdir=str(tmp_path),
                     backend=Oracle64Backend(O.MODE_DET), small=8, verbose=False, fmt="wave64")
    assert r["cmas"].shape == (2, 2) and np.all(r["total"] > 0)
    st = container.load_state(tmp_path / "bitstreams" / "mnist" / "nz2" / "Bit-Swap" / "Bit-Swap_6bits_nz2_experiment1_wave64")
    assert len(st) == 64 and all(sub[-1] >= 1 << 32 for sub in st)
    import json
    meta = json.load(open(tmp_path / "bitstreams" / "mnist" / "nz2" / "Bit-Swap" / "stream_meta.json"))
    assert meta["backend"] == "oracle-wave64" and meta["stream_format"] == "wave64"


def test_wave64_container_and_demo_path():
    from oracle.backend import Oracle64Backend, split_state
    ob = Oracle64Backend(O.MODE_DET)
    setup = cli.crop_setup(-1, nz=2, quantbits=6, backend=ob, small=8)
    rng = np.random.RandomState(2)
    blocks = tiling.extract_blocks(rng.randint(0, 256, (64, 96, 3)).astype(np.uint8))[0]
    (st, mins, bpd), = cli.compress_images([blocks], quantbits=6, nz=2, setup=setup, backend=ob, fmt="wave64")
    assert len(st) == 64 and len(mins) == 64 and bpd > 0
    arr = container.pack64(st, mins, len(blocks), 64, 96)
    assert arr.dtype == np.uint32 and container.is_pack64(arr) and len(arr) < 10000
    assert not container.is_pack64(container.pack([5, 6, 7 << 32], 0, 1, 32, 32))     # a reference container
    st2, nb, h, w = container.unpack64(arr)
    assert (nb, h, w) == (6, 64, 96) and [s[-1] for s in st2] == [s[-1] for s in st]
    assert all(a == b[m:] for a, b, m in zip(st2, st, mins))
    out, rest = cli.decompress_image(st2, nb, quantbits=6, nz=2, setup=setup, backend=ob)
    assert np.array_equal(out, blocks)
    init = split_state(reference_init_state())
    assert rest == [s[m:] for s, m in zip(init, mins)]          # every state: the untouched tail of its initial words
    with pytest.raises(ValueError):
        container.unpack64(arr[:-1])


def test_elbo_metric_batched_equals_per_image_loss(monkeypatch):
    """model.elbo_bits (the `elbos` metric of the CLIs, mnist_compress.py:170-174) in one batched pass against the
    reference's formulation -- Model.loss() on one image at a time -- with the sampling noise pinned."""
    from bitswap_amd.model import elbo_bits
    m = workload.synthetic_model("cifar", 2, "cpu", small=8)
    x = workload.synthetic_blocks(5, m.xs, seed=1).view(-1, *m.xs).float()
    monkeypatch.setattr(rand, "logistic_eps", lambda shape, device, bound=1e-5: torch.full(shape, 0.3, device=device))
    got = elbo_bits(m, x, batch=2)
    m.compress(False)
    with torch.no_grad():
        want = torch.stack([(-lr + torch.sum(-ld + le)) for lr, ld, le, _ in (m.loss(x[i:i + 1]) for i in range(5))])
    assert torch.allclose(got, want, rtol=1e-5)


def test_container_tiling_and_sharding_properties():
    """Randomised round trips of the host-side formats: tiling of arbitrary image sizes, both containers with arbitrary
    trimming, LPT / round-robin sharding (every chain exactly once, LPT never worse than 4/3 of the ideal makespan + the
    largest item)."""
    from bitswap_amd import dist
    rng = np.random.RandomState(123)
    for _ in range(25):
        h, w = rng.randint(32, 200, size=2)
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        blocks, hh, ww = tiling.extract_blocks(img)
        assert (hh, ww) == (h - h % 32, w - w % 32) and blocks.shape == ((hh // 32) * (ww // 32), 32, 32, 3)
        assert np.array_equal(tiling.unextract_blocks(blocks, hh, ww), img[:hh, :ww])
        assert np.array_equal(tiling.chw_flat_to_blocks(tiling.blocks_to_chw_flat(blocks)), blocks)
        n = int(rng.randint(2, 300))
        state = [int(v) for v in rng.randint(0, 1 << 32, size=n, dtype=np.uint64)] + [int(rng.randint(1 << 32, 1 << 62))]
        m = int(rng.randint(0, n))
        arr = container.pack(state, m, len(blocks), hh, ww)
        st, nb, a, b = container.unpack(arr)
        assert st == state[m:] and (nb, a, b) == (len(blocks), hh, ww) and not container.is_pack64(arr)
        subs = [[int(v) for v in rng.randint(0, 1 << 32, size=rng.randint(1, 9), dtype=np.uint64)] +
                [int(rng.randint(1 << 32, 1 << 62))] for _ in range(64)]
        mins = [int(rng.randint(0, len(s_))) for s_ in subs]
        arr = container.pack64(subs, mins, 7, 64, 96)
        got, nb, a, b = container.unpack64(arr)
        assert container.is_pack64(arr) and got == [s_[mm:] for s_, mm in zip(subs, mins)] and (nb, a, b) == (7, 64, 96)
        nch, world = int(rng.randint(1, 40)), int(rng.randint(1, 9))
        wts = [int(v) for v in rng.randint(1, 300, size=nch)]
        for weights in (None, wts):
            owners = [dist.shard_chains(nch, world, r, weights=weights) for r in range(world)]
            assert sorted(c for o in owners for c in o) == list(range(nch))
        loads = [sum(wts[c] for c in o) for o in owners]
        assert max(loads) <= (4 / 3) * (sum(wts) / world) + max(wts)


def test_kernel_support_predicates():
    """Host-side shape checks in front of bs_conv3_wino_f32 / bs_wino_gemm_f32 (no GPU involved): the model falls back
    to MIOpen / the BLAS library exactly where the kernels would answer BS_EUNSUPPORTED."""
    from bitswap_amd import hip
    assert hip.conv3_wino_supported(8, 16, 16) and hip.conv3_wino_supported(16, 32, 32) and hip.conv3_wino_supported(8, 8, 8)
    assert not hip.conv3_wino_supported(8, 64, 64)        # 256 tiles: more than a wavefront
    assert not hip.conv3_wino_supported(8, 16, 18)        # not a multiple of 4
    assert not hip.conv3_wino_supported(16, 8, 8)         # 16 images x 16 planes exceed the LDS
    U, V = torch.zeros(3, 32, 48), torch.zeros(3, 48, 100)
    assert hip.wino_gemm_supported(U, V)
    assert not hip.wino_gemm_supported(torch.zeros(3, 32, 40), torch.zeros(3, 40, 100))      # Cin % 16
    assert not hip.wino_gemm_supported(U, torch.zeros(3, 48, 102))                            # cols % 4
    assert not hip.wino_gemm_supported(U, V.transpose(1, 2).contiguous().transpose(1, 2))     # not contiguous
    assert not hip.wino_gemm_supported(U.double(), V.double())


def test_grouped_codec_enqueue_order():
    """GroupedCodec._round_robin (pure host logic): coding operations of the chain groups are enqueued alternately,
    group g starting g*skew operations late, every generator run to its end."""
    from bitswap_amd.codec import GroupedCodec
    gc = GroupedCodec.__new__(GroupedCodec)
    gc.group_streams = None
    log = []

    def ops(g, n):
        for k in range(n):
            log.append((g, k))
            yield k
    last = gc._round_robin([ops(0, 3), ops(1, 2), ops(2, 1)], skew=1)
    assert log == [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)] and last == [2, 1, 0]
    log.clear()
    gc._round_robin([ops(0, 2), ops(1, 2)], skew=0)
    assert log == [(0, 0), (1, 0), (0, 1), (1, 1)]


def test_stream_fingerprint_is_enforced_by_receivers(tmp_path):
    """ADVICE r2 / VERDICT r2 #5: stream_meta.json is read back by the receiver path and a format / CDF-spec / conv-route
    mismatch is refused loudly instead of decoding to garbage.  cli.decompress_streams is the receiver-only counterpart
    of the reference's inline receiver (mnist_compress.py:277-358): another codec object, streams from disk."""
    import json
    from bitswap_amd import meta
    ob = OracleBackend(O.MODE_DET)
    cli.compress(8, 2, 1, 0, dataset="mnist", experiments=3, ndatapoints=2, decompress=False, outdir=str(tmp_path),
                 backend=ob, small=8, verbose=False)
    out = cli.decompress_streams(8, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=ob, small=8, verbose=False)
    assert out.shape == (3, 2, 1024)
    # a receiver built with the other CDF specification (pixels K = 256 and latents K = 256 qualify for spec 2) refuses
    with pytest.raises(meta.StreamMismatch, match="cdf_spec"):
        cli.decompress_streams(8, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=ob, small=8, verbose=False,
                               cdf_spec=1)
    # a receiver that is not told a spec takes the one the record names: streams written with an older default (here: 2) decode
    cli.compress(8, 2, 0, 0, dataset="mnist", experiments=2, ndatapoints=2, decompress=False, outdir=str(tmp_path),
                 backend=OracleBackend(O.MODE_DET), small=8, verbose=False, cdf_spec=2)
    old = json.load(open(tmp_path / "bitstreams" / "mnist" / "nz2" / "BB-ANS" / "stream_meta.json"))
    assert meta.receiver_settings(old) == {"cdf_spec": 2, "gemm_arith": "fp32"} and meta.DEFAULT_CDF_SPEC != 2
    assert cli.decompress_streams(8, 2, 0, 0, dataset="mnist", outdir=str(tmp_path), backend=OracleBackend(O.MODE_DET), small=8,
                                  verbose=False).shape == (2, 2, 1024)
    # ... and so does any receiver if the record names another conv route or stream format
    mp = tmp_path / "bitstreams" / "mnist" / "nz2" / "Bit-Swap" / "stream_meta.json"
    good = json.load(open(mp))
    for key, val in (("conv_route", dict(good["conv_route"], rev=good["conv_route"]["rev"] + 1)), ("stream_format", "wave64"),
                     ("ansbits", 28)):
        json.dump(dict(good, **{key: val}), open(mp, "w"))
        with pytest.raises((meta.StreamMismatch, FileNotFoundError)):
            cli.decompress_streams(8, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=ob, small=8, verbose=False)
    json.dump(good, open(mp, "w"))
    cli.decompress_streams(8, 2, 1, 0, dataset="mnist", outdir=str(tmp_path), backend=ob, small=8, verbose=False)


def test_stream_set_records_its_draws_and_keeps_its_own_indices(tmp_path):
    """ADVICE r5 (medium + low): stream_meta.json says how the initial words were drawn and against how many test images
    (under the reference's order, mnist_compress.py:94,133-137,158, the words depend on both), a receiver with another test
    set is refused by name, streams written before the record existed still verify (both earlier conventions are tried), and
    the datapoint indices live next to the streams: a later run of another shape in the same outdir rewrites the reference's
    shared bitstreams/<ds>/indices.npy without stranding them."""
    import json
    from bitswap_amd import meta
    from bitswap_amd.codec import initial_states, reference_draws
    ob = OracleBackend(O.MODE_DET)
    kw = dict(dataset="mnist", outdir=str(tmp_path), backend=ob, small=8, verbose=False)
    cli.compress(8, 2, 1, 0, experiments=3, ndatapoints=2, decompress=False, **kw)
    sdir = tmp_path / "bitstreams" / "mnist" / "nz2" / "Bit-Swap"
    good = json.load(open(sdir / "stream_meta.json"))
    assert good["init_draws"] == "reference_order" and good["ntest"] == 512
    idx = np.load(sdir / "indices.npy")
    assert idx.shape == (3, 2) and np.array_equal(idx, reference_draws(512, 3, 2)[0])
    # the two conventions really are different words (else the record would be decoration)
    assert cli.experiment_draws(512, 3, 2)[1] != initial_states(3) and cli.experiment_draws(512, 3, 2, convention="seed_then_words")[1] == initial_states(3)
    assert cli.experiment_draws(4, 3, 2)[2] == "seed_then_words"
    # another run of another shape rewrites the shared index file; the first stream set still decodes
    cli.compress(8, 2, 0, 0, experiments=2, ndatapoints=3, decompress=False, **kw)
    assert np.load(tmp_path / "bitstreams" / "mnist" / "indices.npy").shape == (2, 3)
    assert cli.decompress_streams(8, 2, 1, 0, **kw).shape == (3, 2, 1024)
    # a stream set without the record (rounds 3-5) still verifies
    old = {k: v for k, v in good.items() if k not in ("init_draws", "ntest")}
    json.dump(old, open(sdir / "stream_meta.json", "w"))
    cli.decompress_streams(8, 2, 1, 0, **kw)
    # a receiver whose test set has another size is told so; an unknown convention is refused
    json.dump(dict(good, ntest=10000), open(sdir / "stream_meta.json", "w"))
    with pytest.raises(meta.StreamMismatch, match="10000 test images"):
        cli.decompress_streams(8, 2, 1, 0, **kw)
    json.dump(dict(good, init_draws="something_else"), open(sdir / "stream_meta.json", "w"))
    with pytest.raises(meta.StreamMismatch, match="something_else"):
        cli.decompress_streams(8, 2, 1, 0, **kw)
    # ... and the fingerprint word of the 64-state container does not move with the new informational fields
    assert meta.word(good) == meta.word(old)


def test_container_fingerprint_and_sidecar():
    """The 64-state container (version 2) carries the CRC-32 of the sender's fingerprint; version 1 files still read; the
    single-image receiver refuses a sidecar / header word it does not reproduce."""
    from bitswap_amd import meta
    from oracle.backend import Oracle64Backend
    ob = Oracle64Backend(O.MODE_DET)
    setup = cli.crop_setup(-1, nz=2, quantbits=6, backend=ob, small=8)
    rng = np.random.RandomState(3)
    blocks = tiling.extract_blocks(rng.randint(0, 256, (32, 64, 3)).astype(np.uint8))[0]
    res = cli.compress_images([blocks], quantbits=6, nz=2, setup=setup, backend=ob, fmt="wave64")
    (st, mins, _), fp = res[0], res.fingerprint
    assert fp["stream_format"] == "wave64" and fp["conv_route"]["rev"] == meta.ROUTE_REV
    arr = container.pack64(st, mins, len(blocks), 32, 64, fingerprint=meta.word(fp))
    assert container.fingerprint64(arr) == meta.word(fp) and int(arr[1]) == 2
    v1 = np.concatenate([arr[:1], np.array([1], dtype=np.uint32), arr[2:3], arr[4:]])
    assert container.is_pack64(v1) and container.fingerprint64(v1) is None
    assert container.unpack64(v1)[0] == container.unpack64(arr)[0]
    st2, nb, h, w = container.unpack64(arr)
    out, _ = cli.decompress_image(st2, nb, quantbits=6, nz=2, setup=setup, backend=ob, expect=fp,
                                  expect_word=container.fingerprint64(arr))
    assert np.array_equal(out, blocks)
    with pytest.raises(meta.StreamMismatch, match="fingerprint"):
        cli.decompress_image(st2, nb, quantbits=6, nz=2, setup=setup, backend=ob, expect_word=meta.word(fp) ^ 1)
    bad = json_roundtrip(dict(fp, cdf_spec={"z": [2, 2], "x": 2}))
    with pytest.raises(meta.StreamMismatch, match="cdf_spec.z"):
        cli.decompress_image(st2, nb, quantbits=6, nz=2, setup=setup, backend=ob, expect=bad)
    # informational fields do not take part
    assert meta.word(dict(fp, backend="hip-wave64", world_size=8)) == meta.word(fp)
    meta.check(json_roundtrip(dict(fp, backend="elsewhere", nblocks=2)), fp)


def json_roundtrip(d):
    import json
    return json.loads(json.dumps(d))


def test_lowrate_regime_codes_its_own_samples_at_a_trained_rate():
    """workload.calibrate_lowrate + lowrate_blocks: the calibrated synthetic model codes blocks drawn
    from its own generative path at a few bits/dim -- the regime of peaked tables and saturated tails the 26 bits/dim of
    plain random weights never reaches -- losslessly, through the same schedule code and the oracle."""
    model, zend, zcen = workload.build("imagenet4", "cpu", quantbits=10, small=16, regime="lowrate")
    plain, _, _ = workload.build("imagenet4", "cpu", quantbits=10, small=16)
    B, n = 3, 3
    imgs = workload.lowrate_blocks(model, B * n, seed=5).view(B, n, -1).to(torch.int32)
    assert imgs.min() >= 0 and imgs.max() <= 255 and imgs.float().std() > 2.0       # not a constant image
    with torch.no_grad():
        model.compress(True)
        mu, sc = model.infer(0)(given=imgs[:, 0].float())
        assert float(sc.max()) < 0.1001 and float(sc.min()) >= 0.1                    # the clamp of mnist_train.py:349
        mu, sc = model.generate(0)(given=torch.zeros(B, model.zdim_flat))
        assert 0.003 < float(sc.min()) and float(sc.max()) < 0.004
    codec = BitSwapCodec(model, zend, zcen, quantbits=10, backend=OracleBackend(O.MODE_DET, threads=8))
    st, met = codec.compress(imgs)
    assert 2.0 < met["nets"].mean() < 7.0, met["nets"].mean()
    # most of a pixel row is saturated tail: f = 1 (the reference's +1, mnist_compress.py:33) in > 90 % of the 256 bins
    xe = codec.xend.numpy()[:64]
    pm = O.logistic_pmf(xe, mu[0, :64].double().numpy(), sc.expand_as(mu)[0, :64].double().numpy(), O.MODE_DET)
    f, _, rc = O.tables(pm, 31, 8)
    assert rc == O.OK and (f == 1).mean() > 0.9
    out = codec.decompress(st, n)
    assert torch.equal(out, imgs) and st.to_lists() == initial_states(B)


def test_timeline_tool_classifies_wall_time(tmp_path):
    """tools/prof_summary.py timeline (the evidence behind DESIGN 3.7): a synthetic kernel trace with known overlaps comes
    out as the right split of wall time into idle / serial-only / one bulk kernel (+ serial) / two bulk kernels, and bulk
    calls are classified by their company; the 36- and 64-position GEMMs are told apart by the transform in front of them."""
    import csv
    import json
    import subprocess
    import sys
    ms = 1_000_000
    rows = [("void k_wino_fused<6, 6>(float)", 0, 1 * ms, 1), ("void k_wino_gemm<4, 2, 2>(float)", 1 * ms, 3 * ms, 1),
            ("void k_rans_pop_pivot<16, float, 8>(x)", 1 * ms, 3 * ms, 2),            # beside the first GEMM, all of it
            ("void k_wino_fused<8, 8>(float)", 3 * ms, 4 * ms, 1), ("void k_wino_gemm<4, 2, 2>(float)", 4 * ms, 6 * ms, 1),
            ("void k_logistic<16, float, 4, true>(y)", 5 * ms, 7 * ms, 2),            # shares 1 ms with the second GEMM
            ("k_rans_push(z)", 8 * ms, 9 * ms, 2)]                                     # serial only, after 1 ms of nothing
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"])
        for r in rows:
            w.writerow(r)
    out = tmp_path / "tl.json"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "prof_summary.py"), "timeline", str(d), str(out), "9", "0"],
                          stdout=subprocess.DEVNULL)
    res = json.load(open(out))
    wall = res["wall_ms"]
    assert abs(res["span_ms"] - 9.0) < 1e-9
    assert abs(wall["bulk_1"] - 4.0) < 1e-9 and abs(wall["bulk_1+serial"] - 2.0) < 1e-9 and abs(wall["bulk_2+"] - 1.0) < 1e-9
    assert abs(wall["idle"] - 1.0) < 1e-9 and abs(wall["serial_only"] - 1.0) < 1e-9
    calls = res["bulk_call_us"]
    g36 = [v for k, v in calls.items() if "[T=36]" in k][0]
    g64 = [v for k, v in calls.items() if "[T=64]" in k][0]
    assert g36["beside_serial"] == {"calls": 1, "mean_us": 2000.0} and g36["shared"]["calls"] == 0
    assert g64["shared"] == {"calls": 1, "mean_us": 2000.0} and g64["beside_serial"]["calls"] == 0


def test_overlap_stats_tool(tmp_path):
    """tools/overlap_stats.py (is the many-chain step bound by its bulk work or by the serial chain): union of the
    bulk / serial intervals over the window that ends with the last coding kernel."""
    import csv
    import subprocess
    import sys
    ms = 1_000_000
    rows = [("void k_wino_gemm<4, 2, 2, 2>(float)", 0, 4 * ms), ("void k_logistic<16, float, 4, true>(y)", 2 * ms, 5 * ms),
            ("void k_rans_pop_pivot<16, float, 4>(x)", 1 * ms, 3 * ms), ("k_rans_push(z)", 6 * ms, 7 * ms)]
    d = tmp_path / "trace" / "host"
    d.mkdir(parents=True)
    with open(d / "t_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"])
        for n, a, b in rows:
            w.writerow([n, a, b, 1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "overlap_stats.py"), str(tmp_path / "trace"), "--ms", "7"],
                                  text=True)
    head = out.splitlines()[0]
    # bulk union 0..5 = 5 ms, serial union 1..3 + 6..7 = 3 ms, anything 0..5 + 6..7 = 6 ms, only serial 1 ms, nothing 1 ms
    assert "some kernel 6.0 ms" in head and "a bulk kernel 5.0 ms" in head and "a serial kernel 3.0 ms" in head
    assert "only serial 1.0 ms" in head and "nothing 1.0 ms" in head
    assert "sum of bulk kernel time 7.0 ms, of serial kernel time 3.0 ms" in out


def test_bf16x3_async_fragment_loads_are_not_touched_before_their_wait(tmp_path):
    """bs_wino_gemm_bf16x3 requests its U fragments with inline-asm global loads and places the s_waitcnt by hand; the
    compiler treats the destination registers as defined from the asm statement on, so a copy or spill of them ahead of the wait
    would read stale data -- a property of the register allocation, not of the source.  Checked on the BUILT code: hipcc -S of
    the translation unit, then no instruction of the four product kernels (both tile shapes, 6 and 9 limb products) reads or
    writes an asynchronously loaded register before the next vmcnt wait (tools/isa_count.py::async_load_hazards)."""
    import importlib.util
    import subprocess
    from bitswap_amd import build
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "bitswap_amd", "csrc", "wino_gemm_bf16x3.hip")
    out = str(tmp_path / "x3.s")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run([build.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    spec = importlib.util.spec_from_file_location("isa_count", os.path.join(ROOT, "tools", "isa_count.py"))
    ic = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ic)
    lines = [l.strip() for l in open(out).read().split("\n")]
    for name in ("k_wino_gemm_bf16x3_o2ILi6ELb1ELb0ELb0ELb0ELb0E", "k_wino_gemm_bf16x3_o2ILi9ELb1ELb0ELb0ELb0ELb0E",
                 "k_wino_gemm_bf16x3ILi6ELi0ELb1ELb0ELb0E", "k_wino_gemm_bf16x3ILi9ELi0ELb1ELb0ELb0E"):
        res = ic.async_load_hazards(lines, name)
        assert res is not None, name
        nloads, hazards = res
        assert nloads >= 12 and not hazards, (name, nloads, hazards[:3])



def test_conv_epilogue_kernels_are_built_without_packed_float32_operations(tmp_path):
    """Round 5, visits v-z: the one wrong result of the bf16x3 hunt was a v_pk_add_f32 of k_wino_fused<6,6> -- hipcc's SLP
    vectorizer packs the transforms' additions -- that lost its result in lanes 48..63 beside bf16 MFMA wavefronts.  net_epilogue.hip
    is therefore compiled with -fno-slp-vectorize (bitswap_amd/build.py::FILE_FLAGS); checked on the BUILT code: with the flags the
    product build uses, the translation unit holds no packed float32 arithmetic at all."""
    import subprocess
    from bitswap_amd import build
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "bitswap_amd", "csrc", "net_epilogue.hip")
    assert "-fno-slp-vectorize" in build.FILE_FLAGS["net_epilogue.hip"]
    out = str(tmp_path / "net.s")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + build.FILE_FLAGS["net_epilogue.hip"]
    r = subprocess.run([build.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    packed = [l.strip() for l in open(out) if l.strip().startswith(("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"))]
    assert not packed, packed[:5]
    # the source-level half of the guard: a build system that does not know about the flag gets an error, not the
    # packed code
    bare = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run([build.hipcc_path()] + bare + ["-fsyntax-only", "--cuda-device-only", src], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "-fno-slp-vectorize" in r.stderr


def test_library_built_with_other_flags_counts_as_stale(monkeypatch, tmp_path):
    """build.py keys the in-tree library by its flag set (global, BITSWAP_HIPCC_EXTRA and per-file flags -- round 5's
    -fno-slp-vectorize on net_epilogue.hip is a correctness flag): the stamp beside the library must match, a library named by hand
    (BITSWAP_HIP_LIB, diagnostics) is taken as it is."""
    from bitswap_amd import build
    lib = tmp_path / "libx.so"
    lib.write_bytes(b"\0")
    future = os.path.getmtime(build.DEV_HDR) + 10 ** 6
    os.utime(lib, (future, future))                       # newer than every source
    monkeypatch.setattr(build, "LIB", str(lib))
    monkeypatch.delenv("BITSWAP_HIP_LIB", raising=False)
    assert build.is_stale()                               # no stamp
    (tmp_path / "libx.so.flags").write_text(build._flag_hash() + "\n")
    assert not build.is_stale()
    monkeypatch.setitem(build.FILE_FLAGS, "net_epilogue.hip", [])          # the per-file flag dropped: another hash
    assert build.is_stale()
    monkeypatch.setenv("BITSWAP_HIP_LIB", str(lib))
    assert not build.is_stale()


def test_shipped_library_holds_no_packed_float32_arithmetic(tmp_path):
    """The same property on what actually ships: every gfx950 code object inside the built libbitswap_hip.so (one per translation
    unit), disassembled -- no v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 anywhere (a packed addition beside bf16 MFMA
    wavefronts was the one wrong result of round 5; only net_epilogue.hip ever had any, and only it carries the flag)."""
    import importlib.util
    import re
    from bitswap_amd import build
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = build.build_hip()
    spec = importlib.util.spec_from_file_location("isa_count", os.path.join(ROOT, "tools", "isa_count.py"))
    ic = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ic)
    objs = ic.code_objects(lib)
    assert len(objs) == len(build.SRCS), [i for i, _ in objs]
    for k, (ident, blob) in enumerate(objs):
        text = ic.disassemble(blob, str(tmp_path), f"co{k}.o")
        assert text.count("\n") > 500, (k, ident)                      # it did disassemble something
        packed = re.findall(r"\bv_pk_(?:add|mul|fma)_f32[^\n]*", text)
        assert not packed, (k, packed[:3])


def test_repro_tool_embedded_programs_parse():
    """tools/bf16x3_repro.py carries the programs of its legs as strings run in child processes on the GPU box; a syntax slip in one
    would only show there."""
    import ast
    import re
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(ROOT, "tools", "bf16x3_repro.py")).read()
    ast.parse(src)
    names = re.findall(r'^([A-Z_]+_CODE) = r"""', src, re.M)
    assert {"TRAIL_CODE", "RECORD_CODE", "STACKS_CODE"} <= set(names), names
    for n in names:
        ast.parse(re.search(n + r' = r"""(.*?)"""', src, re.S).group(1))


def test_bench_headline_line_is_compact_and_complete():
    """The line the driver parses (bench.py: LAST stdout line): below 8 KB -- the driver keeps 8 KB of stdout; round 5's 26 KB
    line reached it cut in two and BENCH_r05.json has `parsed: null` -- with the contract's keys, a `roofline` and a
    `cpu_baseline` object, ONE copy of the measured shapes.  Input: a full record as bench.py assembles it (the committed
    round-5 record with all twenty sub-results: the worst case)."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module_headline", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.loads(open(os.path.join(root, "profiles", "r05E_bench.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000 and len(full["extra"]) == 20
    line = bench.headline_line(full, os.path.join(root, "gpurun_out", "bench_full.json"))
    assert "\n" not in line and len(line) < bench.HEADLINE_MAX_BYTES < 8192
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "lossless", "bits_per_dim"):
        assert k in h, k
    assert h["value"] == full["value"] and h["ms_per_step"] == full["ms_per_step"] and h["vs_baseline"] is None
    assert set(h["config"]) >= {"workload", "chains_per_gpu", "chain_groups", "cdf_spec", "conv_dtype", "stream_format"}
    r = h["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] <= 1 and r["traffic"] > 0
    assert r["kernel"]["avg_launch_ms"] > 0 and 0 < r["kernel"]["valu_issue_frac"] <= 1
    assert r["mfma"]["unit"] == "TFLOP/s" and 0 < r["mfma"]["frac"] <= 1
    c = h["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"] and c["reference_python"]["value"] > 0
    assert len(h["measured_shapes"]) == 20 and "extra" not in h and "summary" not in h and h["schema"] == bench.BENCH_SCHEMA
    # a record without sub-results or a roofline (a strong-scaling child, a failed CPU baseline) still makes a line
    bare = dict(full, roofline=None, cpu_baseline=None, extra=None, stream_gather=None)
    bare["config"] = dict(full["config"], measured_shapes=None, predicted_scaling=None)
    hb = json.loads(bench.headline_line(bare))
    assert hb["roofline"] is None and hb["cpu_baseline"] is None and hb["measured_shapes"] is None and hb["value"] == full["value"]


def test_seeded_full_models_are_the_reference_models():
    """conftest.seeded_full_model (the weights of the chain_<data>_full fixtures rebuilt from their seed, for the GPU tests that
    run the conv stacks at the BASELINE models' real width) against the imported reference Model built the way
    tests/golden/make_golden.py::_full_chain builds it: every tensor identical.  Needs the reference checkout (skipped on the GPU
    box; the GPU tests additionally hold the resulting (mu, scale) to the fixtures')."""
    import os
    import subprocess
    import sys
    ref = os.environ.get("BITSWAP_REFERENCE", "/root/reference")
    if not os.path.isdir(ref):
        pytest.skip("reference not present on this host")
    here = os.path.dirname(os.path.abspath(__file__))
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import make_golden as mg
from conftest import seeded_full_model, FULL_CHAIN_SEEDS
for data, (xs, nz, zch, rw) in {"cifar": ((3, 32, 32), 8, 8, 252), "mnist": ((1, 32, 32), 2, 1, 63)}.items():
    torch.manual_seed(FULL_CHAIN_SEEDS[data])
    ref = mg.RefModel(xs=xs, nz=nz, zchannels=zch, nprocessing=4, kernel_size=3, resdepth=8, reswidth=rw, root_process=False)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith(".b") or n.endswith("gen_std"):
                p.add_(torch.randn_like(p) * 0.3)
            if n.endswith(".gain"):
                p.add_(torch.randn_like(p) * 0.2)
    g = np.load(%r + "/golden/chain_" + data + "_full_bitswap.npz")
    ours = seeded_full_model(g, data)
    a, b = ref.state_dict(), ours.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), data
print("same")
""" % (os.path.join(here, "golden"), here, os.path.dirname(here), here)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("same"), out.stderr[-2000:]
