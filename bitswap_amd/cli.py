"""Command-line drivers mirroring the reference's scripts.

`<dataset>_compress.py --gpu G --nz N --quantbits Q --bitswap S` (mnist_compress.py:368-386 and
its cifar/imagenet siblings) run `experiments` x `ndatapoints` images (100 x 100 in the reference,
:102-103) and write the same artefacts: per-experiment pickled bitstreams under
`bitstreams/<ds>/nz<N>/<Scheme>/` (:265-267) and the four metric arrays under `plots/<ds><N>/`
(:363-366).  Internals are new: the 100 experiments are 100 chains coded in lock-step on the GPU
(codec.BitSwapCodec), sharded over ranks when launched under torchrun.

Offline there are no datasets or checkpoints: `--data file.npy` supplies uint8 images
[N,H,W,C] / [N,C,H,W], `--params file` a reference checkpoint, `--synthetic` uses the seeded
synthetic stand-ins of bitswap_amd.workload.
"""
import argparse
import os
import random
import time

import numpy as np
import torch

from . import container, dist, meta, tiling, workload
from .bins import discretize, _cache_names as _bins_cache_names
from .codec import BitSwapCodec, initial_states, reference_draws
from .model import elbo_bits, preset

SCHEME = {1: "Bit-Swap", 0: "BB-ANS"}
TITLE = {"mnist": "MNIST", "cifar": "CIFAR-10", "imagenet": "ImageNet (32x32)", "imagenetcrop": "ImageNet (unscaled)"}


def seed_everything():
    """mnist_compress.py:94-99"""
    np.random.seed(100)
    random.seed(50)
    torch.manual_seed(50)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(50)
    # cudnn.deterministic / benchmark (:98-99) are scoped to the codec's conv stacks: codec.deterministic_convs()


def experiment_draws(ntest, experiments, ndatapoints, nwords=10000, convention=None):
    """(randindices, initial states, convention) of a dataset script run.  convention "reference_order": the reference's numpy
    draw order (codec.reference_draws: seed, choice(replace=False), then the words experiment by experiment -- the words depend
    on `ntest`, the permutation consumes the generator); "seed_then_words": sampling with replacement + codec.initial_states()
    -- the fallback for shapes the reference's own sequence cannot serve (fewer test images than experiments x ndatapoints)
    and what every stream written before round 5 used.  convention=None picks by shape; a receiver passes the one recorded in
    stream_meta.json ("init_draws")."""
    if convention is None:
        convention = "reference_order" if ntest >= experiments * ndatapoints else "seed_then_words"
    if convention == "reference_order":
        return reference_draws(ntest, experiments, ndatapoints, nwords, seed=100) + (convention,)
    if convention != "seed_then_words":
        raise meta.StreamMismatch(f"unknown initial-word convention {convention!r} in stream_meta.json")
    np.random.seed(100)
    randindices = np.random.choice(ntest, size=(experiments, ndatapoints), replace=ntest < experiments * ndatapoints)
    return randindices, initial_states(experiments, nwords, seed=100), convention


def load_images(dataset, path, synthetic, xs, n):
    """uint8 images as flat CHW rows [N, C*H*W]."""
    if path:
        a = np.load(path)
        if a.ndim == 3:
            a = a[:, None]
        if a.shape[-1] in (1, 3) and a.shape[1] not in (1, 3):
            a = a.transpose(0, 3, 1, 2)
        if dataset == "mnist" and a.shape[-1] == 28:       # transforms.Pad(2), mnist_compress.py:128
            a = np.pad(a, ((0, 0), (0, 0), (2, 2), (2, 2)))
        return torch.from_numpy(np.ascontiguousarray(a)).view(a.shape[0], -1)
    if synthetic:
        return workload.synthetic_blocks(n, xs, seed=100)
    raise FileNotFoundError(
        f"no test images: the {dataset} dataset cannot be downloaded here -- pass --data <uint8 .npy> or --synthetic")


def load_model(dataset, nz, device, params, synthetic, nn_batch=None):
    path = params or f"model/params/{dataset}/nz{nz}"
    if os.path.exists(path):
        m = preset(dataset, nz, nn_batch=nn_batch)
        m.load_state_dict(torch.load(path, map_location="cpu"))
        m = m.to(device).eval().fold()
        return m.fuse() if torch.device(device).type == "cuda" else m
    if synthetic:
        return workload.synthetic_model(dataset, nz, device, nn_batch=nn_batch)
    raise FileNotFoundError(f"checkpoint {path} not found -- pass --params <file> or --synthetic")


def stream_dir(outdir, dataset, nz, bitswap):
    scheme = SCHEME[int(bool(bitswap))]
    return os.path.join(outdir, "bitstreams", dataset, f"nz{nz}", scheme), scheme


def stream_name(scheme, quantbits, nz, c, wave64=False):
    """Pickle name of experiment c (0-based), mnist_compress.py:265-267."""
    return f"{scheme}_{quantbits}bits_nz{nz}_experiment{c + 1}" + ("_wave64" if wave64 else "")


def compress(quantbits, nz, bitswap, gpu, dataset="mnist", experiments=100, ndatapoints=100, decompress=False,
             synthetic=False, data=None, params=None, outdir=".", backend=None, small=None, verbose=True,
             save_bins=False, fmt="reference", cdf_spec=meta.DEFAULT_CDF_SPEC):
    """One (dataset, nz, quantbits, scheme) experiment set.  Returns dict of the metric arrays on
    rank 0 (None on other ranks).  fmt "wave64": the opt-in 64-state stream format (pickles then hold 64 sub-state
    lists per experiment and carry the suffix _wave64)."""
    rank, world = dist.init()
    dev = torch.device("cpu") if backend is not None else (torch.device("cuda", gpu) if world == 1 else dist.local_device())
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    if verbose and rank == 0:
        print(f"{SCHEME[int(bool(bitswap))]} - {TITLE[dataset]} - {nz} latent layers - {quantbits} bits quantization")
    seed_everything()

    if small:
        model = workload.synthetic_model(dataset, nz, dev, small=small)
    else:
        model = load_model(dataset, nz, dev, params, synthetic)
    images = load_images(dataset, data, synthetic or bool(small), model.xs, max(experiments * ndatapoints, 512))
    bins_data = images[: min(len(images), 4096)].view((-1,) + tuple(model.xs))
    # bins fitted here come from the first test images, not from the training set the reference samples
    # (discretization.py:34-40): they are only written under the reference's cache names on request (--save-bins).
    # The cache is written by rank 0 only; the other ranks wait and load it (no concurrent writers)
    keep = bool(save_bins) and not (synthetic or small)
    if world > 1 and keep and rank != 0:
        dist.barrier()
    zend, zcen = discretize(nz, quantbits, torch.float64, dev, model, dataset, data=bins_data,
                            ppb=2 if (synthetic or small) else 30, cache_dir=os.path.join(outdir, "bins"),
                            save=keep and rank == 0)
    if world > 1 and keep and rank == 0:
        dist.barrier()

    # (experiments, ndatapoints) test images per experiment without replacement, then the initial words of every
    # experiment -- numpy's generator consumed in the reference's order (:94,133-137,158): seed(100), choice(), randint() per
    # experiment.  (Round 4 re-seeded before the words; experiment i then started from other "random" words than it does
    # in the reference script.)  A saved index file selects the images, as it would in the reference if its guard ever hit,
    # but the draw still happens.  Fewer images than experiments x ndatapoints (small synthetic sets): the reference's
    # choice(replace=False) would raise; documented fallback = sampling with replacement + initial_states().
    idx_path = os.path.join(outdir, "bitstreams", dataset, "indices.npy")
    randindices, inits, draws = experiment_draws(len(images), experiments, ndatapoints)
    have = os.path.exists(idx_path)
    if world > 1:
        dist.barrier()       # every rank has looked before rank 0 may write
    saved = np.load(idx_path) if have else None
    if saved is not None and saved.shape == randindices.shape:
        randindices = saved
    elif rank == 0:
        os.makedirs(os.path.dirname(idx_path), exist_ok=True)
        np.save(idx_path, randindices)
    if world > 1:
        dist.barrier()       # nobody may race ahead and find rank 0's half-written indices file on a later call

    mine = dist.shard_chains(experiments, world, rank)
    codec = BitSwapCodec(model, zend, zcen, quantbits=quantbits, bitswap=bool(bitswap),
                         backend=_format_backend(fmt, backend, dev), cdf_spec=cdf_spec)
    wave64 = hasattr(codec.backend.new_state([inits[0]], 16), "len64") if fmt == "wave64" else False
    x = images[torch.from_numpy(randindices[mine].reshape(-1))].view(len(mine), ndatapoints, -1).to(torch.int32)
    state = codec.new_states(len(mine), ndatapoints, states=[inits[c] for c in mine])

    t0 = time.perf_counter()
    state, met = codec.compress(x.to(dev), state=state)
    t_send = time.perf_counter() - t0
    elbos = np.zeros((len(mine), ndatapoints))
    for xi in range(ndatapoints):                                # ELBO metric (:170-174), off the coding path
        xb = x[:, xi].to(dev).float().view((-1,) + tuple(model.xs))
        elbos[:, xi] = (elbo_bits(model, xb) / model.xdim).cpu().numpy()
    sent = state.to_lists()

    sdir, scheme = stream_dir(outdir, dataset, nz, bitswap)
    os.makedirs(sdir, exist_ok=True)
    for c, s in zip(mine, sent):
        container.save_state(os.path.join(sdir, stream_name(scheme, quantbits, nz, c, wave64)), s)
    # what a receiver must reproduce for these streams to decode (bitswap_amd/meta.py); read back and enforced by
    # decompress_streams() and by the --decompress leg below
    cpc = len(mine)
    if world > 1:       # every rank's shard size (they differ when the experiments do not divide): one record for all ranks
        per_rank = [int(round(v)) for v in dist.allreduce_sum([float(len(mine)) if r == rank else 0.0 for r in range(world)])]
        cpc = per_rank[0] if len(set(per_rank)) == 1 else per_rank
    fp = meta.fingerprint(codec, chains_per_call=cpc)
    if rank == 0:
        # ... and what its end-condition checks need: how the initial words were drawn (they depend on the test-set size under
        # the reference's order) and WHICH datapoints these streams hold -- next to the streams, because the reference's shared
        # bitstreams/<ds>/indices.npy is rewritten by any later run of another shape in the same outdir
        meta.save(os.path.join(sdir, "stream_meta.json"), fp, world_size=world, experiments=experiments,
                  ndatapoints=ndatapoints, init_draws=draws, ntest=int(len(images)))
        np.save(os.path.join(sdir, "indices.npy"), randindices)
    if world > 1:
        dist.barrier()

    t_recv = 0.0
    if decompress:
        meta.check(meta.load(os.path.join(sdir, "stream_meta.json")), fp, f"{sdir}/stream_meta.json")
        t0 = time.perf_counter()
        out = codec.decompress(state, ndatapoints)
        t_recv = time.perf_counter() - t0
        assert torch.equal(out.cpu(), x), "decoded datapoint does not match"            # (:319,354)
        want = [inits[c] for c in mine]
        if wave64:
            from .hip import split_state
            want = [split_state(s) for s in want]
        assert state.to_lists() == want, "initial state not restored"                       # (:358)

    rows = {k: dist.gather_rows(v, mine, experiments) for k, v in
            dict(nets=met["nets"], elbos=elbos, cmas=met["cma"], total=met["total"]).items()}
    packer = (lambda s: container.pack64(s, [0] * 64, ndatapoints, 32, 32)[:-3]) if wave64 else \
             (lambda s: container.pack(s, 0, ndatapoints, 32, 32)[:-3])
    words = dist.gather_streams([packer(s) for s in sent], mine, experiments)
    tot = dist.allreduce_sum([float(met["total"][:, -1].sum()), float(len(mine) * ndatapoints * model.xdim),
                              t_send + t_recv])
    if rank != 0:
        return None
    nets, el = rows["nets"], rows["elbos"]
    if verbose:
        print(f"N:{nets.mean():.4f}±{nets.std():.2f}, E:{el.mean():.4f}±{el.std():.2f}, D:{nets.mean() - el.mean():.6f}")
        px = experiments * ndatapoints * 1024
        print(f"sender {t_send:.2f}s" + (f", receiver {t_recv:.2f}s, lossless, {px / (t_send + t_recv):.0f} pixels/s (enc+dec)"
                                        if decompress else f", {px / t_send:.0f} pixels/s (enc)") +
              f", {world} GPU(s), {sum(len(w) for w in words) * 4} bytes gathered")
    pdir = os.path.join(outdir, "plots", f"{dataset}{nz}")
    os.makedirs(pdir, exist_ok=True)
    tag = "bitswap" if bitswap else "bbans"
    for k, v in rows.items():
        np.save(os.path.join(pdir, f"{tag}_{quantbits}bits_{k}"), v)          # (:363-366)
    rows["bits_per_dim"] = tot[0] / tot[1]
    return rows


def decompress_streams(quantbits, nz, bitswap, gpu, dataset="mnist", synthetic=False, data=None, params=None,
                       outdir=".", backend=None, small=None, verbose=True, cdf_spec=None):
    """Receiver only (the reference decodes inside compress(), mnist_compress.py:277-358; a real receiver is another
    process): load the experiment pickles and stream_meta.json a sender wrote under `outdir`, REFUSE to decode unless this
    receiver reproduces the recorded format / CDF specification / conv route, decode every experiment, and assert the
    reference's two end conditions (every datapoint matches, :319,354; the initial state is restored, :358).
    Returns the decoded images [experiments, ndatapoints, X] on rank 0 (single process: streams are read from disk)."""
    dev = torch.device("cpu") if backend is not None else torch.device("cuda", gpu)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    seed_everything()
    sdir, scheme = stream_dir(outdir, dataset, nz, bitswap)
    written = meta.load(os.path.join(sdir, "stream_meta.json"))
    fmt = written.get("stream_format", "reference")
    # the receiver takes the CDF spec and the conv arithmetic the record names (a sender's defaults may have moved on since the
    # stream was written); an explicit cdf_spec argument overrides -- and is then refused by meta.check if it does not match
    if cdf_spec is None:
        cdf_spec = meta.receiver_settings(written)["cdf_spec"]
    experiments, ndatapoints = int(written["experiments"]), int(written["ndatapoints"])
    model = workload.synthetic_model(dataset, nz, dev, small=small) if small else load_model(dataset, nz, dev, params, synthetic)
    images = load_images(dataset, data, synthetic or bool(small), model.xs, max(experiments * ndatapoints, 512))
    bins_data = images[: min(len(images), 4096)].view((-1,) + tuple(model.xs))
    zend, zcen = discretize(nz, quantbits, torch.float64, dev, model, dataset, data=bins_data,
                            ppb=2 if (synthetic or small) else 30, cache_dir=os.path.join(outdir, "bins"), save=False)
    meta.adopt_route(model, written)
    codec = BitSwapCodec(model, zend, zcen, quantbits=quantbits, bitswap=bool(bitswap),
                         backend=_format_backend(fmt, backend, dev), cdf_spec=cdf_spec)
    # Which chains were coded together.  A batch-invariant conv route (the GPU's) does not care: everything in one call.
    # Otherwise the streams of a multi-process sender decode only in the sender's own shards (dist.shard_chains is a pure
    # function of the counts the record holds): this single process then plays the ranks one after the other.
    mine = meta.fingerprint(codec, chains_per_call=experiments)
    shards = [list(range(experiments))]
    sender_world = int(written.get("world_size", 1))
    if not meta.batch_invariant(mine) and sender_world > 1:
        shards = [dist.shard_chains(experiments, sender_world, r) for r in range(sender_world)]
        sizes = [len(sh) for sh in shards]
        mine = meta.fingerprint(codec, chains_per_call=sizes[0] if len(set(sizes)) == 1 else sizes)
        shards = [sh for sh in shards if sh]
    meta.check(written, mine, f"{sdir}/stream_meta.json")
    wave64 = fmt == "wave64" and backend is None
    states = [container.load_state(os.path.join(sdir, stream_name(scheme, quantbits, nz, c, wave64)))
              for c in range(experiments)]
    nwords = max((sum(len(x) for x in s) if wave64 else len(s)) for s in states)
    # the sender's draw sequence: recorded since round 6; streams of rounds 3-5 carry no record -- round 5's followed the
    # reference's order whenever the shape allowed it, rounds 3-4 seeded and drew all words at once: try the first, accept
    # the second (the end condition below tells which one it was)
    recorded = written.get("init_draws")
    if recorded is not None and "ntest" in written and int(written["ntest"]) != len(images):
        raise meta.StreamMismatch(f"{sdir}/stream_meta.json: the streams were written against {written['ntest']} test images, this "
                                  f"receiver has {len(images)} (initial words and datapoint indices depend on that number)")
    _, inits, _ = experiment_draws(len(images), experiments, ndatapoints, convention=recorded)
    alt_inits = None if recorded is not None else initial_states(experiments, 10000, seed=100)
    if wave64:
        from .hip import split_state
        inits = [split_state(s) for s in inits]
        alt_inits = None if alt_inits is None else [split_state(s) for s in alt_inits]
    out = torch.zeros((experiments, ndatapoints, model.xdim), dtype=torch.int32)
    for sh in shards:
        state = codec.backend.new_state([states[c] for c in sh], nwords + ndatapoints * (model.xdim + 64) + 4 * model.zdim_flat)
        out[sh] = codec.decompress(state, ndatapoints).cpu().to(torch.int32)
        got = state.to_lists()
        assert got == [inits[c] for c in sh] or (alt_inits is not None and got == [alt_inits[c] for c in sh]), \
            "initial state not restored"                                                   # (:358)
    own_idx = os.path.join(sdir, "indices.npy")             # the stream set's own copy (round 6); else the shared file
    randindices = np.load(own_idx if os.path.exists(own_idx) else os.path.join(outdir, "bitstreams", dataset, "indices.npy"))
    want = images[torch.from_numpy(randindices.reshape(-1))].view(experiments, ndatapoints, -1).to(torch.int32)
    assert torch.equal(out, want), "decoded datapoint does not match"                     # (:319,354)
    if verbose:
        print(f"decoded {experiments} x {ndatapoints} datapoints from {sdir}: lossless, initial states restored")
    return out


def dataset_main(dataset, default_nz, nz_loop=None):
    """argparse front end shared by the four <dataset>_compress.py scripts (flags :369-373)."""
    p = argparse.ArgumentParser()
    p.add_argument('--gpu', default=0, type=int)
    p.add_argument('--nz', default=default_nz, type=int)
    p.add_argument('--quantbits', default=10, type=int)
    p.add_argument('--bitswap', default=1, type=int)
    # extras (not in the reference)
    p.add_argument('--decompress', default=0, type=int, help="also run the receiver and assert losslessness")
    p.add_argument('--experiments', default=100, type=int)
    p.add_argument('--ndatapoints', default=100, type=int)
    p.add_argument('--synthetic', action='store_true', help="seeded synthetic weights/images (no datasets offline)")
    p.add_argument('--data', default=None, help="uint8 .npy test images")
    p.add_argument('--params', default=None, help="reference checkpoint (state_dict)")
    p.add_argument('--outdir', default=".")
    p.add_argument('--format', default="reference", choices=["reference", "wave64"],
                   help="stream format: the reference's single-state stream, or the opt-in 64-state format")
    p.add_argument('--decompress-only', action='store_true',
                   help="receiver only: decode the pickles a previous run wrote under --outdir (checks stream_meta.json first)")
    p.add_argument('--cdf-spec', default=None, type=int, choices=list(meta.CDF_SPECS),
                   help="deterministic CDF specification (include/bitswap_hip.h) of the tables whose bins are uniform.  Senders: "
                        "default 4 (round 6: blocks of 8 bins, one reciprocal + a Newton correction per quotient); 3 = round 5, "
                        "2 = rounds 3-4, 1 = one sigmoid per endpoint everywhere (rounds 1-2).  Receivers take the spec the "
                        "stream's record names unless this flag overrides it")
    p.add_argument('--save-bins', action='store_true',
                   help="write bins fitted on the given test images under the reference's cache names (bins/*.pt)")
    args = p.parse_args()
    print(args)
    for nz in (nz_loop or [args.nz]):      # imagenet_compress.py:382 ignores --nz and runs [2, 4]
        if args.decompress_only:
            decompress_streams(args.quantbits, nz, args.bitswap, args.gpu, dataset=dataset, synthetic=args.synthetic,
                               data=args.data, params=args.params, outdir=args.outdir, cdf_spec=args.cdf_spec)
            continue
        compress(args.quantbits, nz, args.bitswap, args.gpu, dataset=dataset, experiments=args.experiments,
                 ndatapoints=args.ndatapoints, decompress=bool(args.decompress), synthetic=args.synthetic,
                 data=args.data, params=args.params, outdir=args.outdir, save_bins=args.save_bins, fmt=args.format,
                 cdf_spec=args.cdf_spec)


# ---------------------------------------------------------------------------------------------
# single-image path: imagenetcrop_compress.compress / demo_compress.compress / demo_decompress.decompress
# ---------------------------------------------------------------------------------------------
def crop_setup(gpu, nz=4, quantbits=10, synthetic=False, params=None, outdir=".", backend=None, small=None,
               nn_batch=None):
    """Model + bins of the crop/demo scripts.  nn_batch: the conv stacks always run on micro-batches of
    exactly this many blocks (zero padded), so a block's (mu, scale) bits do not depend on which other
    images are coded next to it -- an image compressed in a lock-step batch of many can be decompressed
    on its own (demo_decompress.py) and vice versa.  32: enough columns for the Winograd-domain GEMM
    path (4.2 vs 2.3 Mpixel/s for 100 images on one GPU), 0.4 s to decode a single 80-block image."""
    if nn_batch is None:
        nn_batch = int(os.environ.get("BITSWAP_CROP_NN_BATCH", "32"))
    rank, world = dist.init() if backend is None else (0, 1)
    if backend is not None:
        dev = torch.device("cpu")
    else:
        # one process per GPU: under torchrun the device is this rank's LOCAL_RANK, not the --gpu flag; the HIP
        # kernels are enqueued on the CURRENT device's stream, so it has to be selected before anything runs
        dev = dist.local_device() if world > 1 else torch.device("cuda", max(gpu, 0))
        torch.cuda.set_device(dev)
    if small:
        model = workload.synthetic_model("imagenetcrop", nz, dev, small=small, nn_batch=nn_batch)
    else:
        model = load_model("imagenetcrop", nz, dev, params, synthetic, nn_batch=nn_batch)
    data = workload.synthetic_blocks(512, model.xs, seed=100).view((-1,) + tuple(model.xs))
    # bins fitted on synthetic blocks are never written under the reference's cache names (they would silently
    # replace properly fitted bins of a real checkpoint); a real checkpoint needs its cached bins or real data
    cache = os.path.join(outdir, "bins")
    have = all(os.path.exists(f) for f in _bins_cache_names(cache, "imagenetcrop", nz, quantbits))
    if not (synthetic or small or have):
        raise FileNotFoundError(f"no cached bins under {cache} for imagenetcrop nz{nz}: fit them with "
                                "bitswap_amd.bins.discretize(..., data=<training images>) or pass --synthetic")
    zend, zcen = discretize(nz, quantbits, torch.float64, dev, model, "imagenetcrop", data=data, ppb=2,
                            cache_dir=cache, save=False)
    return model, zend, zcen, dev


def _format_backend(fmt, backend, dev):
    """fmt "reference": the reference's single-state stream (HipBackend unless a backend is injected);
    "wave64": the opt-in 64-state format (Hip64Backend)."""
    if fmt not in ("reference", "wave64"):
        raise ValueError(f"unknown stream format {fmt!r}")
    if backend is None and fmt == "wave64":
        from .codec import Hip64Backend
        return Hip64Backend(dev)
    return backend


class ImageStreams(list):
    """compress_images() result: per image (state list, min_words, bits/dim); `.fingerprint` is the record a receiver
    must reproduce (bitswap_amd/meta.py) -- demo_compress.py writes it next to the container."""
    fingerprint = None


def compress_images(images_blocks, quantbits=10, nz=4, bitswap=1, gpu=0, hwc_quirk=False, setup=None, backend=None,
                    trim=True, fmt="reference", cdf_spec=meta.DEFAULT_CDF_SPEC):
    """images_blocks: list of [n_i, 32, 32, 3] uint8 block arrays (one per image; every image is a
    chain, imagenetcrop_compress.py:279-300).  Chains of different length run in lock-step and
    drop out as they finish.  Returns per image (state list, min_words, bits/dim); in the 64-state format the state is
    a list of 64 sub-state lists and min_words a list of 64."""
    results = ImageStreams()
    if len(images_blocks) == 0:          # a rank that owns no image (more ranks than images)
        return results
    model, zend, zcen, dev = setup
    flat = [torch.from_numpy((tiling.blocks_to_hwc_flat(b) if hwc_quirk else tiling.blocks_to_chw_flat(b)).astype(np.int32))
            for b in images_blocks]
    codec = BitSwapCodec(model, zend, zcen, quantbits=quantbits, bitswap=bool(bitswap),
                         backend=_format_backend(fmt, backend, dev), cdf_spec=cdf_spec)
    results.fingerprint = meta.fingerprint(codec, chains_per_call=len(flat))
    np.random.seed(100)   # every image starts from the same 'random' stack (imagenetcrop_compress.py:249,122)
    nmax = max(len(f) for f in flat)
    one = initial_states(1, 10000, seed=100)[0]
    state = codec.new_states(len(flat), nmax, states=[list(one) for _ in flat])
    state.min_len = (state.len64 if hasattr(state, "len64") else state.len).clone()
    state, order, met = codec.compress_ragged(flat, state=state)
    mins = state.min_len.cpu().tolist()
    lists = state.to_lists()
    results.extend([None] * len(flat))
    for k, i in enumerate(order):
        m = mins[k]
        m = ([int(v) if trim else 0 for v in m] if isinstance(m, list) else (int(m) if trim else 0))
        results[i] = (lists[k], m, float(met["cma"][k]))
    return results


def decompress_image(state, nblocks, quantbits=10, nz=4, gpu=0, setup=None, backend=None, hwc_quirk=False,
                     expect=None, expect_word=None, cdf_spec=None):
    """demo_decompress.decompress (:69-148): -> [nblocks, 32, 32, 3] uint8 blocks.  A state that is a list of 64
    sub-state lists (container.unpack64) is decoded in the 64-state format.  expect: the sender's fingerprint record
    (the container's sidecar) / expect_word: its CRC-32 (64-state container header): decoding is REFUSED
    (meta.StreamMismatch) unless this receiver's codec reproduces it."""
    model, zend, zcen, dev = setup
    fmt = "wave64" if isinstance(state[0], (list, tuple)) else "reference"
    if expect is not None:       # configure from the sender's record what a receiver can adopt (meta.receiver_settings)
        if cdf_spec is None:
            cdf_spec = meta.receiver_settings(expect)["cdf_spec"]
        meta.adopt_route(model, expect)
    codec = BitSwapCodec(model, zend, zcen, quantbits=quantbits, bitswap=True, backend=_format_backend(fmt, backend, dev),
                         cdf_spec=cdf_spec)
    mine = meta.fingerprint(codec, chains_per_call=1)
    if expect is not None:
        meta.check(expect, mine, "container")
    if expect_word is not None and int(expect_word) != 0 and int(expect_word) != meta.word(mine) and expect is None:
        raise meta.StreamMismatch(f"container fingerprint {int(expect_word):#010x} != this receiver's {meta.word(mine):#010x}: "
                                  "it was written with another CDF specification / conv route / library revision "
                                  "(the sidecar <name>_bitswap.meta.json names the fields)")
    nwords = sum(len(s) for s in state) if fmt == "wave64" else len(state)
    st = codec.backend.new_state([list(state)], nwords + nblocks * (model.xdim + 64) + 4 * model.zdim_flat)
    out = codec.decompress(st, nblocks)[0].cpu().numpy()
    if hwc_quirk:
        return out.reshape(nblocks, 32, 32, 3).astype(np.uint8), st.to_lists()[0]
    return tiling.chw_flat_to_blocks(out), st.to_lists()[0]
