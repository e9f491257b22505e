"""Hierarchical-VAE `Model` for the compression path: PyTorch-ROCm modules whose convolutions -- the only MFMA-shaped work on
the path -- run, fused, as Winograd-domain batched GEMMs on our own MFMA kernels between HIP transform kernels (no library call
in the compress path since round 3; the MIOpen / BLAS routes that remain below are env-gated fallbacks for experiments).

Mirrors the class surface of the reference's Model (model/mnist_train.py:17-438; the cifar /
imagenet variants are identical, imagenetcrop_train.py:306-315,417 makes gen_std a conv):
same constructor arguments, same `compress()/infer(i)/generate(i)/loss()` methods and the same
state-dict keys, so reference checkpoints (`model/params/<ds>/nz<k>`) load unchanged.

What is different, MI355X-first:
  * no tensorboard logger is created at construction (reference :54-61 side effect);
  * weight normalisation w = v * g / (||v|| + 1e-10) (utils/torch/modules.py:98-106) is folded
    ONCE into a cached conv weight (`fold()`), not recomputed on every forward;
  * compress mode accepts a whole batch of chains [B, D] and returns float32 [B, D]
    (the reference asserts batch == 1, :372,431, and round-trips through float64);
    1-D inputs keep the reference behaviour (output dtype = input dtype);
  * `nn_batch`: convs always run on micro-batches of one fixed shape (zero padded), so a
    sample's (mu, scale) bits do not depend on how many chains are coded together -- the
    decoder must reproduce the encoder's parameters bit for bit.
  * `fuse()`: on a HIP device in compress mode the pointwise work between the convolutions (bias,
    ELU, residual add, the scale heads) runs as ONE launch per convolution
    (bitswap_amd/csrc/net_epilogue.hip) and the mu/std head pair is a single convolution with
    stacked filters.  The 3x3 / 5x5 ResNet and head convolutions
    themselves run in the Winograd domain (`conv_algo = "winograd"`, bitswap_amd/winograd.py): HIP transform
    kernels around one batched GEMM per convolution.  Sender and receiver must both use the same route: the
    parameters differ from the unfused path in the last float32 bits.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

_SMALL = 1e-10  # utils/torch/modules.py:14


def softplus(x):
    """-logsigmoid(-x), the reference's numerically stable softplus (modules.py:112-114)."""
    return -F.logsigmoid(-x)


class WnConv2d(nn.Module):
    """Weight-normalised conv (modules.py:57-109): parameters v, gain, b with identical names."""

    def __init__(self, in_dim, out_dim, kernel_size, stride, padding, init_scale=1.0, loggain=True, bias=True):
        super().__init__()
        self.in_dim, self.out_dim, self.kernel_size = in_dim, out_dim, kernel_size
        self.stride, self.padding, self.init_scale, self.loggain = stride, padding, init_scale, loggain
        self.v = nn.Parameter(torch.empty(out_dim, in_dim, kernel_size, kernel_size))
        self.gain = nn.Parameter(torch.empty(out_dim))
        self.b = nn.Parameter(torch.empty(out_dim), requires_grad=bias)
        nn.init.normal_(self.v, 0.0, 0.05)
        (nn.init.zeros_ if loggain else nn.init.ones_)(self.gain)
        nn.init.zeros_(self.b)
        self._w = None      # folded weight (fold())
        self._w5 = None     # per-kernel-row GEMM operands of a 5x5 conv (Model.fuse())
        self._wu = None     # Winograd-domain weights U [36, Cout, Cin] (Model.fuse()); channel-padded, see Model.fuse
        self._wp = None     # input convs: weight with zero filters appended (padded output channels)
        self._bp = None     # bias with zeros appended

    def weight(self):
        g = softplus(self.gain) if self.loggain else self.gain
        vnorm = self.v.view(self.out_dim, -1).norm(p=2, dim=1)
        return self.v * (g / (vnorm + _SMALL)).view(self.out_dim, 1, 1, 1)

    def fold(self):
        with torch.no_grad():
            self._w = self.weight().contiguous()

    def unfold(self):
        self._w = self._w5 = self._wu = self._wp = self._bp = None

    def bias_p(self):
        """Bias of the Winograd route (zero-padded to the padded channel count when there is one)."""
        return self._bp if self._bp is not None else self.b

    def forward(self, x):
        w = self._w if self._w is not None else self.weight()
        return F.conv2d(x, w, self.b, stride=self.stride, padding=self.padding)

    def _load_from_state_dict(self, *a, **k):
        self._w = self._w5 = self._wu = self._wp = self._bp = None
        return super()._load_from_state_dict(*a, **k)


class Pass(nn.Module):
    def forward(self, x):
        return x


class Squeeze2d(nn.Module):
    """[C,H,W] -> [C*f*f, H/f, W/f] (modules.py:169-191)."""

    def __init__(self, factor=2):
        super().__init__()
        self.factor = factor

    def forward(self, x):
        f = self.factor
        n, c, h, w = x.shape
        x = x.view(n, c, h // f, f, w // f, f).permute(0, 1, 3, 5, 2, 4)
        return x.reshape(n, c * f * f, h // f, w // f)


class UnSqueeze2d(nn.Module):
    """[C,H,W] -> [C/f/f, H*f, W*f] (modules.py:194-213)."""

    def __init__(self, factor=2):
        super().__init__()
        self.factor = factor

    def forward(self, x):
        f = self.factor
        n, c, h, w = x.shape
        x = x.view(n, c // (f * f), f, f, h, w).permute(0, 1, 4, 2, 5, 3)
        return x.reshape(n, c // (f * f), h * f, w * f)


class ResNetLayer(nn.Module):
    """x + conv2(act(conv1(act(x)))) (modules.py:216-241); dropout is identity at p = 0 / eval."""

    def __init__(self, inchannels, outchannels, kernel_size, padding, dropout_p, act):
        super().__init__()
        self.act = act
        self.conv1 = WnConv2d(inchannels, outchannels, kernel_size, 1, padding, init_scale=1.0, loggain=True)
        self.dropout = nn.Dropout(dropout_p)
        self.dropout_p = dropout_p
        self.conv2 = WnConv2d(outchannels, outchannels, kernel_size, 1, padding, init_scale=0.0, loggain=False)

    def forward(self, x):
        h = self.act(self.conv1(self.act(x)))
        if self.dropout_p > 0.0:
            h = self.dropout(h)
        return x + self.conv2(h)


def _resblock(width, kernel_size, padding, nlayers, dropout_p, act):
    """ResNetBlock (modules.py:244-250): a Sequential whose children are named res<C>layer<i>."""
    blk = nn.Sequential()
    for i in range(nlayers):
        blk.add_module(f"res{width}layer{i + 1}", ResNetLayer(width, width, kernel_size, padding, dropout_p, act))
    return blk


def blas_backend_available(name):
    """Can torch route its GEMMs to `name`?  Probed once when a Model is fused: the route is part of what sender and
    receiver must share (the GEMM kernels differ in the last float32 bits), so it is decided up front and visibly,
    never by a silent fallback in the middle of a stream."""
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            prev = torch.backends.cuda.preferred_blas_library()
            torch.backends.cuda.preferred_blas_library(name)
            torch.backends.cuda.preferred_blas_library(prev)
        return True
    except Exception:
        return False


class _blas:
    """Route torch's GEMMs to a BLAS backend for the duration of a block (the previous choice comes back on exit).
    The Winograd-domain GEMMs (batch 36/64 of [C x C] x [C x tiles]) run at 102-107 TFLOP/s on the composable_kernel
    backend where the default heuristic picks a 60-76 TFLOP/s kernel for batch 36 (measured in round 1)."""

    def __init__(self, name):
        self.name, self.prev = name, None

    def __enter__(self):
        if self.name:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.prev = torch.backends.cuda.preferred_blas_library()
                torch.backends.cuda.preferred_blas_library(self.name)

    def __exit__(self, *exc):
        if self.prev is not None:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.backends.cuda.preferred_blas_library(self.prev)


class _RawConv:
    """Output of an input convolution whose bias + ELU the next block's first fused pass will apply."""

    def __init__(self, c, b):
        self.c, self.b = c, b


class _RawM:
    """Output of an input convolution run in the Winograd domain: M [ts*ts, Cout, N*T] = U x V, still without its
    bias and ELU -- the next fused pass applies A^T M A + b, ELU, and whatever follows."""

    def __init__(self, m, b, ts, shape):
        self.m, self.b, self.ts, self.shape = m, b, ts, tuple(shape)


class _ConvDone:
    """An input convolution already taken through bias, ELU and the forward transform of the block that follows
    (hip.conv3_wino): h = ELU(conv(x) + b) and V = B^T ELU(h) B."""

    def __init__(self, h, v):
        self.h, self.v = h, v


class _WinoOperand:
    """A block output that only exists as the Winograd operand V [36, C, N*T] of the 3x3 convs that follow."""

    def __init__(self, v, shape):
        self.v, self.shape = v, tuple(shape)


class Model(nn.Module):
    def __init__(self, xs=(3, 32, 32), nz=1, zchannels=16, nprocessing=1, kernel_size=3, resdepth=2,
                 reswidth=256, dropout_p=0., tag='', root_process=True, conditional_gen_std=False, nn_batch=None):
        super().__init__()
        self.compressing = False
        self.xs, self.nz, self.zchannels, self.nprocessing = tuple(xs), nz, zchannels, nprocessing
        self.zdim = (zchannels, 16, 16)
        self.resdepth, self.reswidth, self.kernel_size = resdepth, reswidth, kernel_size
        self.bitsscale = np.log2(np.e)
        self.perdimsscale = 1. / np.prod(self.xs)
        self.tag = tag
        self.best_elbo = np.inf
        self.nn_batch = nn_batch
        self.fused = False
        # ResNet convs of the fused path: "winograd" (transform-domain batched GEMM, 3x3 and 5x5), "gemm5" (5x5 as
        # five row GEMMs, 3x3 on MIOpen) or "miopen"; below gemm_min_batch images MIOpen always runs them.  Since round 2
        # the Winograd route wins at every batch size (13 chains: 19.8 -> 13.1 ms per step, one chain: 12.9 -> 10.4,
        # profiles/r02j), so the threshold is 1; BITSWAP_GEMM_MIN_BATCH overrides it for experiments
        self.conv_algo = "winograd"
        # input convs of the stacks (Cin = zchannels or 4 x image channels) in the Winograd domain too: leaves nothing to
        # MIOpen from gemm_min_batch blocks per call on, but the K = 8..12 batched GEMMs run no faster than MIOpen's
        # kernels (profiles/r02g_kernel_stats.txt: 177 vs 177 ms per step), so it is an option, not the default
        self.wino_inputs = os.environ.get("BITSWAP_WINO_INPUTS", "0") == "1"
        # ... except the 5x5 conv that opens the bottom inference stack (12 squeezed image channels -> reswidth), which
        # has no direct kernel of ours: it takes the Winograd-domain route (transform, bs_small_k_gemm_f32, fused pass) by
        # default since round 3, so that NO convolution of the compress path is left to MIOpen's per-shape algorithm choice
        self.wino_in5 = os.environ.get("BITSWAP_WINO_IN5", "1") == "1"
        # 3x3 input convs (Cin = zchannels) as a direct fp32 convolution fused with bias, ELU and the forward transform of
        # the block behind them (bs_conv3_wino_f32) instead of an MIOpen launch + a transform pass.  As fast as the pair it
        # replaces (155 us against 72 + 93 us alone at 400 blocks, the same bench line: profiles/r02x, r02y); kept on
        # because it takes a library's algorithm choice out of what sender and receiver have to agree on.
        self.fused_inputs = os.environ.get("BITSWAP_FUSED_INPUTS", "1") == "1"
        # Winograd route: run with the ResNet width padded to the next multiple of 64 when that costs at most 8 dead
        # channels (252, 254, 255 -> 256).  The batched GEMMs of [C x C] x [C x tiles] are 15-20 % faster at C = 256
        # (tile quantisation: 3 row tiles of 96 for 252 rows; profiles/r02m), the transform kernels pay 1.6 % more
        # traffic.  The dead channels carry zero weights and zero biases, so they stay exactly zero through ELU and the
        # residual adds and never reach a result.
        self.pad_channels = os.environ.get("BITSWAP_PAD_CHANNELS", "1") == "1"
        # the batched GEMMs of the Winograd route on our own fp32 MFMA kernel (bs_wino_gemm_f32) instead of the BLAS
        # library: every output is summed in one fixed order that depends on Cin alone, so (mu, scale) do not depend on
        # how many chains are coded per call, on the library version or on its heuristics.  Since round 3 there is no
        # size threshold: the persistent kernel balances any column count over the chip and has a one-wave shape for
        # the 16-channel head convolutions (round 2 sent products under 64 channels / 512 columns to the library, which
        # made a stream decodable only with the sender's chains-per-call).  BITSWAP_OWN_GEMM=0 restores the library for
        # experiments; the choice is part of the stream fingerprint (route_fingerprint()).
        self.own_gemm = os.environ.get("BITSWAP_OWN_GEMM", "1") == "1"
        # arithmetic of those products.  "bf16x3" (default since round 6, meta.DEFAULT_GEMM_ARITH) / "bf16x3x9": every float32
        # operand split exactly into three bfloat16 limbs, the product assembled from 6 / 9 limb products per k block on the bf16
        # matrix cores (bs_wino_gemm_bf16x3), float32 accumulate, one fixed order per output -- batch-invariant like the fp32
        # kernel, as accurate against float64 (profiles/r06e_bf16x3_error.json), but a different rounding of (mu, scale): the
        # fingerprint names it and receivers adopt what the record says (meta.adopt_route).  "fp32" (rounds 2-5):
        # v_mfma_f32_32x32x2_f32, float32 in, float32 accumulate.  The limb route is taken for the products with at least 128
        # output channels (the ResNet convolutions: 96 of the 130 products of a block step and ~95 % of their flops); the 16- /
        # 24-channel head products stay on the fp32 kernel.
        from .meta import DEFAULT_GEMM_ARITH
        self.gemm_arith = os.environ.get("BITSWAP_GEMM_ARITH") or DEFAULT_GEMM_ARITH
        assert self.gemm_arith in ("fp32", "bf16x3", "bf16x3x9"), self.gemm_arith
        self._ufrags = {}
        self._cp = reswidth
        self.gemm_min_batch = int(os.environ.get("BITSWAP_GEMM_MIN_BATCH", "1"))
        # torch.backends.cuda.preferred_blas_library for the Winograd-domain GEMMs ("" = torch's default)
        self.gemm_backend = os.environ.get("BITSWAP_GEMM_BACKEND", "ck") or None
        self._heads, self._heads_u, self._gen_mu_u, self._gen_b2 = {}, {}, None, None
        self.conditional_gen_std = conditional_gen_std
        pad5, pad = 2, (kernel_size - 1) // 2
        assert kernel_size % 2 == 1

        # ResNet layers dealt round-robin over the latent layers (mnist_train.py:66-72)
        depth = [0] * nz
        for k in range(resdepth):
            depth[k % nz] += 1
        scale = 1.0 / (nz ** 0.5)
        W, C, X4 = reswidth, zchannels, 4 * xs[0]

        self.softplus, self.sigmoid, self.act, self.actresnet = nn.Softplus(), nn.Sigmoid(), nn.ELU(), nn.ELU()
        act, actres = self.act, self.actresnet

        def conv_in(cin, k, p):
            return nn.Sequential(WnConv2d(cin, W, k, 1, p, init_scale=1.0, loggain=True), act)

        def res(k, p, n):
            return nn.Sequential(_resblock(W, k, p, n, dropout_p, actres), act) if n > 0 else Pass()

        def head(cout, s):
            return WnConv2d(W, cout, kernel_size, 1, pad, init_scale=s)

        # inference model, bottom layer (x -> z1)
        self.infer_in = nn.Sequential(Squeeze2d(2), WnConv2d(X4, W, 5, 1, pad5, init_scale=1.0, loggain=True), act)
        self.infer_res0 = res(5, pad5, nprocessing)
        self.infer_res1 = res(kernel_size, pad, depth[0])
        top = scale if nz > 1 else 2 ** 0.5 * scale
        self.infer_mu = head(C, top)
        self.infer_std = head(C, top)
        # deeper inference layers (z_i -> z_{i+1})
        self.deepinfer_in = nn.ModuleList([conv_in(C, kernel_size, pad) for _ in range(nz - 1)])
        self.deepinfer_res = nn.ModuleList([res(kernel_size, pad, depth[i + 1]) for i in range(nz - 1)])
        self.deepinfer_mu = nn.ModuleList(
            [nn.Sequential(head(C, scale if i < nz - 2 else 2 ** 0.5 * scale)) for i in range(nz - 1)])
        self.deepinfer_std = nn.ModuleList(
            [nn.Sequential(head(C, scale if i < nz - 2 else 2 ** 0.5 * scale)) for i in range(nz - 1)])
        # deeper generative layers (z_{i+1} -> z_i)
        self.deepgen_in = nn.ModuleList([conv_in(C, kernel_size, pad) for _ in range(nz - 1)])
        self.deepgen_res = nn.ModuleList([res(kernel_size, pad, depth[i + 1]) for i in range(nz - 1)])
        self.deepgen_mu = nn.ModuleList([nn.Sequential(head(C, scale)) for _ in range(nz - 1)])
        self.deepgen_std = nn.ModuleList([nn.Sequential(head(C, scale)) for _ in range(nz - 1)])
        # generative model, bottom layer (z1 -> x)
        self.gen_in = conv_in(C, kernel_size, pad)
        self.gen_res1 = res(kernel_size, pad, depth[0])
        self.gen_res0 = res(5, pad5, nprocessing)
        self.gen_mu = nn.Sequential(head(X4, 0.1), UnSqueeze2d(2))
        if conditional_gen_std:   # imagenetcrop_train.py:306-315
            self.gen_std = nn.Sequential(head(X4, 0.1), UnSqueeze2d(2))
        else:                     # mnist_train.py:306-308
            self.gen_std = nn.Parameter(torch.zeros(*self.xs))

    # ----------------------------------------------------------------------------------------
    def load_state_dict(self, *a, **k):
        """New weights invalidate everything derived from the old ones: call fold()/fuse() again."""
        self.fused, self._heads, self._heads_u, self._gen_mu_u, self._ufrags = False, {}, {}, None, {}
        return super().load_state_dict(*a, **k)

    def compress(self, compress=True):
        self.compressing = compress

    def fold(self):
        """Cache the weight-normalised conv weights (call after loading a checkpoint, in eval)."""
        for m in self.modules():
            if isinstance(m, WnConv2d):
                m.fold()
        return self

    def unfold(self):
        for m in self.modules():
            if isinstance(m, WnConv2d):
                m.unfold()
        return self

    @property
    def xdim(self):
        return int(np.prod(self.xs))

    @property
    def zdim_flat(self):
        return int(np.prod(self.zdim))

    # ---- fused compress-mode path (HIP only) ------------------------------------------------------
    def fuse(self, enable=True):
        """Enable the fused epilogues for compress mode (needs fold(); tensors on a HIP device)."""
        self.fused = bool(enable)
        self._heads = {}
        if enable and self.gemm_backend and not blas_backend_available(self.gemm_backend):
            import warnings
            warnings.warn(f"BLAS backend {self.gemm_backend!r} is not available in this torch build: the Winograd-domain "
                          "GEMMs use the default backend.  Streams written with one backend only decode with the same one.")
            self.gemm_backend = None
        if enable:
            assert all(m.dropout_p == 0.0 for m in self.modules() if isinstance(m, ResNetLayer)) or not self.training
            self.fold()
            with torch.no_grad():
                def stack(a, b):
                    a, b = (a[0] if isinstance(a, nn.Sequential) else a), (b[0] if isinstance(b, nn.Sequential) else b)
                    return torch.cat([a._w, b._w], 0).contiguous(), torch.cat([a.b, b.b], 0).contiguous()
                self._heads["infer0"] = stack(self.infer_mu, self.infer_std)
                for i in range(self.nz - 1):
                    self._heads[f"infer{i + 1}"] = stack(self.deepinfer_mu[i], self.deepinfer_std[i])
                    self._heads[f"gen{i + 1}"] = stack(self.deepgen_mu[i], self.deepgen_std[i])
                if not self.conditional_gen_std:
                    self._gen_scale = (((2. / 255.) / 8.) + softplus(self.gen_std)).contiguous()
                W = self.reswidth
                cp = (W + 63) // 64 * 64
                self._cp = cp if (self.pad_channels and 0 < cp - W <= 8) else W
                extra = self._cp - W

                def padw(w, cout=False, cin=False):
                    """Zero filters / zero input planes appended where a dimension is the ResNet width."""
                    if extra and cout and w.shape[0] == W:
                        w = torch.cat([w, w.new_zeros((extra,) + tuple(w.shape[1:]))], 0)
                    if extra and cin and w.shape[1] == W:
                        w = torch.cat([w, w.new_zeros((w.shape[0], extra) + tuple(w.shape[2:]))], 1)
                    return w.contiguous()

                def padb(b):
                    return torch.cat([b, b.new_zeros(extra)]).contiguous() if (extra and b.shape[0] == W) else b
                # the head convs (3x3) in the Winograd domain as well: U [36, Cout, W (padded)]
                from .winograd import transform_weights as _tw
                self._heads_u = {k: _tw(padw(w, cin=True)) for k, (w, b) in self._heads.items() if w.shape[-1] == 3}
                g0 = self.gen_mu[0]
                self._gen_mu_u = _tw(padw(g0._w, cin=True)) if g0.kernel_size == 3 else None
                if self.conditional_gen_std and g0.kernel_size == 3:
                    # crop model (imagenetcrop_train.py:306-315,417): the pixel mean and the pixel scale are two head
                    # convolutions of the same activations -- one Winograd-domain product with stacked filters
                    gs = self.gen_std[0]
                    self._gen_mu_u = _tw(padw(torch.cat([g0._w, gs._w], 0).contiguous(), cin=True))
                    self._gen_b2 = torch.cat([g0.b, gs.b], 0).detach().contiguous()
                # ResNet convs (3x3 and 5x5, Cin == Cout) in the Winograd domain: U = G w G^T, [36, Cout, Cin]; the input
                # convs of every stack (Cin = zchannels or 4 x image channels -> reswidth) get their U as well
                # (Model.wino_inputs)
                from .winograd import transform_weights
                ins = [self.infer_in[1], self.gen_in[0]] + [q[0] for q in self.deepinfer_in] + [q[0] for q in self.deepgen_in]
                for m in self.modules():
                    if (isinstance(m, WnConv2d) and m.kernel_size in (3, 5) and m.stride == 1
                            and m.padding == m.kernel_size // 2
                            and (m.in_dim == m.out_dim or any(m is q for q in ins))):
                        m._wu = transform_weights(padw(m._w, cout=True, cin=True))
                        m._bp = padb(m.b.detach())
                        if any(m is q for q in ins):
                            m._wp = padw(m._w, cout=True)
                    # 5x5 as five GEMMs (one per kernel row), the alternative path: W_dy [Cout, Cin*5]
                    if isinstance(m, WnConv2d) and m.kernel_size == 5 and m.in_dim == m.out_dim and m.stride == 1:
                        m._w5 = [m._w[:, :, dy, :].reshape(m.out_dim, m.in_dim * 5).contiguous() for dy in range(5)]
                # opt-in bf16x3 arithmetic: the weights' limbs, split and tiled as MFMA fragments once
                self._ufrags = {}
                if self.gemm_arith != "fp32" and self.own_gemm:
                    from . import hip as _hip
                    for m in self.modules():
                        u = getattr(m, "_wu", None)
                        if isinstance(m, WnConv2d) and u is not None and u.is_cuda and u.shape[1] >= 128 and u.shape[2] % 16 == 0:
                            self._ufrags[u.data_ptr()] = (u, _hip.frags_bf16x3(u))   # the tensor itself pins the address
        return self

    def set_gemm_arith(self, arith):
        """fp32 | bf16x3 | bf16x3x9 for the ResNet products of the fused route (see __init__); rebuilds the weight fragments.
        Receivers call it with the arithmetic the stream's fingerprint names (meta.adopt_route)."""
        assert arith in ("fp32", "bf16x3", "bf16x3x9"), arith
        if arith != self.gemm_arith:
            self.gemm_arith = arith
            if self.fused:
                self.fuse()
        return self

    @staticmethod
    def _conv_nb(m, x, padded=False):
        return F.conv2d(x, m._wp if (padded and m._wp is not None) else m._w, None, stride=m.stride, padding=m.padding)

    def _unpad(self, h):
        """NCHW activation leaving the Winograd route towards an unpadded consumer: drop the dead channels."""
        return h[:, : self.reswidth].contiguous() if (torch.is_tensor(h) and h.dim() == 4 and h.shape[1] == self._cp
                                                      and self._cp != self.reswidth) else h

    def _fused_in(self, seq, x, nxt=None):
        """Sequential([Squeeze2d,] WnConv2d, act) -> ELU(conv(x) + b), one epilogue launch.  If the block `nxt`
        that follows runs in the Winograd domain, bias and ELU are left to its first fused pass (_RawConv)."""
        from . import hip
        mods = list(seq.children())
        if isinstance(mods[0], Squeeze2d):
            x = mods[0](x).contiguous()
            mods = mods[1:]
        m = mods[0]
        follows = nxt is not None and not isinstance(nxt, Pass)
        if (self.conv_algo == "winograd" and (self.wino_inputs or (m.kernel_size == 5 and self.wino_in5))
                and m._wu is not None and x.shape[0] >= self.gemm_min_batch
                and x.shape[-1] % 4 == 0 and x.shape[-2] % 4 == 0):
            # the input conv itself in the Winograd domain: V = B^T x B (no bias, no activation in front of it),
            # M = U x V; its bias + ELU ride on the next fused pass
            ts = int(round(m._wu.shape[0] ** 0.5))
            shape_out = (x.shape[0], m.out_dim) + tuple(x.shape[2:])
            v = hip.wino_fused(x.contiguous(), tuple(x.shape), 0, None, None, False, ts_out=ts)[2]
            shape_out = (x.shape[0], m._wu.shape[1]) + tuple(x.shape[2:])        # padded channel count
            mm = hip.small_k_gemm(m._wu, v) if (m._wu.shape[2] <= 64 and v.shape[2] % 4 == 0) else torch.bmm(m._wu, v)
            raw = _RawM(mm, m.bias_p(), ts, shape_out)
            if follows and self._wino_ok(list(nxt[0].children()), torch.empty((shape_out[0], 0, shape_out[2], shape_out[3]))):
                return raw
            return self._unpad(hip.wino_fused(raw.m, shape_out, ts, m.bias_p(), None, True, want_act=True)[1])  # ELU(A^T M A + b)
        if follows and self._wino_ok(list(nxt[0].children()), x):
            nts = int(round(list(nxt[0].children())[0].conv1._wu.shape[0] ** 0.5))
            if (self.fused_inputs and m.kernel_size == 3 and m.in_dim <= 16 and m.stride == 1 and m.padding == 1
                    and hip.conv3_wino_supported(m.in_dim, x.shape[2], x.shape[3])):
                # direct 3x3 conv on the fp32 VALU + bias + ELU + forward transform in one launch (no MIOpen call)
                wp = m._wp if m._wp is not None else m._w
                return _ConvDone(*hip.conv3_wino(x.contiguous(), wp, m.bias_p(), 3, True, nts))
            return _RawConv(self._conv_nb(m, x, padded=True), m.bias_p())       # zero filters appended: padded width
        return hip.bias_residual_elu(self._conv_nb(m, x), m.b)[1]

    def _wino_ok(self, layers, h):
        return (self.conv_algo == "winograd" and h.shape[0] >= self.gemm_min_batch and layers[0].conv1._wu is not None
                and h.shape[-1] % 4 == 0 and h.shape[-2] % 4 == 0
                and int(round(layers[0].conv1._wu.shape[0] ** 0.5)) - layers[0].conv1.kernel_size + 1 == 4)

    @staticmethod
    def _conv5_gemm(m, ax):
        """5x5 'same' convolution on the x-expanded operand ax [n, Cin*5, H+4, W] (hip.expand_rows5): kernel
        row dy is one strided-batched GEMM on rocBLAS/hipBLASLt (MFMA), accumulated in place.  At 100-200
        images this runs at ~103 TFLOP/s against 74-80 for MIOpen's decomposition of 5x5 into 3x3 Winograd
        tiles (measured in round 1).  No bias."""
        n, k5, hp, w = ax.shape
        h = hp - 4
        out = torch.empty((n, m.out_dim, h * w), dtype=ax.dtype, device=ax.device)
        for dy in range(5):
            a = ax[:, :, dy:dy + h, :].flatten(2)                       # [n, Cin*5, H*W], row stride (H+4)*W
            wd = m._w5[dy].unsqueeze(0).expand(n, m.out_dim, k5)
            if dy == 0:
                torch.bmm(wd, a, out=out)
            else:
                out.baddbmm_(wd, a)
        return out.view(n, m.out_dim, h, w)

    def _res5_gemm(self, layers, h):
        """The 5x5 ResNet block on GEMMs: the ELU in front of every conv is applied while its operand is
        expanded (bias of conv1 included), so a layer is 2 expansions + 10 GEMMs + 1 residual epilogue."""
        from . import hip
        for k, L in enumerate(layers):
            c1 = self._conv5_gemm(L.conv1, hip.expand_rows5(h, None, act=True))
            c2 = self._conv5_gemm(L.conv2, hip.expand_rows5(c1, L.conv1.b, act=True))
            if k == len(layers) - 1:      # the block is followed by act: only ELU(sum) is needed
                return hip.bias_residual_elu(c2, L.conv2.b, h)[1]
            h = hip.bias_residual_elu(c2, L.conv2.b, h, want_sum=True, want_act=False)[0]

    def _res_wino(self, layers, h, want_v=False):
        """A ResNet block in the Winograd domain (bitswap_amd/winograd.py): per conv ONE batched GEMM of 36 / 64
        [C x C] x [C x tiles] products on rocBLAS/hipBLASLt, and between two GEMMs ONE fused pass
        (k_wino_fused: inverse transform, bias, residual, ELU, forward transform of the next operand).  2.25x (3x3) /
        6.25x (5x5) fewer multiplications than the direct convolution, all of them on the MFMA units."""
        from . import hip
        ts = int(round(layers[0].conv1._wu.shape[0] ** 0.5))
        if isinstance(h, _ConvDone):
            h, v = h.h, h.v
            shape = tuple(h.shape)
        elif isinstance(h, _RawM):      # same, the input conv still in the Winograd domain: h = ELU(A^T M A + b)
            shape = h.shape
            _, h, v = hip.wino_fused(h.m, shape, h.ts, h.b, None, 3, want_act=True, ts_out=ts)
        elif isinstance(h, _RawConv):   # the input conv's bias + ELU ride along: h = ELU(c + b), V = B^T ELU(h) B
            shape = tuple(h.c.shape)
            _, h, v = hip.wino_fused(h.c, shape, 0, h.b, None, 3, want_act=True, ts_out=ts)
        else:
            cin = layers[0].conv1._wu.shape[2]
            if h.shape[1] < cin:            # an unpadded activation entering the padded route: append the dead channels
                h = torch.cat([h, h.new_zeros((h.shape[0], cin - h.shape[1]) + tuple(h.shape[2:]))], 1)
            shape = tuple(h.shape)
            if ts - layers[0].conv1.kernel_size + 1 != 4:          # F(2x2,5x5) alternative: separate transforms
                return self._res_wino_unfused(layers, h, (ts, ts - layers[0].conv1.kernel_size + 1))
            v = hip.wino_fused(h, shape, 0, None, None, True, ts_out=ts)[2]                   # B^T ELU(h) B
        for k, L in enumerate(layers):
            b1, b2 = L.conv1.bias_p(), L.conv2.bias_p()
            v = hip.wino_fused(self._bmm(L.conv1._wu, v), shape, ts, b1, None, True, ts_out=ts)[2]
            m2 = self._bmm(L.conv2._wu, v)
            if k == len(layers) - 1:      # the block is followed by act: only ELU(sum) is needed
                if want_v:                # ... and only as the operand of the 3x3 head convs
                    return _WinoOperand(hip.wino_fused(m2, shape, ts, b2, h, True, ts_out=6)[2], shape)
                return hip.wino_fused(m2, shape, ts, b2, h, True, want_act=True)[1]
            h, _, v = hip.wino_fused(m2, shape, ts, b2, h, True, want_sum=True, ts_out=ts)

    def _bmm(self, U, V):
        """The batched product of a Winograd-domain convolution: our MFMA kernel (one fixed summation order per output,
        whatever the batch) whenever it takes the shapes (Cin % 16 == 0: every preset of the reference), else the BLAS
        library (narrow test models) -- a property of the model, never of the call."""
        from . import hip
        if self.own_gemm and hip.wino_gemm_supported(U, V):
            uf = self._ufrags.get(U.data_ptr()) if self._ufrags else None
            if uf is not None and uf[0] is U:
                return hip.wino_gemm_bf16x3(uf[1], V, 9 if self.gemm_arith == "bf16x3x9" else 6)
            return hip.wino_gemm(U, V)
        return torch.bmm(U, V)

    def _res_wino_unfused(self, layers, h, cfg):
        from . import hip
        shape = tuple(h.shape)
        for k, L in enumerate(layers):
            m1 = self._bmm(L.conv1._wu, hip.wino_in(h, None, True, cfg))              # conv1(ELU(h))
            t = hip.wino_out(m1, shape, L.conv1.bias_p(), None, False, True, cfg)[1]  # ELU(. + b1)
            m2 = self._bmm(L.conv2._wu, hip.wino_in(t, None, False, cfg))
            if k == len(layers) - 1:
                return hip.wino_out(m2, shape, L.conv2.bias_p(), h, False, True, cfg)[1]
            h = hip.wino_out(m2, shape, L.conv2.bias_p(), h, True, False, cfg)[0]

    def _fused_res(self, seq, h, want_v=False):
        """Sequential(ResNetBlock, act) on an activated input h (Pass: identity).  Per layer
        x + conv2(act(conv1(act(x)))): two convs, two epilogue launches.  want_v: the caller only feeds 3x3
        head convs with the result and accepts it as a Winograd operand (_WinoOperand) instead of NCHW."""
        from . import hip
        if isinstance(seq, Pass):
            return h
        layers = list(seq[0].children())
        if isinstance(h, (_RawConv, _RawM, _ConvDone)):
            return self._res_wino(layers, h, want_v)
        if (self.conv_algo == "winograd" and h.shape[0] >= self.gemm_min_batch and layers[0].conv1._wu is not None
                and h.shape[-1] % 4 == 0 and h.shape[-2] % 4 == 0):
            return self._res_wino(layers, h, want_v and layers[0].conv1._wu.shape[0] in (36, 64))
        h = self._unpad(h)
        if (self.conv_algo in ("winograd", "gemm5") and layers[0].conv1.kernel_size == 5
                and h.shape[0] >= self.gemm_min_batch and h.shape[-1] % 4 == 0 and layers[0].conv1._w5 is not None):
            return self._res5_gemm(layers, h)
        a = hip.bias_residual_elu(h, None, inplace=False)[1]          # act(x) of the first layer
        for k, L in enumerate(layers):
            t = hip.bias_residual_elu(self._conv_nb(L.conv1, a), L.conv1.b)[1]
            c2 = self._conv_nb(L.conv2, t)
            if k == len(layers) - 1:      # the block is followed by act: only ELU(sum) is needed
                return hip.bias_residual_elu(c2, L.conv2.b, h)[1]
            h, a = hip.bias_residual_elu(c2, L.conv2.b, h, want_sum=True, want_act=True)

    def _fused_head(self, key, h, mode):
        from . import hip
        w, b = self._heads[key]
        if isinstance(h, _WinoOperand):     # head convs as one more batched GEMM on the operand the block left
            x = hip.wino_fused(self._bmm(self._heads_u[key], h.v), (h.shape[0], w.shape[0]) + h.shape[2:], 6, None,
                               None, False, want_sum=True)[0]
            return hip.head_params(x, b, mode)
        return hip.head_params(F.conv2d(self._unpad(h), w, None, stride=1, padding=(w.shape[-1] - 1) // 2), b, mode)

    def _infer_stack_fused(self, i, h):
        from . import hip
        hv = f"infer{i}" in self._heads_u
        if i == 0:
            h = self._fused_res(self.infer_res1, self._fused_res(self.infer_res0, self._fused_in(self.infer_in, h, self.infer_res0)), hv)
        else:
            h = self._fused_res(self.deepinfer_res[i - 1], self._fused_in(self.deepinfer_in[i - 1], h, self.deepinfer_res[i - 1]), hv)
        return self._fused_head(f"infer{i}", h, hip.HEAD_SIGMOID)

    def _gen_stack_fused(self, i, h):
        from . import hip
        if i == 0:
            hv = self._gen_mu_u is not None
            h = self._fused_res(self.gen_res0, self._fused_res(self.gen_res1, self._fused_in(self.gen_in, h, self.gen_res1)), hv)
            if isinstance(h, _WinoOperand) and self.conditional_gen_std:
                c = self.gen_mu[0].out_dim
                x = hip.wino_fused(self._bmm(self._gen_mu_u, h.v), (h.shape[0], 2 * c) + h.shape[2:], 6, self._gen_b2,
                                   None, False, want_sum=True)[0]
                return self.gen_mu[1](x[:, :c]), ((2. / 255.) / 8.) + softplus(self.gen_std[1](x[:, c:]))
            if isinstance(h, _WinoOperand):
                g0 = self.gen_mu[0]
                x = hip.wino_fused(self._bmm(self._gen_mu_u, h.v), (h.shape[0], g0.out_dim) + h.shape[2:], 6, g0.b,
                                   None, False, want_sum=True)[0]
                return self.gen_mu[1](x), self._gen_scale
            h = self._unpad(h)
            mu = self.gen_mu(h)
            if self.conditional_gen_std:
                scale = ((2. / 255.) / 8.) + softplus(self.gen_std(h))
            else:
                scale = self._gen_scale
            return mu, scale
        h = self._fused_res(self.deepgen_res[i - 1], self._fused_in(self.deepgen_in[i - 1], h, self.deepgen_res[i - 1]),
                            f"gen{i}" in self._heads_u)
        return self._fused_head(f"gen{i}", h, hip.HEAD_SOFTPLUS)

    def _use_fused(self, h):
        return self.fused and self.compressing and h.is_cuda and not torch.is_grad_enabled()

    # the conv stacks proper, on a [n, C, H, W] float32 batch -------------------------------
    def _infer_stack(self, i, h):
        if self._use_fused(h):
            with _blas(self.gemm_backend if self.conv_algo == "winograd" else None):
                return self._infer_stack_fused(i, h.contiguous())
        if i == 0:
            h = self.infer_res1(self.infer_res0(self.infer_in(h)))
            mu = self.infer_mu(h)
            scale = 0.1 + 0.9 * self.sigmoid(self.infer_std(h) + 2.)
        else:
            h = self.deepinfer_res[i - 1](self.deepinfer_in[i - 1](h))
            mu = self.deepinfer_mu[i - 1](h)
            scale = 0.1 + 0.9 * self.sigmoid(self.deepinfer_std[i - 1](h) + 2.)
        return mu, scale

    def _gen_stack(self, i, h):
        if self._use_fused(h):
            with _blas(self.gemm_backend if self.conv_algo == "winograd" else None):
                return self._gen_stack_fused(i, h.contiguous())
        if i == 0:
            h = self.gen_res0(self.gen_res1(self.gen_in(h)))
            mu = self.gen_mu(h)
            std = self.gen_std(h) if self.conditional_gen_std else self.gen_std
            scale = ((2. / 255.) / 8.) + softplus(std)
        else:
            h = self.deepgen_res[i - 1](self.deepgen_in[i - 1](h))
            mu = self.deepgen_mu[i - 1](h)
            scale = 0.1 + 0.9 * softplus(self.deepgen_std[i - 1](h) + np.log(np.exp(1.) - 1.))
        return mu, scale

    def _chunked(self, fn, h, out_shape):
        """Run fn on fixed-shape micro-batches so results are independent of the chain count."""
        nb = self.nn_batch
        n = h.shape[0]
        if not nb or n == nb:
            mu, sc = fn(h)
            return mu, sc.expand_as(mu)
        mus, scs = [], []
        for s in range(0, n, nb):
            part = h[s:s + nb]
            k = part.shape[0]
            if k < nb:
                part = torch.cat([part, part.new_zeros((nb - k,) + tuple(part.shape[1:]))], 0)
            mu, sc = fn(part)
            mus.append(mu[:k])
            scs.append(sc.expand_as(mu)[:k])
        return torch.cat(mus, 0), torch.cat(scs, 0)

    # ----------------------------------------------------------------------------------------
    def infer(self, i):
        def distribution(given):
            h = given
            if self.compressing:
                flat1d = h.dim() == 1
                in_dtype = h.dtype
                h = h.float().view((-1,) + (self.xs if i == 0 else self.zdim))
                mu, scale = self._chunked(lambda t: self._infer_stack(i, t), h, None)
                mu, scale = mu.reshape(mu.shape[0], -1), scale.reshape(scale.shape[0], -1)
                if flat1d:  # reference behaviour: batch of one, flattened, back in the input dtype
                    assert mu.shape[0] == 1
                    return mu.view(-1).to(in_dtype), scale.view(-1).to(in_dtype)
                return mu, scale
            if i == 0:
                h = (h - 127.5) / 127.5
            return self._infer_stack(i, h)
        distribution.bs_key = ("infer", i)        # which stack this closure is (schedulers and tests tell them apart by it)
        return distribution

    def generate(self, i):
        def distribution(given):
            h = given
            if self.compressing:
                flat1d = h.dim() == 1
                in_dtype = h.dtype
                h = h.float().view((-1,) + self.zdim)
                mu, scale = self._chunked(lambda t: self._gen_stack(i, t), h, None)
                mu, scale = mu.reshape(mu.shape[0], -1), scale.reshape(scale.shape[0], -1).contiguous()
                if flat1d:
                    assert mu.shape[0] == 1
                    return mu.view(-1).to(in_dtype), scale.view(-1).to(in_dtype)
                return mu, scale
            return self._gen_stack(i, h)
        distribution.bs_key = ("generate", i)
        return distribution

    # ELBO terms for the `elbos` metric of the CLIs (mnist_train.py:441-490) ------------------
    def loss(self, x):
        from . import rand as random
        B = x.shape[0]
        logenc = torch.zeros((self.nz, B, self.zdim[0]), device=x.device)
        logdec = torch.zeros((self.nz, B, self.zdim[0]), device=x.device)
        zsamples = torch.zeros((self.nz, B, self.zdim_flat), device=x.device)
        z = None
        for i in range(self.nz):
            mu, scale = self.infer(i)(given=x if i == 0 else z)
            z_next = random.transform(random.logistic_eps(mu.shape, device=mu.device), mu, scale)
            zsamples[i] = z_next.flatten(1)
            logenc[i] += torch.sum(random.logistic_logp(mu, scale, z_next), dim=2)
            mu, scale = self.generate(i)(given=z_next)
            if i == 0:
                logrecon = torch.sum(random.discretized_logistic_logp(mu, scale, x), dim=1)
            else:
                logdec[i - 1] += torch.sum(random.logistic_logp(mu, scale, z), dim=2)
            z = z_next
        one, zero = torch.ones(1, device=x.device), torch.zeros(1, device=x.device)
        logdec[self.nz - 1] += torch.sum(random.logistic_logp(zero, one, z), dim=2)
        logenc = torch.mean(logenc, dim=1) * self.bitsscale
        logdec = torch.mean(logdec, dim=1) * self.bitsscale
        logrecon = torch.mean(logrecon) * self.bitsscale
        return logrecon, logdec, logenc, zsamples


def elbo_bits(model, x, batch=256):
    """Per-image negative ELBO in bits, [B]: what the reference computes with a batch of one as
    `-logrecon + sum(-logdec + logenc)` (mnist_compress.py:170-174) for its `elbos` metric.  Model.loss()
    (mnist_train.py:441-490) averages its terms over the batch; here the same terms are kept per image, so one pass
    over a batch gives every image's value (the reference's loop makes 10,000 batch-1 passes for 100 x 100 images)."""
    from . import rand as random
    was = model.compressing
    model.compress(False)
    try:
        with torch.no_grad():
            out = []
            for s in range(0, x.shape[0], batch):
                xb = x[s:s + batch]
                n = xb.shape[0]
                total = torch.zeros(n, device=xb.device)
                z = None
                for i in range(model.nz):
                    mu, scale = model.infer(i)(given=xb if i == 0 else z)
                    z_next = random.transform(random.logistic_eps(mu.shape, device=mu.device), mu, scale)
                    total += random.logistic_logp(mu, scale, z_next).flatten(1).sum(1)               # + logenc_i
                    mu, scale = model.generate(i)(given=z_next)
                    if i == 0:
                        total -= random.discretized_logistic_logp(mu, scale, xb).sum(1)             # - logrecon
                    else:
                        total -= random.logistic_logp(mu, scale, z).flatten(1).sum(1)               # - logdec_{i-1}
                    z = z_next
                one, zero = torch.ones(1, device=xb.device), torch.zeros(1, device=xb.device)
                total -= random.logistic_logp(zero, one, z).flatten(1).sum(1)                       # - logdec_{nz-1}
                out.append(total * model.bitsscale)
            return torch.cat(out)
    finally:
        model.compress(was)


# dataset presets of the reference CLIs ----------------------------------------------------------
def preset(dataset, nz, **kw):
    """Model configured like <dataset>_compress.py (widths: mnist_compress.py:81-88,
    cifar_compress.py:80-87, imagenet_compress.py:83-90, imagenetcrop_compress.py:100)."""
    if dataset == "mnist":
        w = {8: 61, 4: 62, 2: 63}.get(nz, 64)
        return Model(xs=(1, 32, 32), nz=nz, zchannels=1, nprocessing=4, kernel_size=3, resdepth=8, reswidth=w, **kw)
    if dataset == "cifar":
        w = {8: 252, 4: 254, 2: 255}.get(nz, 256)
        return Model(xs=(3, 32, 32), nz=nz, zchannels=8, nprocessing=4, kernel_size=3, resdepth=8, reswidth=w, **kw)
    if dataset == "imagenet":
        w = {8: 252, 4: 254, 2: 255}.get(nz, 256)
        return Model(xs=(3, 32, 32), nz=nz, zchannels=8, nprocessing=4, kernel_size=3, resdepth=8, reswidth=w, **kw)
    if dataset == "imagenetcrop":
        return Model(xs=(3, 32, 32), nz=nz, zchannels=8, nprocessing=4, kernel_size=3, resdepth=8, reswidth=256,
                     conditional_gen_std=True, **kw)
    raise ValueError(dataset)
