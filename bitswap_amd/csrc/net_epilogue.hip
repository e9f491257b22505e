// net_epilogue.hip -- fused pointwise epilogues between the MIOpen convolutions of the conv stacks.
//
// The convolutions of Model.infer / Model.generate stay PyTorch-ROCm (MIOpen) calls.  What sits
// between them in the reference -- bias add, ELU, residual add (utils/torch/modules.py:216-241) and
// the scale heads (model/mnist_train.py:349,368,426) -- is HBM-bound pointwise work that torch runs as
// 3-5 separate launches per ResNet layer.  Here each conv is followed by exactly one pass over its
// output: one 16-byte load per operand, one 16-byte store per result.  NCHW float32, contiguous.
//
// These kernels are deterministic and batch-invariant by construction (every output element depends
// on its own inputs only), which is all the decoder needs.
//
// BUILD REQUIREMENT (correctness, not tuning): this file must be compiled with -fno-slp-vectorize.  Under plain -O3 hipcc packs the
// transforms' scalar float32 additions into v_pk_add_f32; beside the bf16 MFMA wavefronts of bs_wino_gemm_bf16x3 one such packed
// addition lost its result in lanes 48..63 about once in 10^5 workgroups (profiles/r05_packed_add_report.md).  The
// build says so with -DBS_BUILT_WITHOUT_SLP_VECTORIZER (bitswap_amd/build.py::FILE_FLAGS); any other build system gets an error
// here instead of a library that decodes one chain in 10^5 wrong.
#if !defined(BS_BUILT_WITHOUT_SLP_VECTORIZER) && !defined(BS_PACKED_EPILOGUE_DIAGNOSTIC)
#error "net_epilogue.hip: compile with -fno-slp-vectorize -DBS_BUILT_WITHOUT_SLP_VECTORIZER (see the comment above)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/bitswap_hip.h"

// Issue priority of the transform kernels' wavefronts (k_wino_fused, k_conv3_wino).  In the two-group pipeline such a kernel of
// one chain group starts beside an OLDER table kernel or GEMM of the other group, and a SIMD picks the wavefront to issue by
// priority, then age: at equal priority these short memory-bound kernels wait behind long VALU-bound ones.  One step above the
// default lets them through -- shortest job first: 1000 chains 161.8 -> 158.6 ms per step (three repetitions, 1 / 2 / 3 the same),
// the other shapes -0.5 % (profiles/r06v_priority_ab.txt); the same knob on the GEMM's wavefronts moves nothing
// (profiles/r06u_priority_ab.txt).  Scheduling only: no result bit depends on it.  The serial coder kernels run at 3.
#ifndef BS_XFORM_PRIO
#define BS_XFORM_PRIO 1
#endif

namespace {

__device__ __forceinline__ float elu1(float v) { return v > 0.0f ? v : expm1f(v); }

// softplus as the reference writes it: -logsigmoid(-x) (modules.py:112-114) = max(x, 0) + log1p(exp(-|x|))
__device__ __forceinline__ float softplus_ref(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }

// one thread = 4 consecutive elements of one (n, c) plane (HW % 4 == 0)
template <bool HAS_RES, bool WANT_SUM, bool WANT_ACT>
__global__ __launch_bounds__(256) void k_bias_res_elu(const float* __restrict__ x, const float* __restrict__ bias,
                                                      const float* __restrict__ res, float* __restrict__ sum_out,
                                                      float* __restrict__ act_out, int64_t n4, int C, int hw4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i / hw4) % C);
    const float b = bias ? bias[c] : 0.0f;
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (HAS_RES) {
        const float4 r = reinterpret_cast<const float4*>(res)[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (WANT_SUM) reinterpret_cast<float4*>(sum_out)[i] = v;
    if (WANT_ACT) {
        float4 a;
        a.x = elu1(v.x); a.y = elu1(v.y); a.z = elu1(v.z); a.w = elu1(v.w);
        reinterpret_cast<float4*>(act_out)[i] = a;
    }
}

// scalar tail version for planes whose size is not a multiple of 4
template <bool HAS_RES, bool WANT_SUM, bool WANT_ACT>
__global__ __launch_bounds__(256) void k_bias_res_elu_1(const float* __restrict__ x, const float* __restrict__ bias,
                                                        const float* __restrict__ res, float* __restrict__ sum_out,
                                                        float* __restrict__ act_out, int64_t n, int C, int hw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / hw) % C);
    float v = x[i] + (bias ? bias[c] : 0.0f);
    if (HAS_RES) v += res[i];
    if (WANT_SUM) sum_out[i] = v;
    if (WANT_ACT) act_out[i] = elu1(v);
}

// heads: x [N, 2C, HW] = one conv with the mu and the std filters stacked; mu = x[:, :C] + b,
// scale = transform(x[:, C:] + b)
__global__ __launch_bounds__(256) void k_head_params(const float* __restrict__ x, const float* __restrict__ bias,
                                                     float* __restrict__ mu, float* __restrict__ scale, int64_t n,
                                                     int C, int hw, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into [N, C, HW]
    if (i >= n) return;
    const int64_t plane = i / hw;
    const int p = (int)(i - plane * hw);
    const int c = (int)(plane % C);
    const int64_t nb = plane / C;
    const float* xr = x + (nb * 2 * C) * hw;
    const float m = xr[(int64_t)c * hw + p] + bias[c];
    const float s = xr[(int64_t)(C + c) * hw + p] + bias[C + c];
    mu[i] = m;
    float sc;
    if (mode == BS_HEAD_SIGMOID) sc = 0.1f + 0.9f * sigmoid_f32(s + 2.0f);                // mnist_train.py:349,368
    else sc = 0.1f + 0.9f * softplus_ref(s + 0.54132485461291810f /* log(e - 1) */);     // mnist_train.py:426
    scale[i] = sc;
}

// x-expansion for the 5x5 convolutions-as-GEMMs (model.py, _conv5_gemm): out [N, C*5, H+4, W] with
//   out[n, c*5 + dx, yy, x] = act(in[n, c, yy - 2, x + dx - 2] + bias[c])   (0 outside the image)
// so that kernel row dy of the convolution is ONE strided-batched GEMM W_dy [Cout, C*5] x out[n, :, dy:dy+H, :]
// (a [C*5, H*W] matrix with leading dimension (H+4)*W: the H rows starting at dy are contiguous).
// One thread = 4 consecutive x of one output row (W % 4 == 0): one 16-byte store, <= 4 cached loads.
__global__ __launch_bounds__(256) void k_expand_rows5(const float* __restrict__ in, const float* __restrict__ bias,
                                                      float* __restrict__ out, int64_t total4, int C, int H, int W,
                                                      int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int w4 = W / 4;
    const int xq = (int)(i % w4);
    int64_t r = i / w4;
    const int yy = (int)(r % (H + 4));
    r /= (H + 4);
    const int dx = (int)(r % 5);
    r /= 5;
    const int c = (int)(r % C);
    const int64_t n = r / C;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int y = yy - 2;
    if (y >= 0 && y < H) {
        const float b = bias ? bias[c] : 0.0f;
        const float* row = in + ((n * C + c) * H + y) * (int64_t)W;
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xq * 4 + k + dx - 2;
            float e = 0.0f;
            if (x >= 0 && x < W) {
                e = row[x] + b;
                if (act) e = elu1(e);
            }
            t[k] = e;
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
    }
    reinterpret_cast<float4*>(out)[i] = v;
}

// ------------------------------------------------------------------------------------------
// Winograd-domain convolutions.  Three Cook-Toom algorithms (bitswap_amd/winograd.py):
//     F(4x4, 3x3)  6x6 tiles, stride 4, pad 1    points {0, +-1, +-2, inf}           36 products / 16 outputs
//     F(2x2, 5x5)  6x6 tiles, stride 2, pad 2    same points, same B^T             36 products /  4 outputs
//     F(4x4, 5x5)  8x8 tiles, stride 4, pad 2    points {0, +-1, +-2, +-1/2, inf}  64 products / 16 outputs
// The TS*TS element-wise products per tile become ONE batched GEMM on rocBLAS/hipBLASLt (MFMA):
//     M[t] [Cout, N] = U[t] [Cout, Cin] x V[t] [Cin, N],  t = 0..TS*TS-1,  N = images * tiles
// with 2.25x / 2.78x / 6.25x fewer multiplications than the direct convolution.  The two transforms
// below are the HBM-bound ends of it; they also carry the pointwise work of the ResNet layer (bias / ELU
// in front, bias / residual / ELU behind), so nothing else touches the activations between two GEMMs.
// fp32 throughout; error vs the direct convolution a few 1e-6 of the output range
// (tests/test_codec_gpu.py::test_winograd_convs_match_torch).  Formulas: tools/gen_wino.py.
// ------------------------------------------------------------------------------------------
template <int TS>
__device__ __forceinline__ void wino_bt(const float (&d)[TS], float (&o)[TS]) {
    if (TS == 6) {
        o[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
        o[1] = -4.0f * d[1] - 4.0f * d[2] + d[3] + d[4];
        o[2] = 4.0f * d[1] - 4.0f * d[2] - d[3] + d[4];
        o[3] = -2.0f * d[1] - d[2] + 2.0f * d[3] + d[4];
        o[4] = 2.0f * d[1] - d[2] - 2.0f * d[3] + d[4];
        o[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
    } else {
        o[0] = -d[0] + 5.25f * d[2] - 5.25f * d[4] + d[6];
        o[1] = d[1] + d[2] - 4.25f * d[3] - 4.25f * d[4] + d[5] + d[6];
        o[2] = -d[1] + d[2] + 4.25f * d[3] - 4.25f * d[4] - d[5] + d[6];
        o[3] = 0.5f * d[1] + 0.25f * d[2] - 2.5f * d[3] - 1.25f * d[4] + 2.0f * d[5] + d[6];
        o[4] = -0.5f * d[1] + 0.25f * d[2] + 2.5f * d[3] - 1.25f * d[4] - 2.0f * d[5] + d[6];
        o[5] = 2.0f * d[1] + 4.0f * d[2] - 2.5f * d[3] - 5.0f * d[4] + 0.5f * d[5] + d[6];
        o[6] = -2.0f * d[1] + 4.0f * d[2] + 2.5f * d[3] - 5.0f * d[4] - 0.5f * d[5] + d[6];
        o[7] = -d[1] + 5.25f * d[3] - 5.25f * d[5] + d[(TS == 8) ? 7 : 0];
    }
}

template <int TS, int MS>
__device__ __forceinline__ void wino_at(const float (&m)[TS], float (&y)[MS]) {
    if (TS == 6) {
        y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
        if (MS == 2) {
            y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4] + m[5];
        } else {
            y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4];
            y[(MS == 4) ? 2 : 0] = m[1] + m[2] + 4.0f * m[3] + 4.0f * m[4];
            y[(MS == 4) ? 3 : 0] = m[1] - m[2] + 8.0f * m[3] - 8.0f * m[4] + m[5];
        }
    } else {
        constexpr int a6 = (TS == 8) ? 6 : 0, a7 = (TS == 8) ? 7 : 0;
        y[0] = m[0] + m[1] + m[2] + m[3] + m[4] + m[5] + m[a6];
        y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4] + 0.5f * m[5] - 0.5f * m[a6];
        y[(MS == 4) ? 2 : 0] = m[1] + m[2] + 4.0f * m[3] + 4.0f * m[4] + 0.25f * m[5] + 0.25f * m[a6];
        y[(MS == 4) ? 3 : 0] = m[1] - m[2] + 8.0f * m[3] - 8.0f * m[4] + 0.125f * m[5] - 0.125f * m[a6] + m[a7];
    }
}

// in [N, C, H, W] -> V [TS*TS, C, N*T] (T = (H/MS)*(W/MS) tiles per image, column = n*T + ty*(W/MS) + tx):
// V[t] = (B^T d B)[t/TS][t%TS] of the TS x TS window of act(in + bias) at rows ty*MS - PAD .. (0 outside).
// One thread per (channel, column): its TS*TS stores are coalesced across the wave.
template <int TS, int MS>
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ in, const float* __restrict__ bias,
                                                 float* __restrict__ V, int64_t total, int C, int H, int W,
                                                 int64_t ncols, int act) {
    constexpr int PAD = (TS - MS) / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t col = i % ncols;
    const int c = (int)(i / ncols);
    const int ntx = W / MS, T = (H / MS) * ntx;
    const int64_t n = col / T;
    const int tile = (int)(col - n * T);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const float b = bias ? bias[c] : 0.0f;
    const float* plane = in + (n * C + c) * (int64_t)H * W;
    float t1[TS][TS];  // B^T d: the window is read column by column and transformed on the fly
#pragma unroll
    for (int q = 0; q < TS; ++q) {
        const int x = tx * MS - PAD + q;
        float colv[TS], o[TS];
#pragma unroll
        for (int r = 0; r < TS; ++r) {
            const int y = ty * MS - PAD + r;
            float e = 0.0f;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                e = plane[y * W + x] + b;
                if (act) e = elu1(e);
            }
            colv[r] = e;
        }
        wino_bt<TS>(colv, o);
#pragma unroll
        for (int r = 0; r < TS; ++r) t1[r][q] = o[r];
    }
    float* out = V + (int64_t)c * ncols + col;
    const int64_t tstride = (int64_t)C * ncols;
#pragma unroll
    for (int r = 0; r < TS; ++r) {  // (B^T d) B
        float o[TS];
        wino_bt<TS>(t1[r], o);
#pragma unroll
        for (int q = 0; q < TS; ++q) out[(int64_t)(r * TS + q) * tstride] = o[q];
    }
}

// M [TS*TS, C, N*T] -> s = A^T M A + bias[c] (+ res) as [N, C, H, W]; sum_out = s (nullable), act_out = ELU(s)
// (nullable).  One thread per (channel, column) = one MS x MS output tile.
template <int TS, int MS>
__global__ __launch_bounds__(256) void k_wino_out(const float* __restrict__ M, const float* __restrict__ bias,
                                                  const float* __restrict__ res, float* __restrict__ sum_out,
                                                  float* __restrict__ act_out, int64_t total, int C, int H, int W,
                                                  int64_t ncols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t col = i % ncols;
    const int c = (int)(i / ncols);
    const int ntx = W / MS, T = (H / MS) * ntx;
    const int64_t n = col / T;
    const int tile = (int)(col - n * T);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const float* src = M + (int64_t)c * ncols + col;
    const int64_t tstride = (int64_t)C * ncols;
    float t1[MS][TS];  // A^T M  (columns)
#pragma unroll
    for (int q = 0; q < TS; ++q) {
        float colv[TS], y[MS];
#pragma unroll
        for (int r = 0; r < TS; ++r) colv[r] = src[(int64_t)(r * TS + q) * tstride];
        wino_at<TS, MS>(colv, y);
#pragma unroll
        for (int r = 0; r < MS; ++r) t1[r][q] = y[r];
    }
    const float b = bias ? bias[c] : 0.0f;
    const int64_t base = ((n * C + c) * (int64_t)H + ty * MS) * W + tx * MS;
#pragma unroll
    for (int r = 0; r < MS; ++r) {
        float y[MS];
        wino_at<TS, MS>(t1[r], y);
        const int64_t o = base + (int64_t)r * W;
#pragma unroll
        for (int q = 0; q < MS; ++q) {
            float v = y[q] + b;
            if (res) v += res[o + q];
            if (sum_out) sum_out[o + q] = v;
            if (act_out) act_out[o + q] = elu1(v);
        }
    }
}

// floats per LDS plane of k_wino_fused: (H + 4) rows of W + 4, + 4 for the last row's right halo, rounded up to 256 bytes
__host__ __device__ __forceinline__ constexpr int fused_lp(int H, int W) { return ((H + 4) * (W + 4) + 4 + 63) & ~63; }

__device__ __forceinline__ void lds_wave_sync() {
    // a wavefront's LDS operations execute in issue order: all that is needed is that the compiler keeps them in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the TS x TS window whose 4x4 centre starts at `mine` (a 16-byte aligned interior position), PAD = (TS - 4) / 2
// WHOLE (k_wino_fused): the empty asm statements keep every 16-byte read whole.  Left alone, hipcc fetches exactly the TS x TS
// dwords it needs as ds_read2_b32 pairs -- 32-lane groups whose lanes sit 4 dwords apart, on 8 of the 32 banks: a 4-way conflict
// on every read.  k_conv3_wino keeps the dword reads (WHOLE = false): whole rows cost it 13-38 registers, and at 78 two of its
// wavefronts fit beside the persistent GEMM's on a SIMD -- with whole reads it was 8 us faster alone and cost the 1000-chain
// step 7 ms (round 4, visit Y).
typedef float lds_f4 __attribute__((ext_vector_type(4)));
template <int TS, bool WHOLE = true>
__device__ __forceinline__ void lds_window(const float* mine, int LW, float (&p)[TS][TS]) {
    constexpr int PAD = (TS - 4) / 2;
#pragma unroll
    for (int r = 0; r < TS; ++r) {
        const float* row = mine + (r - PAD) * LW;
        float ax, ay, az, aw, mx, my, mz, mw, zx, zy;
        if (WHOLE) {
            lds_f4 a = *reinterpret_cast<const lds_f4*>(row - 4);
            lds_f4 m = *reinterpret_cast<const lds_f4*>(row);
            lds_f4 z = *reinterpret_cast<const lds_f4*>(row + 4);
            asm volatile("" : "+v"(a), "+v"(m), "+v"(z));
            __builtin_amdgcn_sched_barrier(0);          // one row (12 registers) in flight, not TS: these kernels sit beside the GEMM
            ax = a.x, ay = a.y, az = a.z, aw = a.w, mx = m.x, my = m.y, mz = m.z, mw = m.w, zx = z.x, zy = z.y;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(row - 4);
            const float4 m = *reinterpret_cast<const float4*>(row);
            const float4 z = *reinterpret_cast<const float4*>(row + 4);
            ax = a.x, ay = a.y, az = a.z, aw = a.w, mx = m.x, my = m.y, mz = m.z, mw = m.w, zx = z.x, zy = z.y;
        }
        (void)ax; (void)ay; (void)zy;
        if (PAD == 1) {
            p[r][0] = aw; p[r][1] = mx; p[r][2] = my; p[r][3] = mz; p[r][4] = mw; p[r][TS - 1] = zx;
        } else {
            p[r][0] = az; p[r][1] = aw; p[r][2] = mx; p[r][3] = my; p[r][4] = mz; p[r][5] = mw;
            p[r][(TS == 8) ? 6 : 0] = zx; p[r][(TS == 8) ? 7 : 0] = zy;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_wino_fused<TS_IN, TS_OUT>: everything between two batched GEMMs of a ResNet layer in ONE pass.
//   source   TS_IN == 0: x [N,C,H,W]              TS_IN in {6,8}: M [TS_IN^2, C, N*T] -> A^T M A
//   s = source + bias[c] (+ res);  sum_out = s (nullable);  a = (act & 1) ? ELU(s) : s;  act_out = a (nullable)
//   target   TS_OUT in {6,8}: V [TS_OUT^2, C, N*T] <- B^T a' B of the TS_OUT x TS_OUT windows (0 outside),
//            a' = (act & 2) ? ELU(a) : a
// Tile stride 4 on both sides (F(4x4,3x3): TS 6, F(4x4,5x5): TS 8).  A workgroup owns one channel of
// 256/T consecutive images (T = tiles per plane: 16 for 16x16): thread = (image, tile), so the column
// index n*T + tile is consecutive across the block -- every M read and V write is a contiguous 1-KB
// row -- and the activated planes meet in LDS (zero halo of 2) where the overlapping windows of the
// forward transform are cut out with 16-byte reads.  Nothing pointwise is left outside: a layer
// x + conv2(ELU(conv1(ELU(x)) + b1)) + b2 is  GEMM, fused, GEMM, fused.
// ------------------------------------------------------------------------------------------
// M is read once (written by the GEMM in front) and V written once (read by the GEMM behind): streaming loads / stores.
// Alone at 400 blocks x 256 channels: <6,6> 114 -> 105 us (5.6 TB/s), with the residual sum 137 -> 113, <8,8> 179 -> 158
// (6.0 TB/s), tools/probes/fused_probe.py, round 3
constexpr bool FUSED_NT = true;
// (Round 6, r06o: the <6,6> flavour capped at 64 registers -- amdgpu_waves_per_eu(8): no spill -- fits beside a workgroup of the
// bf16x3 GEMM (2 x 224 of a SIMD's 512 registers), its natural 66 do not: 160.0 against 159.0 ms per step.  Co-residency with the
// GEMM is not what this kernel lacks; not kept.)
template <int TS_IN, int TS_OUT>
__global__ __launch_bounds__(256) void k_wino_fused(const float* __restrict__ src, const float* __restrict__ bias,
                                                    const float* __restrict__ res, float* __restrict__ sum_out,
                                                    float* __restrict__ act_out, float* __restrict__ V, int64_t N,
                                                    int C, int H, int W, int act) {
    if (BS_XFORM_PRIO) __builtin_amdgcn_s_setprio(BS_XFORM_PRIO);
    // LDS planes as in k_conv3_wino (round 4, visit Y): rows W + 4 apart -- a row's right halo IS the next row's left halo -- and
    // the window cut out with three aligned 16-byte reads per row.  Round 3 read it dword by dword from rows W + 8 apart: the 32
    // lanes of a ds_read_b32 group then sit 4 dwords apart on rows and planes that are multiples of 32 dwords apart, i.e. on 4 of
    // the 32 banks (SQ_LDS_BANK_CONFLICT 81 % of SQ_LDS_IDX_ACTIVE, profiles/r04s).  With 16 W + 64 bytes per tile row and plane
    // strides that are multiples of 256 bytes, the 16 lanes of a ds_read_b128 group cover the 64 banks once.
    extern __shared__ float lds[];                  // [IMG][fused_lp(H, W)], zero halo
    const int ntx = W / 4, T = (H / 4) * ntx;       // tiles per plane, divides 256
    const int IMG = 256 / T;
    const int LW = W + 4, LP = fused_lp(H, W);      // padded row / plane size
    const int tid = threadIdx.x;
    const int img = tid / T, tile = tid - img * T;
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int c = blockIdx.x;
    const int64_t n = (int64_t)blockIdx.y * IMG + img;
    const int64_t ncols = N * T;
    const int64_t col = n * T + tile;
    const bool live = n < N;
    if (TS_OUT) {
        for (int k = tid * 4; k < IMG * LP; k += 1024) *reinterpret_cast<float4*>(lds + k) = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    float v[4][4];
    const int64_t pbase = ((n * C + c) * (int64_t)H + ty * 4) * W + tx * 4;  // this tile in an [N,C,H,W] tensor
    if (live) {
        if (TS_IN == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 t = *reinterpret_cast<const float4*>(src + pbase + (int64_t)r * W);
                v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
            }
        } else {
            constexpr int TI = TS_IN ? TS_IN : 6;
            const float* m = src + (int64_t)c * ncols + col;
            const int64_t tstride = (int64_t)C * ncols;
            float t1[4][TI];  // A^T M (columns)
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                float colv[TI], y[4];
#pragma unroll
                for (int r = 0; r < TI; ++r) colv[r] = (FUSED_NT && !(act & 4)) ? __builtin_nontemporal_load(m + (int64_t)(r * TI + q) * tstride) : m[(int64_t)(r * TI + q) * tstride];
                wino_at<TI, 4>(colv, y);
#pragma unroll
                for (int r = 0; r < 4; ++r) t1[r][q] = y[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) wino_at<TI, 4>(t1[r], v[r]);
        }
        const float b = bias ? bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 rr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (res) rr = *reinterpret_cast<const float4*>(res + pbase + (int64_t)r * W);
            v[r][0] = (v[r][0] + b) + rr.x; v[r][1] = (v[r][1] + b) + rr.y;   // same association as k_wino_out
            v[r][2] = (v[r][2] + b) + rr.z; v[r][3] = (v[r][3] + b) + rr.w;
            if (sum_out) *reinterpret_cast<float4*>(sum_out + pbase + (int64_t)r * W) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
            if (act & 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[r][q] = elu1(v[r][q]);
            }
            if (act_out) *reinterpret_cast<float4*>(act_out + pbase + (int64_t)r * W) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
            if (act & 2) {  // act_out is the residual stream of a block whose first conv sees ELU of it again
#pragma unroll
                for (int q = 0; q < 4; ++q) v[r][q] = elu1(v[r][q]);
            }
            if (TS_OUT)
                *reinterpret_cast<float4*>(lds + img * LP + (ty * 4 + r + 2) * LW + tx * 4 + 4) =
                    make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
        }
    }
    if (TS_OUT) {
        constexpr int TO = TS_OUT ? TS_OUT : 6;
        __syncthreads();
        if (!live) return;
        float d[TO][TO], t1[TO][TO];
        lds_window<TO>(lds + img * LP + (ty * 4 + 2) * LW + tx * 4 + 4, LW, d);   // the tile's interior position, 16-byte aligned
#pragma unroll
        for (int q = 0; q < TO; ++q) {               // B^T d, column by column
            float colv[TO], o[TO];
#pragma unroll
            for (int r = 0; r < TO; ++r) colv[r] = d[r][q];
            wino_bt<TO>(colv, o);
#pragma unroll
            for (int r = 0; r < TO; ++r) t1[r][q] = o[r];
        }
        float* out = V + (int64_t)c * ncols + col;
        const int64_t tstride = (int64_t)C * ncols;
#pragma unroll
        for (int r = 0; r < TO; ++r) {
            float o[TO];
            wino_bt<TO>(t1[r], o);
#pragma unroll
            for (int q = 0; q < TO; ++q) {
                if (FUSED_NT && !(act & 4)) __builtin_nontemporal_store(o[q], out + (int64_t)(r * TO + q) * tstride);
                else out[(int64_t)(r * TO + q) * tstride] = o[q];
            }
        }
    }
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_rc() { return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <bool R, bool SUM, bool ACT>
int launch_bre(const float* x, const float* bias, const float* res, float* sum_out, float* act_out, int64_t N, int C,
               int HW, hipStream_t st) {
    const int64_t total = N * C * HW;
    const bool vec = (HW % 4 == 0) && aligned16(x) && (!R || aligned16(res)) && (!SUM || aligned16(sum_out)) &&
                     (!ACT || aligned16(act_out));
    if (vec) {
        const int64_t n4 = total / 4;
        hipLaunchKernelGGL((k_bias_res_elu<R, SUM, ACT>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, bias,
                           res, sum_out, act_out, n4, C, HW / 4);
    } else {
        hipLaunchKernelGGL((k_bias_res_elu_1<R, SUM, ACT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x,
                           bias, res, sum_out, act_out, total, C, HW);
    }
    return launch_rc();
}


// ------------------------------------------------------------------------------------------
// k_conv3_wino<TS_OUT>: the INPUT convolution of a stack fused with everything up to the first GEMM of the block that
// follows it: h = ELU(conv3x3(x, w) + b) for Cin = zchannels input planes (8: a dozen KB of weights, 2 GMAC per 400
// blocks -- not matrix-core work), act_out = h, V = B^T ELU(h) B.  Replaces an MIOpen launch (243 us at 400 blocks) plus
// k_wino_fused<0, TS_OUT> (112 us) and the round trip of the conv output between them; what is left is the write of h
// and V (341 MB at 400 blocks).
//   block = the 64 / T images of one wavefront (T tiles of 4x4 per plane) x `cpb` output channels.  ALL Cin planes of
//   those images are staged once in LDS (zero halo; rows are W + 4 apart, so a row's right halo IS the next row's left
//   halo and a thread's three aligned 16-byte reads per window row are bank-conflict free), then each of the 4
//   wavefronts walks its share of the channels two at a time: the 6x6 window of a plane is read once for both, the
//   2 x 9 weights are wave-uniform (scalar loads), no block barrier inside the channel loop.  The activated tile goes
//   through a wave-private LDS tile (LDS is in order within a wavefront) to reach the neighbours' windows for B^T d B.
// ------------------------------------------------------------------------------------------
template <int TS_OUT>
__device__ __forceinline__ void conv3_tail(float (&v)[4][4], float b, int act, bool live, float* __restrict__ act_plane,
                                           float* __restrict__ vout, int64_t tstride, float* smine, int LW, int W) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[r][q] += b;
            if (act & 1) v[r][q] = elu1(v[r][q]);
        }
        if (act_plane && live) *reinterpret_cast<float4*>(act_plane + r * W) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
        if (act & 2) {  // act_out is the residual stream of a block whose first conv sees ELU of it again
#pragma unroll
            for (int q = 0; q < 4; ++q) v[r][q] = elu1(v[r][q]);
        }
        *reinterpret_cast<float4*>(smine + r * LW) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
    }
    lds_wave_sync();
    float d[TS_OUT][TS_OUT], t1[TS_OUT][TS_OUT];
    lds_window<TS_OUT, false>(smine, LW, d);
    lds_wave_sync();                                 // the tile may be overwritten by the next channel from here on
    if (!live) return;
#pragma unroll
    for (int q = 0; q < TS_OUT; ++q) {               // B^T d, column by column
        float colv[TS_OUT], o[TS_OUT];
#pragma unroll
        for (int r = 0; r < TS_OUT; ++r) colv[r] = d[r][q];
        wino_bt<TS_OUT>(colv, o);
#pragma unroll
        for (int r = 0; r < TS_OUT; ++r) t1[r][q] = o[r];
    }
#pragma unroll
    for (int r = 0; r < TS_OUT; ++r) {
        float o[TS_OUT];
        wino_bt<TS_OUT>(t1[r], o);
#pragma unroll
        for (int q = 0; q < TS_OUT; ++q) vout[(int64_t)(r * TS_OUT + q) * tstride] = o[q];
    }
}

template <int TS_OUT>
__global__ __launch_bounds__(256) void k_conv3_wino(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ act_out,
                                                    float* __restrict__ V, int64_t N, int Cin, int C, int H, int W, int act,
                                                    int cpb) {
    if (BS_XFORM_PRIO) __builtin_amdgcn_s_setprio(BS_XFORM_PRIO);
    extern __shared__ float lds[];
    const int ntx = W / 4, T = (H / 4) * ntx;
    const int IMG = 64 / T;                          // images of one wavefront = images of the block
    const int LW = W + 4, LP = (H + 4) * LW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = lane / T, tile = lane - img * T;
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int64_t n0 = (int64_t)blockIdx.x * IMG;
    const int64_t n = n0 + img;
    const bool live = n < N;
    const int nin = IMG * Cin * LP + 4;              // input planes (+ the last row's right halo)
    const int LPS = fused_lp(H, W);                  // plane stride of the activation tiles: a multiple of 256 bytes (bank-clean)
    const int nscr = IMG * LPS + 4;                  // one wavefront's activation tile
    for (int k = tid * 4; k < nin + 4 * nscr; k += 1024) *reinterpret_cast<float4*>(lds + k) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    {   // x[n0 .. n0+IMG) is one contiguous run of IMG*Cin*H*W floats
        const int w4 = W / 4, per_plane4 = H * w4;
        const int64_t avail = (N - n0 < IMG ? N - n0 : IMG) * (int64_t)Cin * per_plane4;
        const float4* xs = reinterpret_cast<const float4*>(x + n0 * Cin * (int64_t)H * W);
        for (int k = tid; k < avail; k += 256) {
            const int pl = k / per_plane4, rem = k - pl * per_plane4;
            const int row = rem / w4, c4 = rem - row * w4;
            *reinterpret_cast<float4*>(lds + pl * LP + (row + 2) * LW + 4 + c4 * 4) = xs[k];
        }
    }
    __syncthreads();
    const int toff = (ty * 4 + 2) * LW + tx * 4 + 4;  // this tile's interior position (16-byte aligned)
    const float* tin = lds + img * Cin * LP + toff;
    float* smine = lds + nin + wave * nscr + img * LPS + toff;
    const int64_t ncols = N * T, col = n * T + tile;
    const int64_t tstride = (int64_t)C * ncols;
    const int cbase = blockIdx.y * cpb;
    const int cend = cbase + cpb < C ? cbase + cpb : C;
    for (int c0 = cbase + 2 * wave; c0 < cend; c0 += 8) {
        const bool two = c0 + 1 < cend;
        const int c1 = two ? c0 + 1 : c0;
        float v0[4][4], v1[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) { v0[r][q] = 0.0f; v1[r][q] = 0.0f; }
        const float* w0 = w + (int64_t)c0 * Cin * 9;
        const float* w1 = w + (int64_t)c1 * Cin * 9;
        for (int ci = 0; ci < Cin; ++ci) {
            float p[6][6];
            lds_window<6, false>(tin + ci * LP, LW, p);
            float k0[9], k1[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { k0[k] = w0[ci * 9 + k]; k1[k] = w1[ci * 9 + k]; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v0[r][q] = fmaf(k0[ky * 3 + kx], p[r + ky][q + kx], v0[r][q]);
                            v1[r][q] = fmaf(k1[ky * 3 + kx], p[r + ky][q + kx], v1[r][q]);
                        }
        }
        const int64_t pix = (int64_t)(ty * 4) * W + tx * 4;
        conv3_tail<TS_OUT>(v0, bias ? bias[c0] : 0.0f, act, live,
                           act_out ? act_out + (n * C + c0) * (int64_t)H * W + pix : nullptr,
                           V + (int64_t)c0 * ncols + col, tstride, smine, LW, W);
        if (two)
            conv3_tail<TS_OUT>(v1, bias ? bias[c1] : 0.0f, act, live,
                               act_out ? act_out + (n * C + c1) * (int64_t)H * W + pix : nullptr,
                               V + (int64_t)c1 * ncols + col, tstride, smine, LW, W);
    }
}

// ------------------------------------------------------------------------------------------
// k_small_k_gemm: M[t] = U[t] x V[t] for the INPUT convolutions of the stacks in the Winograd domain -- Cin = zchannels
// or 4 x image channels (8, 12: the reduction is a dozen terms), Cout = ResNet width.  The product is a write of
// M [T, Cout, cols] and nothing else worth a matrix core; a library GEMM spends 4-5x the time of that write on it
// (profiles/r02g).  Thread = 4 consecutive columns x CO output channels; V rows arrive as 16-byte loads, U through
// the scalar cache (wave-uniform), M leaves as 16-byte stores.
// ------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void k_small_k_gemm(const float* __restrict__ U, const float* __restrict__ V,
                                                      float* __restrict__ M, int Cout, int Cin, int64_t cols) {
    const int t = blockIdx.z;
    const int co0 = blockIdx.y * CO;
    const int64_t c4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c4 >= cols) return;
    const float* u = U + ((int64_t)t * Cout + co0) * Cin;
    const float* v = V + (int64_t)t * Cin * cols + c4;
    float4 acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 x = *reinterpret_cast<const float4*>(v + (int64_t)ci * cols);
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const float w = (co0 + o < Cout) ? u[o * Cin + ci] : 0.f;
            acc[o].x = fmaf(w, x.x, acc[o].x);
            acc[o].y = fmaf(w, x.y, acc[o].y);
            acc[o].z = fmaf(w, x.z, acc[o].z);
            acc[o].w = fmaf(w, x.w, acc[o].w);
        }
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int o = 0; o < CO; ++o)
        if (co0 + o < Cout) {
            v4f r;
            r.x = acc[o].x; r.y = acc[o].y; r.z = acc[o].z; r.w = acc[o].w;
            *reinterpret_cast<v4f*>(M + ((int64_t)t * Cout + co0 + o) * cols + c4) = r;   // read next by the fused pass: keep it cacheable
        }
}

#ifdef BS_CONV3_MFMA
// LAB BUILD ONLY (-DBS_CONV3_MFMA; VERDICT r3 #6 / r5 #5, visit r06w): round 3's matrix-core version of the input convolution
// (profiles/archive/conv3_mfma.hip.txt), spliced back in to be measured in today's pipeline.  Sums in k pairs, so its bits differ
// from k_conv3_wino's ci -> ky -> kx chain: a stream written by such a build is not a stream of the product route.

// ------------------------------------------------------------------------------------------
// k_conv3_mfma<CIN>: the same fused input convolution with its 9 CIN-deep products on the matrix cores, for the shape the
// conv stacks have (16 x 16 planes, F(4x4,3x3) behind it, C a multiple of 32).  k_conv3_wino does 2 x 72 FMAs per output on
// the fp32 VALU: 177-200 us per 500-block launch against a write floor of 85 (h 131 MB + V 295 MB), +79 % beside a serial
// coder kernel.  Here a wavefront owns ONE image of the block's four and 32 output channels at a time:
//   GEMM   [32 channels x 72] x [72 x 256 pixels]: v_mfma_f32_32x32x2_f32, A = the weights (lane: channel l % 32,
//          k = 2s + l / 32: 36 registers per 32-channel block), B = the im2col of the image's zero-haloed planes read
//          straight from LDS (lane: pixel 32t + l % 32 of tile-row pair t, tap k = ci 9 + ky 3 + kx: one ds_read_b32
//          per MFMA), 8 accumulator tiles = the whole plane of 32 channels in 128 registers;
//   tail   four channels at a time (a half-wave's registers 4j .. 4j+3): + bias, ELU, scattered into four haloed LDS
//          planes of the wavefront; h leaves from there as four contiguous 1-KB planes (16-byte stores); lane = (plane,
//          tile) cuts its 6 x 6 window, ELU again, B^T d B, 36 stores (64-byte runs per plane and position; the three
//          other wavefronts of the block write the neighbouring runs of the same 256-byte row).
// Sum per output: k ascending in pairs (2s, 2s + 1) -- fixed, independent of the batch.
// ------------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int CIN>
__global__ __launch_bounds__(256, 2) void k_conv3_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ act_out,
                                                       float* __restrict__ V, int64_t N, int C, int act, int cpb) {
    constexpr int H = 16, W = 16, LW = W + 4, LP = (H + 4) * LW, T = 16, KS = CIN * 9 / 2;
    static_assert((CIN * 9) % 2 == 0, "k pairs");
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, g = lane >> 5;
    const int64_t n0 = (int64_t)blockIdx.x * 4;
    constexpr int nin = 4 * CIN * LP + 4, nscr = 4 * LP + 4;
    for (int k = tid * 4; k < nin + 4 * nscr; k += 1024) *reinterpret_cast<float4*>(lds + k) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    {   // x[n0 .. n0+4) is one contiguous run of 4*CIN*H*W floats
        constexpr int w4 = W / 4, per_plane4 = H * w4;
        const int64_t avail = (N - n0 < 4 ? N - n0 : 4) * (int64_t)CIN * per_plane4;
        const float4* xs = reinterpret_cast<const float4*>(x + n0 * CIN * (int64_t)H * W);
        for (int k = tid; k < avail; k += 256) {
            const int pl = k / per_plane4, rem = k - pl * per_plane4;
            const int row = rem / w4, c4 = rem - row * w4;
            *reinterpret_cast<float4*>(lds + pl * LP + (row + 2) * LW + 4 + c4 * 4) = xs[k];
        }
    }
    __syncthreads();
    const int64_t n = n0 + wave;
    if (n >= N) return;                              // (no block barrier below: the wavefronts are on their own from here)
    // im2col: pixel (y, x) = (2t + l32 / 16, l32 % 16) of tile-row pair t, tap (ci, ky, kx) -> plane ci, row y + ky + 1,
    // column x + kx + 3 of the haloed layout (interior at row + 2, column + 4)
    const float* xin = lds + wave * CIN * LP + ((l32 >> 4) + 1) * LW + (l32 & 15) + 3;
    float* scr = lds + nin + wave * nscr;
    const int64_t ncols = N * T, tstride = (int64_t)C * ncols;
    const int cbase = blockIdx.y * cpb;
    const int cend = cbase + cpb < C ? cbase + cpb : C;
    // transform role of this lane: plane q = lane / 16 of a round, tile = lane % 16
    const int tq = lane >> 4, tile = lane & 15, ty = tile >> 2, tx = tile & 3;
    const float* smine = scr + tq * LP + (ty * 4 + 2) * LW + tx * 4 + 4;
    for (int cb = cbase; cb < cend; cb += 32) {
        float wreg[KS];
        {
            const float* wr = w + (int64_t)(cb + l32) * (CIN * 9) + g;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) wreg[s2] = wr[2 * s2];
        }
        f32x16_t acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.0f;
        float b[2][8];
        auto tap = [&](int s2) {                      // LDS offset of tap k = 2 s2 + g against xin (s2 is a compile-time value)
            const int k0 = 2 * s2, k1 = 2 * s2 + 1;
            const int o0 = (k0 / 9) * LP + ((k0 % 9) / 3) * LW + (k0 % 3), o1 = (k1 / 9) * LP + ((k1 % 9) / 3) * LW + (k1 % 3);
            return g ? o1 : o0;
        };
        {
            const float* p = xin + tap(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) b[0][t] = p[t * 2 * LW];
        }
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(b[s2 & 1][t]));        // (the compiler's wait lands before the next reads)
            __builtin_amdgcn_sched_barrier(0);
            if (s2 + 1 < KS) {
                const float* p = xin + tap(s2 + 1);
#pragma unroll
                for (int t = 0; t < 8; ++t) b[(s2 + 1) & 1][t] = p[t * 2 * LW];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s2], b[s2 & 1][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // register v of lane l: channel cb + (v/4)*8 + (l/32)*4 + v%4, pixel 32 t + l%32
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int c0 = cb + 8 * j + 4 * gg;          // the round's four channels c0 .. c0 + 3 (wave-uniform)
                if (g == gg) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float bq = bias ? bias[c0 + q] : 0.0f;
                        float* pl = scr + q * LP + ((l32 >> 4) + 2) * LW + 4 + (l32 & 15);
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            float v = acc[t][4 * j + q] + bq;
                            if (act & 1) v = elu1(v);
                            pl[t * 2 * LW] = v;
                        }
                    }
                }
                lds_wave_sync();
                if (act_out) {                           // four contiguous 1-KB planes: lane -> row lane / 4, columns 4 (lane % 4) ..
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 hv = *reinterpret_cast<const float4*>(scr + q * LP + ((lane >> 2) + 2) * LW + 4 + (lane & 3) * 4);
                        *reinterpret_cast<float4*>(act_out + ((n * C + c0 + q) * (int64_t)(H * W)) + lane * 4) = hv;
                    }
                }
                float d[6][6], t1[6][6];
                lds_window<6, false>(smine, LW, d);
                lds_wave_sync();                         // the planes may be overwritten by the next round from here on
                if (act & 2) {
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int q = 0; q < 6; ++q) d[r][q] = elu1(d[r][q]);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {            // B^T d, column by column
                    float colv[6], o[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) colv[r] = d[r][q];
                    wino_bt<6>(colv, o);
#pragma unroll
                    for (int r = 0; r < 6; ++r) t1[r][q] = o[r];
                }
                float* vout = V + (int64_t)(c0 + tq) * ncols + n * T + tile;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    float o[6];
                    wino_bt<6>(t1[r], o);
#pragma unroll
                    for (int q = 0; q < 6; ++q) vout[(int64_t)(r * 6 + q) * tstride] = o[q];
                }
            }
        }
    }
}


#endif

template <int TS_OUT>
int launch_conv3(const float* x, const float* w, const float* bias, int act, float* act_out, float* V, int64_t N, int Cin,
                 int C, int H, int W, int cpb, dim3 grid, size_t shm, hipStream_t st) {
    static std::atomic<size_t> granted[64];          // dynamic LDS above 64 KB has to be granted per kernel and device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return BS_ELAUNCH;
    std::atomic<size_t>& g = granted[dev & 63];
    if (shm > g.load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_wino<TS_OUT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return BS_EUNSUPPORTED;
        g.store(shm, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((k_conv3_wino<TS_OUT>), grid, dim3(256), shm, st, x, w, bias, act_out, V, N, Cin, C, H, W, act, cpb);
    return launch_rc();
}

}  // namespace

extern "C" {

int bs_bias_residual_elu_f32(const float* x, const float* bias, const float* res, float* sum_out, float* act_out,
                             int64_t N, int C, int HW, void* stream) {
    if (!x || (!sum_out && !act_out) || N < 0 || C < 1 || HW < 1) return BS_EINVAL;
    if (N == 0) return BS_OK;
    hipStream_t st = S(stream);
    const bool r = res != nullptr, s = sum_out != nullptr, a = act_out != nullptr;
    if (r && s && a) return launch_bre<true, true, true>(x, bias, res, sum_out, act_out, N, C, HW, st);
    if (r && s) return launch_bre<true, true, false>(x, bias, res, sum_out, act_out, N, C, HW, st);
    if (r && a) return launch_bre<true, false, true>(x, bias, res, sum_out, act_out, N, C, HW, st);
    if (s && a) return launch_bre<false, true, true>(x, bias, res, sum_out, act_out, N, C, HW, st);
    if (s) return launch_bre<false, true, false>(x, bias, res, sum_out, act_out, N, C, HW, st);
    return launch_bre<false, false, true>(x, bias, res, sum_out, act_out, N, C, HW, st);
}

int bs_expand_rows5_f32(const float* in, const float* bias, float* out, int64_t N, int C, int H, int W, int act,
                        void* stream) {
    if (!in || !out || N < 0 || C < 1 || H < 1 || W < 4 || (W % 4) != 0 || !aligned16(out)) return BS_EINVAL;
    const int64_t total4 = N * C * 5 * (H + 4) * (W / 4);
    if (total4 == 0) return BS_OK;
    hipLaunchKernelGGL(k_expand_rows5, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, S(stream), in, bias, out,
                       total4, C, H, W, act);
    return launch_rc();
}

static bool wino_cfg_ok(int ts, int ms, int H, int W) {
    return ((ts == 6 && (ms == 4 || ms == 2)) || (ts == 8 && ms == 4)) && H >= ms && W >= ms && H % ms == 0 && W % ms == 0;
}

int bs_conv3_wino_f32(const float* x, const float* w, const float* bias, int act, float* act_out, float* V, int ts_out,
                      int64_t N, int Cin, int C, int H, int W, void* stream) {
    if (!x || !w || !V || N < 0 || Cin < 1 || C < 1 || H < 4 || W < 4 || H % 4 || W % 4 || (ts_out != 6 && ts_out != 8))
        return BS_EINVAL;
    const int T = (H / 4) * (W / 4);
    if (T > 64 || 64 % T) return BS_EUNSUPPORTED;
    const int IMG = 64 / T;
    const size_t LP = (size_t)(H + 4) * (W + 4);
    const size_t shm = ((size_t)IMG * Cin * LP + 4 + 4 * ((size_t)IMG * fused_lp(H, W) + 4)) * sizeof(float);
    if (shm > 160 * 1024) return BS_EUNSUPPORTED;
    if (N == 0) return BS_OK;
    const int64_t groups = (N + IMG - 1) / IMG;
    // channels per block: as many as leave the launch a few blocks per CU slot (2 blocks fit a CU at 77 KB each)
    const int forced = [] { const char* e = getenv("BITSWAP_CONV3_CPB"); return e ? atoi(e) : 0; }();
    int cpb = 64;
    while (cpb > 8 && groups * ((C + cpb - 1) / cpb) < 1536) cpb /= 2;
    if (forced >= 8 && forced % 8 == 0) cpb = forced;
#ifdef BS_CONV3_MFMA
    if (ts_out == 6 && H == 16 && W == 16 && Cin == 8 && C % 32 == 0) {
        static const bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_mfma<8>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess; }();
        (void)once;
        const int cpm = cpb < 32 ? 32 : cpb;
        const size_t shm_m = ((size_t)4 * 8 * LP + 4 + 4 * (4 * LP + 4)) * sizeof(float);
        hipLaunchKernelGGL((k_conv3_mfma<8>), dim3((unsigned)((N + 3) / 4), (unsigned)((C + cpm - 1) / cpm)), dim3(256), shm_m, S(stream),
                           x, w, bias, act_out, V, N, C, act, cpm);
        return launch_rc();
    }
#endif
    dim3 grid((unsigned)groups, (unsigned)((C + cpb - 1) / cpb));
    if (ts_out == 6) return launch_conv3<6>(x, w, bias, act, act_out, V, N, Cin, C, H, W, cpb, grid, shm, S(stream));
    return launch_conv3<8>(x, w, bias, act, act_out, V, N, Cin, C, H, W, cpb, grid, shm, S(stream));
}

int bs_small_k_gemm_f32(const float* U, const float* V, float* M, int T, int Cout, int Cin, int64_t cols, void* stream) {
    if (!U || !V || !M || T < 0 || Cout < 1 || Cin < 1 || Cin > 64 || cols < 0 || cols % 4 != 0) return BS_EINVAL;
    if (T == 0 || cols == 0) return BS_OK;
    constexpr int CO = 8;
    dim3 grid((unsigned)((cols / 4 + 255) / 256), (unsigned)((Cout + CO - 1) / CO), (unsigned)T), block(256);
    hipLaunchKernelGGL((k_small_k_gemm<CO>), grid, block, 0, S(stream), U, V, M, Cout, Cin, cols);
    return launch_rc();
}

int bs_wino_in_f32(const float* in, const float* bias, float* V, int64_t N, int C, int H, int W, int ts, int ms,
                   int act, void* stream) {
    if (!in || !V || N < 0 || C < 1 || !wino_cfg_ok(ts, ms, H, W)) return BS_EINVAL;
    const int64_t ncols = N * (H / ms) * (W / ms), total = ncols * C;
    if (total == 0) return BS_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define BS_WIN(TS, MS) hipLaunchKernelGGL((k_wino_in<TS, MS>), grid, block, 0, S(stream), in, bias, V, total, C, H, W, ncols, act)
    if (ts == 8) BS_WIN(8, 4);
    else if (ms == 4) BS_WIN(6, 4);
    else BS_WIN(6, 2);
#undef BS_WIN
    return launch_rc();
}

int bs_wino_out_f32(const float* M, const float* bias, const float* res, float* sum_out, float* act_out, int64_t N,
                    int C, int H, int W, int ts, int ms, void* stream) {
    if (!M || (!sum_out && !act_out) || N < 0 || C < 1 || !wino_cfg_ok(ts, ms, H, W)) return BS_EINVAL;
    const int64_t ncols = N * (H / ms) * (W / ms), total = ncols * C;
    if (total == 0) return BS_OK;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define BS_WOUT(TS, MS)                                                                                       \
    hipLaunchKernelGGL((k_wino_out<TS, MS>), grid, block, 0, S(stream), M, bias, res, sum_out, act_out, total, C, H, W, \
                       ncols)
    if (ts == 8) BS_WOUT(8, 4);
    else if (ms == 4) BS_WOUT(6, 4);
    else BS_WOUT(6, 2);
#undef BS_WOUT
    return launch_rc();
}

int bs_wino_fused_f32(const float* src, int ts_in, const float* bias, const float* res, int act, float* sum_out,
                      float* act_out, float* V, int ts_out, int64_t N, int C, int H, int W, void* stream) {
    const bool in_ok = ts_in == 0 || ts_in == 6 || ts_in == 8, out_ok = ts_out == 0 || ts_out == 6 || ts_out == 8;
    if (!src || !in_ok || !out_ok || (ts_out ? !V : (!sum_out && !act_out)) || N < 0 || C < 1 || H < 4 || W < 4 ||
        H % 4 || W % 4)
        return BS_EINVAL;
    const int T = (H / 4) * (W / 4);
    if (T > 256 || 256 % T) return BS_EUNSUPPORTED;  // tiles per plane must divide the block
    if (!aligned16(src) || (res && !aligned16(res)) || (sum_out && !aligned16(sum_out)) || (act_out && !aligned16(act_out)))
        return BS_EINVAL;
    if (N == 0) return BS_OK;
    const int IMG = 256 / T;
    dim3 grid((unsigned)C, (unsigned)((N + IMG - 1) / IMG)), block(256);
    size_t shm = ts_out ? (size_t)IMG * fused_lp(H, W) * sizeof(float) : 0;
    // BITSWAP_FUSED_LDS_MIN = bytes (experiment, round 6; default: no claim): claim at least that much LDS per workgroup, i.e. fewer
    // workgroups of this HBM-bound kernel per CU (160 KB / claim) beside the other chain group's GEMM / table kernels.  Measured
    // with 64 KB (two per CU) on four boxes (profiles/r06p .. r06s): cifar8 at 1000 chains 161.7 -> 157.3, 161.0 -> 160.0,
    // 160.2 -> 152.7, 162.4 -> 161.3 ms per step; imagenet4 120.7 -> 122.8, 122.6 -> 126.6 (worse); cifar8 at 1500 chains
    // 222.6 -> 234.3, 233.8 -> 239.4 (worse); one workgroup per CU 164.6 .. 183.  A gain for one mix of kernels on some boxes,
    // a loss for the others: not a default.  BITSWAP_FUSED_LDS_WHICH = 66 | 88 restricts the claim to one flavour.  Same bits.
    size_t want = 0;
    if (const char* e = getenv("BITSWAP_FUSED_LDS_MIN")) want = (size_t)atol(e);
    if (const char* e = getenv("BITSWAP_FUSED_LDS_WHICH")) {
        if (atoi(e) != ts_in * 10 + ts_out) want = 0;
    }
    if (want > shm && want <= 160 * 1024) shm = want;
    act &= 3;
    if (const char* e = getenv("BITSWAP_FUSED_PLAIN")) {      // diagnostics: ordinary instead of nontemporal loads of M / stores of V
        if (atoi(e)) act |= 4;
    }
#define BS_WF(TI, TO)                                                                                             \
    do {                                                                                                          \
        if (shm > 64 * 1024) {                                                                                    \
            static bool raised = false;                                                                           \
            if (!raised) {                                                                                        \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_fused<TI, TO>),                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
                raised = true;                                                                                    \
            }                                                                                                     \
        }                                                                                                         \
        hipLaunchKernelGGL((k_wino_fused<TI, TO>), grid, block, shm, S(stream), src, bias, res, sum_out, act_out, V, N, C, \
                           H, W, act);                                                                            \
    } while (0)
    switch (ts_in * 10 + ts_out) {
        case 6: BS_WF(0, 6); break;
        case 8: BS_WF(0, 8); break;
        case 60: BS_WF(6, 0); break;
        case 66: BS_WF(6, 6); break;
        case 68: BS_WF(6, 8); break;
        case 80: BS_WF(8, 0); break;
        case 86: BS_WF(8, 6); break;
        case 88: BS_WF(8, 8); break;
        default: return BS_EINVAL;  // 0 -> 0 is bs_bias_residual_elu_f32
    }
#undef BS_WF
    return launch_rc();
}

int bs_head_params_f32(const float* x, const float* bias, float* mu, float* scale, int64_t N, int C, int HW, int mode,
                       void* stream) {
    if (!x || !bias || !mu || !scale || N < 0 || C < 1 || HW < 1 || (mode != BS_HEAD_SIGMOID && mode != BS_HEAD_SOFTPLUS))
        return BS_EINVAL;
    const int64_t total = N * C * HW;
    if (total == 0) return BS_OK;
    hipLaunchKernelGGL(k_head_params, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), x, bias, mu, scale,
                       total, C, HW, mode);
    return launch_rc();
}

}  // extern "C"
