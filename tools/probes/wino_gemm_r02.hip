// ROUND-2 version of bitswap_amd/csrc/wino_gemm.hip (tiled launch, 64x64 per wavefront), kept only as the A/B partner of
// tools/gemm_probe.py: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libgemm_r02.so tools/probes/wino_gemm_r02.hip
// wino_gemm.hip -- the batched product in the middle of a Winograd-domain convolution on the matrix cores:
//     M[t] [Cout, cols] = U[t] [Cout, Cin] x V[t] [Cin, cols],   t = 0 .. T-1 (36 or 64 transform positions),
// float32 in, float32 accumulate (v_mfma_f32_32x32x2_f32), row-major throughout (bitswap_amd/model.py::_res_wino;
// the reference runs the same convolutions as cuDNN calls inside utils/torch/modules.py:233-241).
//
// Why not leave it to the BLAS library: (i) a codec needs sender and receiver to add the same products in the same
// order, whatever the batch -- here an output element is the sum over ci in ONE fixed order that depends on nothing but
// Cin (no split-K, no shape-dependent kernel choice), so results are bitwise independent of `cols`, of the library
// version and of its heuristics; (ii) at the reference's own 100 experiments per call (cols = 1600) the library picks a
// 32x32 macro tile and runs at half the rate it reaches at 6400 columns (profiles/r02F).
//
// Tiling for CDNA4: workgroup = 64*WM x 128 outputs of one t, WM x 2 wavefronts of 64 x 64 (2 x 2 MFMA tiles of
// 32 x 32, 64 accumulator registers per lane).  With WM = 4 the workgroup spans all 256 output channels: V -- the big
// operand, cols x Cin x T floats -- is read from HBM exactly once, U[t] (256 KB) stays in L2.  K advances 16 at a time
// through a double-buffered LDS stage (one barrier per step); global loads of step k+1 are in flight while step k
// multiplies.  LDS layouts are chosen so that every read is conflict-free:
//   A: [row][20]  -- a lane reads 4 consecutive k of its row as one 16-byte load (rows 16 apart share banks, and those
//                    sit in different 16-lane phases of the load);
//   B: [k][136]   -- a lane reads one float per k; the two half-waves read rows 4 apart, 4 x 136 = 32 (mod 64) banks.
// The MFMA contraction index is permuted (half-wave g takes k = 8j + 4g + i in step i of chunk j) -- the same
// permutation on both operands, i.e. the same sum in another fixed order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bitswap_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G_BN = 128, G_BK = 16, G_LDA = G_BK + 4, G_LDB = G_BN + 8;

template <int WM>
__global__ __launch_bounds__(128 * WM) void k_wino_gemm(const float* __restrict__ U, const float* __restrict__ V,
                                                        float* __restrict__ M, int Cout, int Cin, int64_t cols) {
    constexpr int BM = 64 * WM, NT = 128 * WM;
    constexpr int STAGE = BM * G_LDA + G_BK * G_LDB;
    constexpr int NA = BM * 4 / NT;                  // float4 loads per thread and stage: A (= 2)
    constexpr int NB = (G_BK * G_BN / 4) / NT;       //                                    B (4 / WM)
    extern __shared__ float lds[];                   // [2][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, g = lane >> 5;
    const int t = blockIdx.z;
    const int co0 = blockIdx.y * BM;
    const int64_t n0 = (int64_t)blockIdx.x * G_BN;
    const float* Ut = U + (int64_t)t * Cout * Cin;
    const float* Vt = V + (int64_t)t * Cin * cols;
    float* Mt = M + (int64_t)t * Cout * cols;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;

    f32x4 ra[NA], rb[NB];
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * NT, row = e >> 2, kq = e & 3;
            ra[i] = (co0 + row < Cout) ? *reinterpret_cast<const f32x4*>(Ut + (int64_t)(co0 + row) * Cin + k0 + kq * 4)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * NT, k = e >> 5, c4 = e & 31;
            rb[i] = (n0 + c4 * 4 < cols) ? *reinterpret_cast<const f32x4*>(Vt + (int64_t)(k0 + k) * cols + n0 + c4 * 4)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_stage = [&](int buf) {
        float* As = lds + buf * STAGE;
        float* Bs = As + BM * G_LDA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + i * NT, row = e >> 2, kq = e & 3;
            *reinterpret_cast<f32x4*>(As + row * G_LDA + kq * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + i * NT, k = e >> 5, c4 = e & 31;
            *reinterpret_cast<f32x4*>(Bs + k * G_LDB + c4 * 4) = rb[i];
        }
    };

    const int nk = Cin / G_BK;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_stage((kt + 1) * G_BK);            // in flight under the multiplies below
        const float* As = lds + buf * STAGE + (wm * 64 + l32) * G_LDA + g * 4;
        const float* Bs = lds + buf * STAGE + BM * G_LDA + (g * 4) * G_LDB + wn * 64 + l32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 a[2];
            float b[2][4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * G_LDA + j * 8);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i) b[ni][i] = Bs[(j * 8 + i) * G_LDB + ni * 32];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][i], b[ni][i], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) store_stage(buf ^ 1);                   // the other buffer: last read before the previous barrier
        __syncthreads();
    }

    // C layout of the 32x32 MFMA: register v of lane l is row (v/4)*8 + (l/32)*4 + v%4, column l%32
    const bool all_rows = co0 + BM <= Cout;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int64_t col = n0 + wn * 64 + ni * 32 + l32;
            if (col >= cols) continue;
            const int row0 = co0 + wm * 64 + mi * 32 + g * 4;
            float* p = Mt + (int64_t)row0 * cols + col;
            if (all_rows) {
#pragma unroll
                for (int v = 0; v < 16; ++v) p[(int64_t)((v >> 2) * 8 + (v & 3)) * cols] = acc[mi][ni][v];
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (row0 + (v >> 2) * 8 + (v & 3) < Cout) p[(int64_t)((v >> 2) * 8 + (v & 3)) * cols] = acc[mi][ni][v];
            }
        }
}

template <int WM>
int launch_gemm(const float* U, const float* V, float* M, int T, int Cout, int Cin, int64_t cols, hipStream_t st) {
    constexpr int BM = 64 * WM;
    dim3 grid((unsigned)((cols + G_BN - 1) / G_BN), (unsigned)((Cout + BM - 1) / BM), (unsigned)T);
    const size_t shm = 2 * (size_t)(BM * G_LDA + G_BK * G_LDB) * sizeof(float);
    hipLaunchKernelGGL((k_wino_gemm<WM>), grid, dim3(128 * WM), shm, st, U, V, M, Cout, Cin, cols);
    return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
}

}  // namespace

extern "C" int bs_wino_gemm_f32(const float* U, const float* V, float* M, int T, int Cout, int Cin, int64_t cols,
                                void* stream) {
    if (!U || !V || !M || T < 0 || T > 65535 || Cout < 1 || Cin < 1 || cols < 0) return BS_EINVAL;
    if (Cin % G_BK != 0 || cols % 4 != 0) return BS_EUNSUPPORTED;
    if (((uintptr_t)U | (uintptr_t)V | (uintptr_t)M) & 15u) return BS_EINVAL;
    if (T == 0 || cols == 0) return BS_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // tallest workgroup the channel count asks for, halved while the launch would leave most CUs without a workgroup
    // (few columns: 13 chains are 208).  The summation order per output does not depend on the choice.
    int wm = Cout > 128 ? 4 : Cout > 64 ? 2 : 1;
    const int64_t ctiles = (cols + G_BN - 1) / G_BN;
    while (wm > 1 && ctiles * ((Cout + 64 * wm - 1) / (64 * wm)) * T < 384) wm /= 2;
    if (wm == 4) return launch_gemm<4>(U, V, M, T, Cout, Cin, cols, st);
    if (wm == 2) return launch_gemm<2>(U, V, M, T, Cout, Cin, cols, st);
    return launch_gemm<1>(U, V, M, T, Cout, Cin, cols, st);
}
