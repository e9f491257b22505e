// wino_gemm.hip -- the batched product in the middle of a Winograd-domain convolution on the matrix cores:
//     M[t] [Cout, cols] = U[t] [Cout, Cin] x V[t] [Cin, cols],   t = 0 .. T-1 (36 or 64 transform positions),
// float32 in, float32 accumulate (v_mfma_f32_32x32x2_f32), row-major throughout (bitswap_amd/model.py::_res_wino;
// the reference runs the same convolutions as cuDNN calls inside utils/torch/modules.py:233-241).
//
// Why not leave it to the BLAS library: a codec needs sender and receiver to add the same products in the same order,
// whatever the batch -- here an output element is the sum over ci in ONE fixed order that depends on nothing but Cin
// (no split-K, no shape-dependent kernel choice), so results are bitwise independent of `cols`, of Cout, of the
// library version and of its heuristics.  Since round 3 EVERY product of the conv stacks takes this kernel (any column
// count, the 16-channel head convolutions included): what a chain decodes to no longer depends on how many chains
// were coded next to it.
//
// Scheduling for CDNA4 (round 3): persistent workgroups over a flat list of work units.  A unit is (t, row tile,
// block of 32 columns); workgroup g of G owns the contiguous unit range [g Nu / G, (g+1) Nu / G) and walks it in
// chunks of up to four column blocks (a 32*NW x 128 output tile, NW wavefronts of 32 rows x 128 columns = 1 x 4 MFMA
// tiles, 64 accumulator registers per lane).  Work per workgroup differs by at most one 32-column block: the tiled
// round-2 launch (1800 tiles of 256 x 128 on 512 resident slots = 3.52 rounds at 400 chains) lost a quarter of the
// chip to its last round.  G = 2 workgroups per CU (58 KB of LDS each), and the logical order of the workgroups
// follows the XCD a workgroup lands on (blockIdx % 8), so that the workgroups sharing an L2 share their U[t].
// K advances 16 at a time through a double-buffered LDS stage (one barrier per step); the global loads of the next
// step -- of the next CHUNK at the end of a chunk -- are in flight while this step multiplies, so the loop never
// drains between tiles.  LDS layouts are chosen so that every read is conflict-free:
//   A: [row][20]  -- a lane reads 4 consecutive k of its row as one 16-byte load (the 16 lanes of a ds_read_b128
//                    phase hit 16 disjoint groups of 4 banks);
//   B: [k][136]   -- a lane reads one float per k; the 32 lanes of a phase read consecutive words.
// The MFMA contraction index is permuted (half-wave g takes k = 8j + 4g + i in step i of chunk j) -- the same
// permutation on both operands, i.e. the same sum in another fixed order (the order of the round-2 kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/bitswap_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// M is written once and read once, by the transform pass that follows: streaming stores keep it from evicting the U[t]
// tiles the XCD's other workgroups are about to re-read from L2 (+1.3 % at the bench's shape, profiles/r03v)
constexpr bool NT_STORE = true;
constexpr int G_BN = 128, G_BK = 16, G_LDA = G_BK, G_LDB = G_BN;      // LDS rows are unpadded: the tiles arrive by LDS-DMA

struct Chunk {
    int t, co0, cb, nb;     // transform position, first output row, first 32-column block, column blocks (1..4)
};

// Global -> LDS staging of one K step (A: BM x 16 of U[t], B: 16 x 128 of V[t]) by LDS-DMA (global_load_lds_dwordx4): no
// staging registers, no ds_write pass -- in the round-3 lab (tools/probes/gemm_lab.hip) the register-staged version of this
// kernel lost 6 % to its LDS stage stores and 7 % to the loads feeding them.  A wave-instruction writes 64 x 16 bytes to
// consecutive LDS addresses (wave-uniform base + lane x 16), so the LDS image is linear in the thread index and any
// permutation has to be applied on the SOURCE side:
//   A: 16-byte granule q = 4 row + slot holds k = 4 (slot ^ ((row >> 2) & 3)) .. + 3 of that row -- rows are 64 bytes apart
//      (no padding possible), and the XOR spreads the 16 rows a ds_read_b128 phase touches over all 64 banks;
//   B: [k][128] as it lies in memory; a lane reads one float per k, the 32 lanes of a phase consecutive words.
template <int NW, int MI>
struct Stager {
    static constexpr int BM = 32 * MI * NW, NT = 64 * NW;
    static constexpr int STAGE = BM * G_LDA + G_BK * G_LDB;
    static constexpr int NA = BM * 4 / NT;                  // 16-byte granules per thread and stage: A (2 MI)
    static constexpr int NB = (G_BK * G_BN / 4) / NT;       //                                         B (8 / NW)
    const float* U;
    const float* V;
    int Cout, Cin, tid, wbase;
    int64_t cols;
    uint32_t offA[NA], offB[NB];             // BYTE offsets of this thread's granules against the chunk's operand bases

    __device__ __forceinline__ void init(const float* U_, const float* V_, int Cout_, int Cin_, int64_t cols_, int tid_) {
        U = U_, V = V_, Cout = Cout_, Cin = Cin_, cols = cols_, tid = tid_;
        wbase = __builtin_amdgcn_readfirstlane(tid & ~63);        // first granule of this wavefront per load instruction
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT, row = q >> 2, kq = (q & 3) ^ ((row >> 2) & 3);
            offA[i] = (uint32_t)(row * Cin + kq * 4) * 4u;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) offB[i] = (uint32_t)(((tid + i * NT) >> 5) * (int)cols + ((tid + i * NT) & 31) * 4) * 4u;
    }
    // issue the DMA of step (c, k0) into LDS stage `buf`
    __device__ __forceinline__ void load(const Chunk& c, int k0, float* lds, int buf) const {
        const char* Ub = reinterpret_cast<const char*>(U + ((int64_t)c.t * Cout + c.co0) * Cin + k0);
        const char* Vb = reinterpret_cast<const char*>(V + ((int64_t)c.t * Cin + k0) * cols + (int64_t)c.cb * 32);
        const int rows_left = Cout - c.co0;
        const int cols_left = (int)min((int64_t)c.nb * 32, cols - (int64_t)c.cb * 32);
        float* As = lds + buf * STAGE;
        float* Bs = As + BM * G_LDA;
        // rows beyond Cout / columns beyond the chunk only feed outputs that are never stored: their loads are redirected
        // to a valid address (row 0 / column 0 of the tile) instead of being masked
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + i * NT;
            const uint32_t off = ((q >> 2) < rows_left) ? offA[i] : (uint32_t)(((q & 3) ^ ((q >> 4) & 3)) * 16);
            __builtin_amdgcn_global_load_lds(Ub + off, (lds_ptr_t)(As + (wbase + i * NT) * 4),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t off = (((tid + i * NT) & 31) * 4 < cols_left) ? offB[i] : offB[i] - (uint32_t)((tid + i * NT) & 31) * 16u;
            __builtin_amdgcn_global_load_lds(Vb + off, (lds_ptr_t)(Bs + (wbase + i * NT) * 4),
                                             16, 0, 0);
        }
    }
};

// One chunk of NBLK column blocks: the K loop (the stage of its first step is in LDS buffer `buf` and synchronised), then
// the stores.  The last step prefetches the first stage of the chunk that follows.  A wavefront owns MI x NBLK MFMA tiles
// (32 MI rows x 32 NBLK columns); ALL operand fragments of a K step are requested from LDS before its first MFMA, so the
// wave pays one LDS latency per 32 MI NBLK / 2 MFMAs instead of one per pair (round-3 visit A: the matrix pipe was busy
// 70 % of the time with the reads interleaved).
template <int NW, int MI, int NBLK>
__device__ __forceinline__ void run_chunk(Stager<NW, MI>& sg, float* lds, float* __restrict__ M, const Chunk& cur,
                                          const Chunk& nxt, bool more, int& buf, int wave, int l32, int g) {
    constexpr int BM = 32 * MI * NW, STAGE = Stager<NW, MI>::STAGE;
    f32x16 acc[MI][NBLK];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NBLK; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
    const int nk = sg.Cin / G_BK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool last = kt + 1 == nk;
        // DMA of the next step into the other buffer (its last reads completed before the previous barrier): in flight
        // under the multiplies below, landed by the barrier at the end of this step
        if (!last) sg.load(cur, (kt + 1) * G_BK, lds, buf ^ 1);
        else if (more) sg.load(nxt, 0, lds, buf ^ 1);        // ... the next chunk's first stage under this chunk's last
        const float* As = lds + buf * STAGE + (wave * 32 * MI + l32) * G_LDA;
        const float* Bs = lds + buf * STAGE + BM * G_LDA + (g * 4) * G_LDB + l32;
        const int sw = (l32 >> 2) & 3;                       // source-side swizzle of the A granules, see Stager
        f32x4 a[2][MI];
        float b[2][NBLK][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[j][mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * G_LDA + (((2 * j + g) ^ sw) << 2));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ni = 0; ni < NBLK; ++ni) b[j][ni][i] = Bs[(j * 8 + i) * G_LDB + ni * 32];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NBLK; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][mi][i], b[j][ni][i], acc[mi][ni], 0, 0, 0);
        __syncthreads();
        buf ^= 1;
    }
    // C layout of the 32x32 MFMA: register v of lane l is row (v/4)*8 + (l/32)*4 + v%4, column l%32
    float* Mt = M + (int64_t)cur.t * sg.Cout * sg.cols;
    const bool all_rows = cur.co0 + BM <= sg.Cout;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row0 = cur.co0 + (wave * MI + mi) * 32 + g * 4;
#pragma unroll
        for (int ni = 0; ni < NBLK; ++ni) {
            const int64_t col = (int64_t)cur.cb * 32 + ni * 32 + l32;
            if (col < sg.cols) {
                float* p = Mt + (int64_t)row0 * sg.cols + col;
                if (all_rows) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        if (NT_STORE) __builtin_nontemporal_store(acc[mi][ni][v], p + (int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols);
                        else p[(int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols] = acc[mi][ni][v];
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (row0 + (v >> 2) * 8 + (v & 3) < sg.Cout) p[(int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols] = acc[mi][ni][v];
                }
            }
        }
    }
}

template <int NW, int MI>
__global__ __launch_bounds__(64 * NW, 2) void k_wino_gemm(const float* __restrict__ U, const float* __restrict__ V,
                                                          float* __restrict__ M, int Cout, int Cin, int64_t cols,
                                                          int ncb, int nrt, int units) {
    constexpr int BM = 32 * MI * NW;
    extern __shared__ float lds[];                   // [2][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, g = lane >> 5;

    // logical workgroup index: the workgroups of one XCD (blockIdx % 8, round-robin dispatch) take consecutive ranges
    const int G = gridDim.x;
    int w = blockIdx.x;
    if ((G & 7) == 0) w = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    int u = __builtin_amdgcn_readfirstlane((int)((int64_t)w * units / G));
    const int uend = __builtin_amdgcn_readfirstlane((int)((int64_t)(w + 1) * units / G));
    if (u >= uend) return;

    const int per_t = nrt * ncb;
    auto decode = [&](int uu) {                      // wave-uniform by construction: say so (the division runs on the VALU)
        Chunk c;
        const int t = uu / per_t, r = uu - t * per_t, rt = r / ncb;
        c.t = __builtin_amdgcn_readfirstlane(t);
        c.co0 = __builtin_amdgcn_readfirstlane(rt * BM);
        c.cb = __builtin_amdgcn_readfirstlane(r - rt * ncb);
        c.nb = min(min(4, ncb - c.cb), uend - uu);
        return c;
    };

    Stager<NW, MI> sg;
    sg.init(U, V, Cout, Cin, cols, tid);
    Chunk cur = decode(u);
    sg.load(cur, 0, lds, 0);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int unext = u + cur.nb;
        const bool more = unext < uend;
        Chunk nxt = cur;
        if (more) nxt = decode(unext);
        if (cur.nb == 4) run_chunk<NW, MI, 4>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        else if (cur.nb == 3) run_chunk<NW, MI, 3>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);   // a range's ragged ends
        else if (cur.nb == 2) run_chunk<NW, MI, 2>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        else run_chunk<NW, MI, 1>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        if (!more) break;
        u = unext;
        cur = nxt;
    }
}

int cu_count() {
    // multiprocessor count of the current device (a device attribute, not library state)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess
        || n <= 0)
        n = 256;
    return n;
}

template <int NW, int MI>
int launch_gemm(const float* U, const float* V, float* M, int T, int Cout, int Cin, int64_t cols, hipStream_t st) {
    constexpr int BM = 32 * MI * NW;
    const int64_t ncb = (cols + 31) / 32, nrt = (Cout + BM - 1) / BM;
    const int64_t units = (int64_t)T * nrt * ncb;
    if (units > 0x7fffffff || (int64_t)Cin * cols > 0x7fffffff) return BS_EUNSUPPORTED;
    const size_t shm = 2 * (size_t)(BM * G_LDA + G_BK * G_LDB) * sizeof(float);
    // workgroups per CU: two of the 256-row shape (48 KB of LDS each, one wavefront per SIMD each, up to 256 registers per lane);
    // the one-wave shape of the head convolutions is bounded by its 22 KB of LDS
    const int per_cu = BM >= 128 ? 2 : BM == 64 ? 4 : 6;
    int64_t G = (int64_t)cu_count() * per_cu;
    if (const char* e = getenv("BITSWAP_GEMM_WGS_PER_CU")) {    // tuning only: the summation order does not depend on it
        const int v = atoi(e);
        if (v > 0) G = (int64_t)cu_count() * v;
    }
    if (G > units) G = units;
    hipLaunchKernelGGL((k_wino_gemm<NW, MI>), dim3((unsigned)G), dim3(64 * NW), shm, st, U, V, M, Cout, Cin, cols, (int)ncb,
                       (int)nrt, (int)units);
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
    // tallest workgroup the channel count fills (rows beyond Cout are redirected operands whose results are dropped); the
    // summation order per output does not depend on the choice
    if (Cout > 128) return launch_gemm<4, 2>(U, V, M, T, Cout, Cin, cols, st);      // 4 wavefronts of 64 rows x 128 columns
    if (Cout > 64) return launch_gemm<4, 1>(U, V, M, T, Cout, Cin, cols, st);
    if (Cout > 32) return launch_gemm<2, 1>(U, V, M, T, Cout, Cin, cols, st);
    return launch_gemm<1, 1>(U, V, M, T, Cout, Cin, cols, st);
}
