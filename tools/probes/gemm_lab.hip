// gemm_lab.hip -- where do the idle matrix-pipe cycles of bs_wino_gemm_f32 go?  The product kernel of
// bitswap_amd/csrc/wino_gemm.hip with pieces switched off (MODE bits: 1 no global loads / LDS stores, 2 no barriers,
// 4 no epilogue stores, 8 operands from registers instead of LDS), timed round-robin in long loops (the clock of an MI355X
// drifts with the load history: short back-to-back loops of different variants are not comparable).  Probe only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gemm_lab tools/probes/gemm_lab.hip && /tmp/gemm_lab [cols]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define BS_OK 0
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G_BN = 128, G_BK = 16, G_LDA = G_BK + 4, G_LDB = G_BN + 8;

struct Chunk {
    int t, co0, cb, nb;     // transform position, first output row, first 32-column block, column blocks (1..4)
};

// Global -> register -> LDS staging of one K step (A: BM x 16 of U[t], B: 16 x 128 of V[t]); the per-thread parts of the
// addresses are 32-bit element offsets against wave-uniform bases.
template <int NW, int MI>
struct Stager {
    static constexpr int BM = 32 * MI * NW, NT = 64 * NW;
    static constexpr int STAGE = BM * G_LDA + G_BK * G_LDB;
    static constexpr int NA = BM * 4 / NT;                  // float4 loads per thread and stage: A (2 MI)
    static constexpr int NB = (G_BK * G_BN / 4) / NT;       //                                    B (8 / NW)
    const float* U;
    const float* V;
    int Cout, Cin, tid;
    int64_t cols;
    int offA[NA], offB[NB];
    f32x4 ra[NA], rb[NB];

    __device__ __forceinline__ void init(const float* U_, const float* V_, int Cout_, int Cin_, int64_t cols_, int tid_) {
        U = U_, V = V_, Cout = Cout_, Cin = Cin_, cols = cols_, tid = tid_;
#pragma unroll
        for (int i = 0; i < NA; ++i) offA[i] = ((tid + i * NT) >> 2) * Cin + ((tid + i * NT) & 3) * 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) offB[i] = ((tid + i * NT) >> 5) * (int)cols + ((tid + i * NT) & 31) * 4;
    }
    __device__ __forceinline__ void load(const Chunk& c, int k0) {
        const float* Ub = U + ((int64_t)c.t * Cout + c.co0) * Cin + k0;
        const float* Vb = V + ((int64_t)c.t * Cin + k0) * cols + (int64_t)c.cb * 32;
        const int rows_left = Cout - c.co0;
        const int cols_left = (int)min((int64_t)c.nb * 32, cols - (int64_t)c.cb * 32);
        // rows beyond Cout / columns beyond the chunk only feed outputs that are never stored: their loads are
        // redirected to a valid address instead of being masked (a select would wait for the load right here)
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = *reinterpret_cast<const f32x4*>(Ub + ((((tid + i * NT) >> 2) < rows_left) ? offA[i] : offA[i] & 15));
#pragma unroll
        for (int i = 0; i < NB; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(Vb + ((((tid + i * NT) & 31) * 4 < cols_left) ? offB[i] : offB[i] - ((tid + i * NT) & 31) * 4));
    }
    __device__ __forceinline__ void consume() const {
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("" ::"v"(rb[i]));
    }
    __device__ __forceinline__ void store(float* lds, int buf) const {
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
    }
};

// One chunk of NBLK column blocks: the K loop (the stage of its first step is in LDS buffer `buf` and synchronised), then
// the stores.  The last step prefetches the first stage of the chunk that follows.  A wavefront owns MI x NBLK MFMA tiles
// (32 MI rows x 32 NBLK columns); ALL operand fragments of a K step are requested from LDS before its first MFMA, so the
// wave pays one LDS latency per 32 MI NBLK / 2 MFMAs instead of one per pair (round-3 visit A: the matrix pipe was busy
// 70 % of the time with the reads interleaved).
template <int NW, int MI, int NBLK, int MODE>
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
        if (!(MODE & 1) && !(MODE & 16)) { if (!last) sg.load(cur, (kt + 1) * G_BK);
        else if (more) sg.load(nxt, 0); }                      // ... the next chunk's first stage under this chunk's last
        const float* As = lds + buf * STAGE + (wave * 32 * MI + l32) * G_LDA + g * 4;
        const float* Bs = lds + buf * STAGE + BM * G_LDA + (g * 4) * G_LDB + l32;
        f32x4 a[2][MI];
        float b[2][NBLK][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[j][mi] = (MODE & 8) ? f32x4{1.f + kt, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4*>(As + mi * 32 * G_LDA + j * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ni = 0; ni < NBLK; ++ni) b[j][ni][i] = (MODE & 8) ? (float)(kt + ni + i) : Bs[(j * 8 + i) * G_LDB + ni * 32];
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
        if (!(MODE & 1) && !(MODE & 32)) { if (!last || more) sg.store(lds, buf ^ 1); }
        if (MODE & 32) sg.consume();
        if (!(MODE & 2)) __syncthreads();
        buf ^= 1;
    }
    if (MODE & 4) { float s = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NBLK; ++ni)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc[mi][ni][v];
        if (s == 1.2345f) M[0] = s;
        return; }
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
                    for (int v = 0; v < 16; ++v) p[(int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols] = acc[mi][ni][v];
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (row0 + (v >> 2) * 8 + (v & 3) < sg.Cout) p[(int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols] = acc[mi][ni][v];
                }
            }
        }
    }
}

template <int NW, int MI, int MODE>
__global__ __launch_bounds__(64 * NW, NW >= 8 ? 4 : 2) void k_lab(const float* __restrict__ U, const float* __restrict__ V,
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
    sg.load(cur, 0);
    sg.store(lds, 0);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int unext = u + cur.nb;
        const bool more = unext < uend;
        Chunk nxt = cur;
        if (more) nxt = decode(unext);
        if (cur.nb == 4) run_chunk<NW, MI, 4, MODE>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        else if (cur.nb == 3) run_chunk<NW, MI, 3, MODE>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);   // a range's ragged ends
        else if (cur.nb == 2) run_chunk<NW, MI, 2, MODE>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        else run_chunk<NW, MI, 1, MODE>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        if (!more) break;
        u = unext;
        cur = nxt;
    }
}


}  // namespace

template <int NW, int MI, int MODE>
void launch(const float* U, const float* V, float* M, int T, int Cout, int Cin, int64_t cols, int per_cu, hipStream_t st) {
    constexpr int BM = 32 * MI * NW;
    const int64_t ncb = (cols + 31) / 32, nrt = (Cout + BM - 1) / BM;
    const int64_t units = (int64_t)T * nrt * ncb;
    const size_t shm = 2 * (size_t)(BM * G_LDA + G_BK * G_LDB) * sizeof(float);
    int64_t G = 256 * per_cu;
    if (G > units) G = units;
    hipLaunchKernelGGL((k_lab<NW, MI, MODE>), dim3((unsigned)G), dim3(64 * NW), shm, st, U, V, M, Cout, Cin, cols, (int)ncb,
                       (int)nrt, (int)units);
}

struct Var { const char* name; void (*fn)(const float*, const float*, float*, int, int, int, int64_t, int, hipStream_t); int per_cu; };

int main(int argc, char** argv) {
    const int T = 36, C = 256;
    const int64_t cols = argc > 1 ? atoll(argv[1]) : 6400;
    float *U, *V, *M;
    hipMalloc(&U, sizeof(float) * T * C * C);
    hipMalloc(&V, sizeof(float) * T * C * cols);
    hipMalloc(&M, sizeof(float) * T * C * cols);
    {   // operands with the statistics of the real ones (zeros would flatter the power-limited clock)
        std::vector<float> h((size_t)T * C * cols);
        uint32_t x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        hipMemcpy(V, h.data(), sizeof(float) * T * C * cols, hipMemcpyHostToDevice);
        hipMemcpy(U, h.data(), sizeof(float) * T * C * C, hipMemcpyHostToDevice);
    }
    std::vector<Var> vars = {
        {"<4,2> product", launch<4, 2, 0>, 2},
        {"<4,2> no global loads", launch<4, 2, 1>, 2},
        {"<4,2> no loads, no barriers", launch<4, 2, 3>, 2},
        {"<4,2> no epilogue stores", launch<4, 2, 4>, 2},
        {"<4,2> no loads/barriers/stores", launch<4, 2, 7>, 2},
        {"<4,2> MFMA only (no LDS either)", launch<4, 2, 15>, 2},
        {"<4,2> LDS stage stores, no loads", launch<4, 2, 16>, 2},
        {"<4,2> loads waited for, no LDS stores", launch<4, 2, 32>, 2},
        {"<4,2> loads+waits, no LDS st, no epi", launch<4, 2, 36>, 2},
        {"<8,1> product", launch<8, 1, 0>, 2},
    };
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double fl = 2.0 * T * C * C * (double)cols;
    for (int pass = 0; pass < (argc > 2 ? 1 : 3); ++pass)
        for (auto& v : vars) {
            const int warm = argc > 2 ? 2 : 150, reps = argc > 2 ? atoi(argv[2]) : 300;
            for (int i = 0; i < warm; ++i) v.fn(U, V, M, T, C, C, cols, v.per_cu, st);
            hipEventRecord(e0, st);
            for (int i = 0; i < reps; ++i) v.fn(U, V, M, T, C, C, cols, v.per_cu, st);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("pass %d  %-36s %8.1f us  %6.1f TF  (%.3f of 157.3)\n", pass, v.name, ms / reps * 1e3, fl / (ms / reps * 1e-3) / 1e12,
                   fl / (ms / reps * 1e-3) / 1e12 / 157.3);
            fflush(stdout);
        }
    return 0;
}
