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
// block of 32 columns); a workgroup owns a contiguous unit range and walks it in chunks of up to four column blocks
// (a 64 NW x 128 output tile at the conv stacks' width: NW wavefronts of 64 rows x 128 columns = 2 x 4 MFMA tiles, 128
// accumulator registers per lane).  Work per workgroup differs by at most one 32-column block: the tiled round-2 launch
// (1800 tiles of 256 x 128 on 512 resident slots = 3.52 rounds at 400 chains) lost a quarter of the chip to its last
// round.  G = 2 workgroups per CU (48 KB of LDS each); the workgroups of one XCD (blockIdx % 8) take consecutive ranges,
// so that the workgroups sharing an L2 share their U[t], and an XCD's longer ranges go to its first workgroups.
// K advances 16 at a time through a double-buffered LDS stage filled by LDS-DMA (one barrier per step); the DMA of the
// next step -- of the next CHUNK at the end of a chunk -- is in flight while this step multiplies, so the loop never
// drains between tiles.  LDS tiles are unpadded (the DMA writes lane-linear):
//   A: [row][16], 16-byte granules XOR-swizzled on the source side -- a lane reads 4 consecutive k of its row as one
//                 ds_read_b128, conflict-free over the 16 lanes of a phase;
//   B: [k][128]   as in memory; MFMA tile ni of a full chunk takes the columns 4 n + ni (n = lane % 32), so a lane's four
//                 B operands of one k are ONE ds_read_b128 and its four results of one row one dwordx4 store.
// Inside a wavefront the K step is software-pipelined by hand: the operands of the next k pair are requested before the
// eight MFMAs of this one are issued (run_chunk_p; what the compiler's own placement cost is in its comment).
// The MFMA contraction index is permuted (half-wave g takes k = 8j + 4g + i in step i of chunk j) -- the same
// permutation on both operands, i.e. the same sum in another fixed order (the order of the round-2 kernel: which lane
// computes an output, and which workgroup, never changes its bits; tools/gemm_probe.py checks own == r02 bitwise).
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
    template <int SKIP = 0>                 // (SKIP: lab only -- 1 leaves the A tile out, 2 the B tile)
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
        for (int i = 0; i < ((SKIP & 1) ? 0 : NA); ++i) {
            const int q = tid + i * NT;
            const uint32_t off = ((q >> 2) < rows_left) ? offA[i] : (uint32_t)(((q & 3) ^ ((q >> 4) & 3)) * 16);
            __builtin_amdgcn_global_load_lds(Ub + off, (lds_ptr_t)(As + (wbase + i * NT) * 4),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ((SKIP & 2) ? 0 : NB); ++i) {
            const uint32_t off = (((tid + i * NT) & 31) * 4 < cols_left) ? offB[i] : offB[i] - (uint32_t)((tid + i * NT) & 31) * 16u;
            __builtin_amdgcn_global_load_lds(Vb + off, (lds_ptr_t)(Bs + (wbase + i * NT) * 4),
                                             16, 0, 0);
        }
    }
};

// The K steps run through NS LDS stages (round 4).  NS = 2 is the double buffer of round 3: the DMA of step k + 1 under the
// multiplies of step k, `__syncthreads()` (vmcnt(0) + barrier) between steps.  NS = 3 keeps the DMA of TWO steps in flight:
// with few columns per launch (13 chains: 208 columns, one 32-column unit per workgroup) a K step has 16 MFMAs per wave
// (0.4 us) and the loop ran at one memory latency per step (16 x 1.4 us = 23 us per launch, profiles/r04b); the wait at the
// end of step k is then `vmcnt(LPS)` -- everything but the newest stage's LPS DMA instructions of this wave has landed --
// followed by a bare s_barrier.  Stage (k + NS - 1) % NS was last read in step k - 1, whose barrier every wave has passed
// before the DMA of step k + NS - 1 is issued in step k.  The summation order does not change.
__device__ __forceinline__ constexpr int waitcnt_vm(int v) {        // s_waitcnt simm16 (gfx9): vmcnt = v, lgkmcnt = 0, expcnt free
    return (v & 15) | (7 << 4) | ((v >> 4) << 14);
}
template <int NS, int LPS>
__device__ __forceinline__ void stage_barrier() {
    if constexpr (NS == 2) {
        __syncthreads();
    } else {
        __builtin_amdgcn_s_waitcnt(waitcnt_vm((NS - 2) * LPS));
        __builtin_amdgcn_s_barrier();
    }
}
template <int NS>
__device__ __forceinline__ int stage_next(int buf, int by = 1) {
    if constexpr (NS == 2) return buf ^ (by & 1);
    const int b = buf + by;
    return b >= NS ? b - NS : b;
}

// One chunk of NBLK column blocks: the K loop (the stage of its first step is in LDS buffer `buf` and synchronised), then
// the stores.  The last step prefetches the first stage of the chunk that follows.  A wavefront owns MI x NBLK MFMA tiles
// (32 MI rows x 32 NBLK columns); ALL operand fragments of a K step are requested from LDS before its first MFMA, so the
// wave pays one LDS latency per 32 MI NBLK / 2 MFMAs instead of one per pair (round-3 visit A: the matrix pipe was busy
// 70 % of the time with the reads interleaved).
template <int NW, int MI, int NBLK, int NS = 2>
__device__ __forceinline__ void run_chunk(Stager<NW, MI>& sg, float* lds, float* __restrict__ M, const Chunk& cur,
                                          const Chunk& nxt, bool more, int& buf, int wave, int l32, int g) {
    constexpr int BM = 32 * MI * NW, STAGE = Stager<NW, MI>::STAGE, LPS = Stager<NW, MI>::NA + Stager<NW, MI>::NB;
    f32x16 acc[MI][NBLK];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NBLK; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
    const int nk = sg.Cin / G_BK;
    for (int kt = 0; kt < nk; ++kt) {
        // DMA of the step NS - 1 ahead into the stage whose last reads completed before the previous barrier: in flight under
        // the multiplies below (NS = 3: and under the next step's).  Past the chunk's last step it is the next chunk's
        // first stage(s); past the last chunk of the range a harmless reload (NS = 3 counts DMA instructions at its barrier)
        const int ahead = kt + NS - 1;
        if (ahead < nk) sg.load(cur, ahead * G_BK, lds, stage_next<NS>(buf, NS - 1));
        else if (more || NS > 2) sg.load(more ? nxt : cur, (ahead - nk) * G_BK, lds, stage_next<NS>(buf, NS - 1));
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
        stage_barrier<NS, LPS>();
        buf = stage_next<NS>(buf);
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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// NBLK consecutive floats of one B row (the lane's columns NBLK l32 .. + NBLK - 1): one ds_read_b128 / b64 where it can be
template <int NBLK>
__device__ __forceinline__ void read_b(float (&b)[NBLK], const float* p) {
    if constexpr (NBLK == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        b[0] = v[0], b[1] = v[1], b[2] = v[2], b[3] = v[3];
    } else if constexpr (NBLK == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(p);
        b[0] = v[0], b[1] = v[1];
    } else {
#pragma unroll
        for (int ni = 0; ni < NBLK; ++ni) b[ni] = p[ni];
    }
}

// The same chunk, software-pipelined inside the wave (round 3, visit C).  In the ISA of run_chunk the compiler sinks every
// operand read to just before its first use: a K step has eight `ds_read ; s_waitcnt lgkmcnt(0)` pairs with nothing
// between them -- eight exposed LDS latencies per 64 MFMAs, which only the co-resident workgroup's wave can fill.  Here
// (+5 % at the bench's shape, profiles/r03C_gemm_probe.txt; one workgroup per CU is now as fast as two were):
//  * the MFMA tile ni of a wave takes the columns NBLK n + ni (n = lane % 32) instead of 32 ni + n: a lane's NBLK B operands
//    of one k are adjacent in LDS (ONE ds_read_b128 instead of four ds_read_b32: 12 LDS instructions per K step instead of
//    36), and its NBLK results of one row are adjacent in memory (dwordx4 stores: 32 per wave and chunk instead of 128).
//    Which lane computes an output does not change its sum;
//  * the operands of sub-step s + 1 (one k pair: MI x NBLK MFMAs) are requested before the MFMAs of sub-step s are issued
//    and held in a second register set (sched_barrier keeps the order), across the barrier too: the barrier sits before the
//    LAST sub-step's MFMAs, whose operands are in registers, and the first reads of the next stage follow it immediately.
template <int NW, int MI, int NBLK, int DMA_AT, int LAB = 0, int NS = 2>
__device__ __forceinline__ void run_chunk_p(Stager<NW, MI>& sg, float* lds, float* __restrict__ M, const Chunk& cur,
                                            const Chunk& nxt, bool more, int& buf, int wave, int l32, int g) {
    constexpr int BM = 32 * MI * NW, STAGE = Stager<NW, MI>::STAGE, LPS = Stager<NW, MI>::NA + Stager<NW, MI>::NB;
    f32x16 acc[MI][NBLK];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NBLK; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
    const int nk = sg.Cin / G_BK;
    const int sw = (l32 >> 2) & 3;                           // source-side swizzle of the A granules, see Stager
    const int offA = (wave * 32 * MI + l32) * G_LDA, offB = BM * G_LDA + (g * 4) * G_LDB + NBLK * l32;
    f32x4 a[2][MI];
    float b[2][NBLK];
    const float* As = lds + buf * STAGE + offA;
    const float* Bs = lds + buf * STAGE + offB;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * G_LDA + ((g ^ sw) << 2));
    read_b<NBLK>(b[0], Bs);
    for (int kt = 0; kt < nk; ++kt) {
        const bool last = kt + 1 == nk;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int j = s >> 2, i = s & 3;
            // a use of this sub-step's operands BEFORE the next ones are requested: the compiler's s_waitcnt lands here, where
            // they are the only reads outstanding (requested one sub-step = MI NBLK MFMAs ago), instead of after the new request
            // -- it emits lgkmcnt(0), not lgkmcnt(n), in front of the first consuming MFMA
#pragma unroll
            for (int ni = 0; ni < NBLK; ++ni) asm volatile("" ::"v"(b[s & 1][ni]));
            if (i == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(a[j][mi]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s < 7) {
                if (s == 3) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        a[1][mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * G_LDA + (((2 + g) ^ sw) << 2));
                }
                read_b<NBLK>(b[(s + 1) & 1], Bs + (((s + 1) >> 2) * 8 + ((s + 1) & 3)) * G_LDB);
            } else {
                if (LAB & 3) {               // LAB: timing experiments with WRONG results (tools/gemm_probe.py --lab)
                    if (!(LAB & 2)) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }
                } else
                stage_barrier<NS, LPS>();    // the next stage has landed (vmcnt) and nobody reads this one any more
                buf = stage_next<NS>(buf);
                As = lds + buf * STAGE + offA;
                Bs = lds + buf * STAGE + offB;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * G_LDA + ((g ^ sw) << 2));
                read_b<NBLK>(b[0], Bs);      // (after the chunk's last step: unused, the next chunk reads its own shape)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == DMA_AT) {
                // the DMA of the next step, branch-free so that its address arithmetic can sit between this sub-step's MFMAs
                // (after the very last step of the workgroup's range: a harmless reload of this chunk's first stage)
                const int ahead = kt + NS - 1;
                const bool over = ahead >= nk;
                Chunk c = cur;
                if (over & more) c = nxt;
                if (!(LAB & 8)) sg.template load<(LAB >> 4) & 3>(c, (over ? ahead - nk : ahead) * G_BK, lds, stage_next<NS>(buf, NS - 1));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NBLK; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][mi][i], b[s & 1][ni], acc[mi][ni], 0, 0, 0);
            if (s == DMA_AT) {               // one MFMA, a slice of the address arithmetic, one DMA instruction, ...
#pragma unroll
                for (int r = 0; r < MI * NBLK; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // register v of lane l: row (v/4)*8 + (l/32)*4 + v%4 of the 32x32 tile; tile ni's column l%32 is column NBLK (l%32) + ni
    // of the chunk, so the lane's NBLK tiles hold NBLK adjacent outputs of that row
    float* Mt = M + (int64_t)cur.t * sg.Cout * sg.cols;
    const int64_t col = (int64_t)cur.cb * 32 + NBLK * l32;
    const bool all_rows = cur.co0 + BM <= sg.Cout;
    auto store = [&](float* q, int mi, int v) {
        if constexpr (NBLK == 4) {
            f32x4 o = {acc[mi][0][v], acc[mi][1][v], acc[mi][2][v], acc[mi][3][v]};
            if (LAB & 64) *reinterpret_cast<f32x4*>(q) = o;
            else __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(q));
        } else if constexpr (NBLK == 2) {
            f32x2 o = {acc[mi][0][v], acc[mi][1][v]};
            __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(q));
        } else {
#pragma unroll
            for (int ni = 0; ni < NBLK; ++ni)
                if (NBLK == 1 || col + ni < sg.cols) __builtin_nontemporal_store(acc[mi][ni][v], q + ni);
        }
    };
    // cols % 4 == 0 and col % NBLK == 0: for NBLK = 4 / 2 / 1 the lane's outputs are all inside or all outside (3: per element)
    if (col < sg.cols && (!(LAB & 4) || acc[0][0][0] == 12345.0f)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row0 = cur.co0 + (wave * MI + mi) * 32 + g * 4;
            float* p = Mt + (int64_t)row0 * sg.cols + col;
            if (all_rows) {
#pragma unroll
                for (int v = 0; v < 16; ++v) store(p + (int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols, mi, v);
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (row0 + (v >> 2) * 8 + (v & 3) < sg.Cout) store(p + (int64_t)((v >> 2) * 8 + (v & 3)) * sg.cols, mi, v);
            }
        }
    }
}

// Issue priority: since the bf16x3 route took the ResNet products, this kernel only runs the heads' and the small products -- short
// launches (75 us alone) that start beside older, longer kernels of the other chain group.  Like the transform kernels
// (net_epilogue.hip, BS_XFORM_PRIO) they go one step above the table kernels and the big GEMM: 156.0 against 156.8 ms per step at
// 1000 chains, three repetitions each (profiles/r06y_priority_ab.txt; the table kernels at 1 or 2: +1.7 % / +3 %).  Scheduling only.
#ifndef BS_GEMM32_PRIO
#define BS_GEMM32_PRIO 1
#endif
template <int NW, int MI, int PIPE, int NS>
__global__ __launch_bounds__(64 * NW, 2) void k_wino_gemm(const float* __restrict__ U, const float* __restrict__ V,
                                                          float* __restrict__ M, int Cout, int Cin, int64_t cols,
                                                          int ncb, int nrt, int units, int even_ranges) {
    if (BS_GEMM32_PRIO) __builtin_amdgcn_s_setprio(BS_GEMM32_PRIO);
    constexpr int BM = 32 * MI * NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [NS][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, g = lane >> 5;

    // Which units: the workgroups of one XCD (blockIdx % 8, round-robin dispatch) take consecutive ranges of one eighth of the
    // list, so that they share their U[t] in that XCD's L2.  The ranges of an XCD differ by one unit; the longer ones go to its
    // FIRST workgroups (blockIdx / 8 = 0, 1, ...): those start first, and consecutive workgroups land on different CUs, so no
    // CU gets two long ranges (evenly spaced long ranges -- every 16th workgroup at the bench's shape -- met on one CU:
    // 30 units there against a mean of 28.1).
    const int G = gridDim.x;
    int u, uend;
    if ((G & 7) == 0 && even_ranges) {          // (probe only: the visit-A/B assignment)
        const int w = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        u = (int)((int64_t)w * units / G);
        uend = (int)((int64_t)(w + 1) * units / G);
    } else if ((G & 7) == 0) {
        const int S = G >> 3, x = blockIdx.x & 7, sl = blockIdx.x >> 3;
        const int ux0 = (int)((int64_t)x * units / 8), nx = (int)((int64_t)(x + 1) * units / 8) - ux0;
        const int base = nx / S, rem = nx - base * S;
        u = ux0 + sl * base + min(sl, rem);
        uend = u + base + (sl < rem ? 1 : 0);
    } else {
        u = (int)((int64_t)blockIdx.x * units / G);
        uend = (int)((int64_t)(blockIdx.x + 1) * units / G);
    }
    u = __builtin_amdgcn_readfirstlane(u);
    uend = __builtin_amdgcn_readfirstlane(uend);
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
    if constexpr (NS > 2) sg.load(cur, G_BK, lds, 1);      // (host: Cin >= 2 G_BK) two steps in flight from the start
    stage_barrier<NS, Stager<NW, MI>::NA + Stager<NW, MI>::NB>();
    int buf = 0;
    while (true) {
        const int unext = u + cur.nb;
        const bool more = unext < uend;
        Chunk nxt = cur;
        if (more) nxt = decode(unext);
        if constexpr (PIPE != 0) {
            constexpr int AT = 1;                 // the sub-step whose MFMAs the DMA issue of the next stage is spread over
            constexpr int LAB = PIPE >= 16 ? PIPE - 16 : 0;
            // full chunks pipelined; a range's ragged ends (1-3 column blocks: too few MFMAs per sub-step to cover a read) as before
            if (cur.nb == 4) run_chunk_p<NW, MI, 4, AT, LAB, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
            else if (cur.nb == 3) run_chunk<NW, MI, 3, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
            else if (cur.nb == 2) run_chunk<NW, MI, 2, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
            else run_chunk<NW, MI, 1, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        } else {
            if (cur.nb == 4) run_chunk<NW, MI, 4, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
            else if (cur.nb == 3) run_chunk<NW, MI, 3, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
            else if (cur.nb == 2) run_chunk<NW, MI, 2, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
            else run_chunk<NW, MI, 1, NS>(sg, lds, M, cur, nxt, more, buf, wave, l32, g);
        }
        if (!more) break;
        u = unext;
        cur = nxt;
    }
    // the range's last K steps issue a redundant DMA (branch-free pipelining): it must have landed before this workgroup's LDS
    // is handed to the next one on the CU (round 4: bs_wino_gemm_bf16x3 decoded garbage beside other kernels until it waited)
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0)
}

int cu_count() {
    // BITSWAP_GEMM_CUS: the compute units the launch stream may use when it is a CU-masked one (bitswap_amd.hip.MaskedStream:
    // the persistent grid is sized by the CUs it can reach, not by the chip)
    if (const char* e = getenv("BITSWAP_GEMM_CUS")) {
        const int v = atoi(e);
        if (v > 0) return v;
    }
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
    // LDS stages: 3 for launches with little work per workgroup (latency-bound K loop, see stage_barrier), else 2.  Small =
    // at most BITSWAP_GEMM_NS3_UNITS units (default: four per resident workgroup slot of the two-stage shape); results are
    // bitwise the same either way
    const int per_cu2 = BM >= 128 ? 2 : BM == 64 ? 4 : 6;
    const long ns3_units = [] { const char* e = getenv("BITSWAP_GEMM_NS3_UNITS"); return e ? atol(e) : -1L; }();
    const int64_t small = ns3_units >= 0 ? ns3_units : (int64_t)cu_count() * per_cu2 * 4;
    const bool ns3 = Cin >= 2 * G_BK && units <= small;
    const size_t shm = (ns3 ? 3 : 2) * (size_t)(BM * G_LDA + G_BK * G_LDB) * sizeof(float);
    // workgroups per CU: two of the 256-row shape (48 / 72 KB of LDS each, one wavefront per SIMD each, up to 256 registers per
    // lane); the one-wave shape of the head convolutions is bounded by its 20 / 30 KB of LDS
    const int per_cu = ns3 ? (BM >= 128 ? 2 : BM == 64 ? 4 : 5) : per_cu2;
    int64_t G = (int64_t)cu_count() * per_cu;
    const int wgs_env = [] { const char* e = getenv("BITSWAP_GEMM_WGS_PER_CU"); return e ? atoi(e) : 0; }();   // tuning only
    if (wgs_env > 0) G = (int64_t)cu_count() * wgs_env;
    // few units per workgroup slot: fewer workgroups with whole four-block chunks (an A tile fetched per 128 columns instead of
    // per 32 or 64, the hand-pipelined chunk code) beat one or two units on every slot -- BITSWAP_GEMM_MIN_UNITS per workgroup
    const int min_units = [] { const char* e = getenv("BITSWAP_GEMM_MIN_UNITS"); return e ? atoi(e) : 1; }();
    if (min_units > 1 && G * min_units > units) G = (units + min_units - 1) / min_units;
    if (G > units) G = units;
    if (G < 1) G = 1;
    const int variant = [] { const char* e = getenv("BITSWAP_GEMM_VARIANT"); return e ? atoi(e) : 2; }();     // tuning only
    const int even = getenv("BITSWAP_GEMM_EVEN_RANGES") ? 1 : 0;
    auto go = [&](auto kern) {
        if (shm > 48 * 1024) {                 // more than the default dynamic LDS limit: raise it once per kernel and device
            static bool raised[16] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev >= 0 && dev < 16 && !raised[dev]) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
                    return false;
                raised[dev] = true;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(64 * NW), shm, st, U, V, M, Cout, Cin, cols, (int)ncb, (int)nrt, (int)units, even);
        return true;
    };
    bool ok = true;
    if (variant == 0) ok = go(k_wino_gemm<NW, MI, 0, 2>);     // the visit-A/B kernel (operand reads where the compiler puts them), for the probe
#ifdef BS_GEMM_LAB    // timing experiments with WRONG results (tools/gemm_probe.py --lab builds its own library with this)
    else if (NW == 4 && MI == 2 && variant == 17) ok = go(k_wino_gemm<4, 2, 17, 2>);      // no vmcnt wait at the barrier
    else if (NW == 4 && MI == 2 && variant == 19) ok = go(k_wino_gemm<4, 2, 19, 2>);      // no barrier at all
    else if (NW == 4 && MI == 2 && variant == 20) ok = go(k_wino_gemm<4, 2, 20, 2>);      // no stores
    else if (NW == 4 && MI == 2 && variant == 24) ok = go(k_wino_gemm<4, 2, 24, 2>);      // no DMA
    else if (NW == 4 && MI == 2 && variant == 31) ok = go(k_wino_gemm<4, 2, 31, 2>);      // MFMAs and LDS reads only
    else if (NW == 4 && MI == 2 && variant == 32) ok = go(k_wino_gemm<4, 2, 32, 2>);      // no DMA of the A tile (U)
    else if (NW == 4 && MI == 2 && variant == 48) ok = go(k_wino_gemm<4, 2, 48, 2>);      // no DMA of the B tile (V)
    else if (NW == 4 && MI == 2 && variant == 80) ok = go(k_wino_gemm<4, 2, 80, 2>);      // (right results) plain stores instead of nt
#endif
    else if (ns3) ok = go(k_wino_gemm<NW, MI, 2, 3>);
    else ok = go(k_wino_gemm<NW, MI, 2, 2>);
    return ok && hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
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
