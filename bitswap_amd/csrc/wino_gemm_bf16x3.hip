// wino_gemm_bf16x3.hip -- the batched product of a Winograd-domain convolution, M[t] = U[t] x V[t], with every float32 operand
// split EXACTLY into three bfloat16 limbs (x = x0 + x1 + x2, 3 x 8 significand bits = the 24 of float32) and the product
// assembled from limb products on the bf16 matrix cores with float32 accumulation (v_mfma_f32_32x32x16_bf16, 16x the rate of
// v_mfma_f32_32x32x2_f32):
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) [+ (a1 b2 + a2 b1) + a2 b2]        NPROD = 6 [9]
// Opt-in (BITSWAP_GEMM_ARITH=bf16x3 / bf16x3x9, bitswap_amd/model.py), its own conv route in the stream fingerprint: the
// float32 bits of (mu, scale) differ from the fp32-MFMA route's, so sender and receiver must both take it.  What it keeps:
// one fixed summation order per output element that depends on Cin alone (k blocks of 16 in order, the limb products of a
// block in the order above: small terms first), no split-K, no shape-dependent kernel choice -- results are bitwise
// independent of `cols` and of the launch shape, like bs_wino_gemm_f32's.  What it changes: with NPROD = 6 the three
// limb products of relative size 2^-24 are dropped (the same size as the rounding of a float32 product); NPROD = 9 keeps all
// nine, so every product is exact and the only roundings left are the float32 accumulations.
// VERDICT r3 #5 asked for the evidence before any promotion: tests/test_codec_gpu.py::test_bf16x3_gemm_* (error against a
// float64 product next to the fp32 route's), profiles/r04*_bf16x3_*.
//
// U arrives pre-split AND pre-tiled (weights: once, at Model.fuse(); bitswap_amd.hip.frags_bf16x3): Uf [T][ceil(Cout/32)]
// [Cin/16][3 limbs][64 lanes][8] bf16, the MFMA A fragments themselves.  V [T][Cin][cols] float32 is split in registers after
// the LDS read (truncation split: x0 = x & 0xffff0000, r = x - x0, ...: two VALU operations per limb, exact).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../../include/bitswap_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Workgroup = 256 rows x 256 columns of one t: 4 wavefronts of 64 rows x 256 columns = 2 x 8 MFMA tiles (256 accumulator
// registers per lane: ONE workgroup per CU, one wavefront per SIMD, up to 512 registers each).  Why this big: at 16x the MFMA
// rate the operands are the bottleneck -- the first version of this kernel (256 x 128 tiles, A limbs through LDS-DMA like
// bs_wino_gemm_f32's A tile) moved 1.17 GB per launch through the LDS-DMA path and ran at its ~6 TB/s, 1.2x the fp32 kernel
// (profiles/r04g).  Here the A operand never touches LDS: the limbs of U arrive PRE-TILED as MFMA fragments
// (Uf [T][Cout/32][Cin/16][3][64 lanes][8 bf16]: what lane l of a wavefront holds for a 32-row tile and a 16-deep k block is
// 16 contiguous bytes, a wavefront's fragment 1 KB), so a wavefront fetches the six fragments of its K step with six
// coalesced global loads, one step ahead, straight into registers -- no other wavefront needs them.  Only V goes through
// LDS (every wavefront multiplies all 256 columns): a [16 k][256] float32 stage of 16 KB, double buffered, filled by LDS-DMA.
constexpr int X_BM = 256, X_BN = 256, X_BK = 16, X_NT = 256;
constexpr int X_STAGE = X_BK * X_BN * 4;                  // 16 KB; three stages per workgroup

union Pack8 {
    u32x4 u;
    bf16x8 b;
};

// limb products of one k block in the order they are added (i = limb of a, j = limb of b): small terms first
__device__ constexpr int X_ORDER9[9][2] = {{2, 2}, {1, 2}, {2, 1}, {0, 2}, {1, 1}, {2, 0}, {0, 1}, {1, 0}, {0, 0}};

// CLAIM / STRICT: diagnostics of the co-residency failure (BITSWAP_BF16X3_DIAG, tests/test_codec_gpu.py::test_bf16x3_gemm_beside_a_small_kernel):
// CLAIM = false leaves the register share as the compiler sized it (a small wavefront of another kernel then fits beside this
// one on the SIMD); STRICT = true replaces every counted wait by vmcnt(0).
// STRAY = true RE-INTRODUCES the round-4 defect for the record: every step issues its LDS-DMA (the last three re-fetch the final
// stage) and the workgroup exits without waiting for them -- a write still in flight lands in the LDS of whatever workgroup the
// CU runs next, which without the register claim may belong to ANOTHER kernel.
template <int NPROD, int LAB = 0, bool CLAIM = true, bool STRICT = false, bool STRAY = false>     // LAB != 0: timing experiments with WRONG results (-DBS_GEMM_LAB builds only)
__global__ __launch_bounds__(X_NT, 1) void k_wino_gemm_bf16x3(const uint16_t* __restrict__ Uf, const float* __restrict__ V,
                                                              float* __restrict__ M, int T, int Cout, int Cin, int64_t cols,
                                                              int ncc, int nrt) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [3][X_STAGE]
    // The wavefront claims the WHOLE register file of its SIMD (256 architectural + 256 accumulation registers; it needs 448).
    // History: round 4 saw the forked and the two-group codecs decode garbage with this arithmetic, blamed a small wavefront of
    // another kernel beside this one, and added the claim so that nothing fits.  Round 5 (tools/bf16x3_repro.py):
    // the claim changes the allocation and nothing else (identical instruction streams); the products are bit-stable with or
    // without it -- 600,000 launches beside the kernels that fit next to it, and every launch doubled inside failing codec runs --
    // and the FORKED block step with the two-workgroup shape fails with the claim too (2.3 % of runs against 5-7 % without).
    // The cause was not in this kernel at all: a packed float32 addition in its NEIGHBOUR k_wino_fused<6,6>; the fix
    // is net_epilogue.hip's -fno-slp-vectorize build flag (bitswap_amd/build.py::FILE_FLAGS).  The claim is a template switch
    // (CLAIM) kept for the A/B of round 6: claimed, nothing else fits on the SIMD; unclaimed, a coder wavefront does.
    if constexpr (CLAIM) asm volatile("" ::: "v255");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, g = lane >> 5;
    // chunk of this workgroup: (t, row tile, 256-column chunk).  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8),
    // each XCD with an L2 of its own: XCD x takes the x-th contiguous eighth of the chunk list, so the workgroups that share
    // an L2 work through the same few t one after the other and the limbs of U[t] (393 KB at 256 x 256) are fetched from HBM
    // once per XCD instead of once per workgroup (visit H: with consecutive chunks on consecutive XCDs every XCD touched all
    // 36 U[t], 14 MB against 4 MB of L2, and the launch moved 1 GB).
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int t = wg / (nrt * ncc), rem = wg - t * (nrt * ncc), rt = rem / ncc, cc = rem - rt * ncc;
    const int co0 = rt * X_BM;
    const int64_t c0 = (int64_t)cc * X_BN;
    const int cols_left = (int)min((int64_t)X_BN, cols - c0);
    const int nk = Cin / X_BK, nrt32 = (Cout + 31) / 32;

    // ---- A: this lane's 16 bytes of fragment (row tile 32, k block, limb): ((((t nrt32 + r32) nk + kb) 3 + limb) 64 + lane) 8
    //      row tiles beyond Cout re-read the last one (their outputs are never stored)
    const uint16_t* a_base[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r32 = min(co0 / 32 + wave * 2 + mi, nrt32 - 1);
        a_base[mi] = Uf + (((int64_t)t * nrt32 + r32) * nk * 3 * 64 + lane) * 8;
    }
    auto load_a = [&](bf16x8 (&dst)[2][3], int kb) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                p.u = *reinterpret_cast<const u32x4*>(a_base[mi] + ((int64_t)kb * 3 + i) * 64 * 8);
                dst[mi][i] = p.b;
            }
    };
    // ---- B: LDS-DMA granule q = tid + j * 256 (j = 0 .. 3): k = q >> 6, columns 4 (q & 63) ..; beyond the chunk: column 0
    const float* b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = tid + j * X_NT, k = q >> 6, cg = (q & 63) * 4;
        b_src[j] = V + ((int64_t)t * Cin + k) * cols + c0 + (cg < cols_left ? cg : 0);
    }
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto load_stage = [&](int k0, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds(b_src[j] + (int64_t)k0 * cols, (lds_ptr_t)(lds + buf * X_STAGE + (wbase + j * X_NT) * 16), 16, 0, 0);
    };

    f32x16 acc[2][8];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;

    // ---- the K loop.  A v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles and hides about five other
    // instructions (MI355X_MICROARCH.md); the split costs 5.5 VALU instructions per float, 56 per column tile of a lane.  The
    // eight column tiles of a wavefront go in two groups of four (group h = columns 128 h + 4 l32 + ni: one ds_read_b128 per k):
    //   phase A of step k: products of group 0 (48 or 72 MFMAs)   | split of group 1 of step k
    //   barrier          : stage k + 1 has landed, nobody reads stage k any more -> DMA of stage k + 2 into its buffer
    //   phase B of step k: products of group 1                    | LDS reads of stage k + 1, split of its group 0
    // Limb-product index outermost inside a phase: consecutive MFMAs hit eight different accumulators, and an output still
    // sees its products in ONE fixed order (k blocks ascending, X_ORDER9 inside a block).
    bf16x8 a[2][3], an[2][3], b[2][4][3];
    f32x4 raw[2][8];                       // raw float32 of this lane's 8 k: group 0 / group 1
    auto read_b = [&](const char* st, int h) {       // 8 k (8 g ..) of columns 128 h + 4 l32 .. + 3
        const float* Bs = reinterpret_cast<const float*>(st) + (g * 8) * X_BN + 128 * h + 4 * l32;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) raw[h][kk] = *reinterpret_cast<const f32x4*>(Bs + kk * X_BN);
    };
    auto split_group = [&](int h) {                  // truncation split into three bf16 limbs: exact
        // Stage by stage over the 8 k of a tile, not float by float: the four operations of one float form a dependent chain
        // (and, subtract, and, subtract), and a lone wavefront pays the full VALU latency for every dependent instruction --
        // written float by float the split did not hide under the MFMAs at all (visit H: 5700 cycles per K step with NO memory
        // instruction in the loop, 3072 of them MFMA).  Eight independent chains side by side issue back to back.
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            uint32_t u0[8], u1[8];
            float r1[8], r2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u0[e] = __float_as_uint(raw[h][e][ni]) & 0xffff0000u;
            if (LAB & 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) u1[e] = u0[e], r2[e] = raw[h][e][ni];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) r1[e] = raw[h][e][ni] - __uint_as_float(u0[e]);
#pragma unroll
                for (int e = 0; e < 8; ++e) u1[e] = __float_as_uint(r1[e]) & 0xffff0000u;
#pragma unroll
                for (int e = 0; e < 8; ++e) r2[e] = r1[e] - __uint_as_float(u1[e]);
            }
            Pack8 p0, p1, p2;
            // bf16 pair: low half = even k, high half = odd k (the top 16 bits of each float32 word)
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p0.u[kp] = __builtin_amdgcn_perm(u0[2 * kp + 1], u0[2 * kp], 0x07060302u);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p1.u[kp] = __builtin_amdgcn_perm(u1[2 * kp + 1], u1[2 * kp], 0x07060302u);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p2.u[kp] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * kp + 1]), __float_as_uint(r2[2 * kp]), 0x07060302u);
            b[h][ni][0] = p0.b, b[h][ni][1] = p1.b, b[h][ni][2] = p2.b;
        }
    };
    auto products = [&](int h) {
#pragma unroll
        for (int p = ((LAB & 2) ? 8 : 9 - NPROD); p < 9; ++p)      // (LAB & 2: one product per tile instead of NPROD)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][4 * h + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][X_ORDER9[p][0]], b[h][ni][X_ORDER9[p][1]],
                                                                                  acc[mi][4 * h + ni], 0, 0, 0);
    };
    auto interleave = [&]() {              // one MFMA, then up to five of the other instructions, ...
#pragma unroll
        for (int r = 0; r < 8 * NPROD; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
    };

    // ---- operands run ahead of the multiplies (visit H: with one step of prefetch and vmcnt(0) barriers a step took 7000
    // cycles -- 3072 of MFMA plus a memory wait that nothing covered: one wavefront per SIMD has no partner to hide behind, and
    // the launch moves 1 GB through L2 and HBM, ~4000 cycles per step and CU at the rate the chip sustains).
    //   V: three LDS stages, the DMA of stage k + 3 issued at the barrier of step k (two steps of flight);
    //   U: the fragments of step k + 1 requested at the top of step k by INLINE-ASM loads, so that the wait is ours to place
    //      (hipcc's own s_waitcnt for a register load in a loop is vmcnt(0), which would also wait for the DMA just issued).
    // Counted waits, per wavefront in issue order  top(s): 6 fragment loads of step s + 1 | mid(s): 4 DMA of stage s + 3:
    //   barrier of step k needs stage k + 1: all but the newest 10 (4 DMA of stage k + 2, 6 loads of step k + 1) -> vmcnt(10)
    //   end of step k needs the fragments of step k + 1: all but the newest 4 (DMA of stage k + 3)              -> vmcnt(4)
    // The last three steps issue no DMA and use vmcnt(0).
    auto load_a_async = [&](bf16x8 (&dst)[2][3], int kb) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                const uint16_t* src = a_base[mi] + ((int64_t)kb * 3 + i) * 64 * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(p.u) : "v"(src) : "memory");
                dst[mi][i] = p.b;
            }
    };
    auto landed = [&](bf16x8 (&x)[2][3]) {                                 // ties later uses to this point (after our s_waitcnt)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                p.b = x[mi][i];
                asm volatile("" : "+v"(p.u));
                x[mi][i] = p.b;
            }
    };
    auto stage_of = [&](int st_idx) -> const char* { return lds + (st_idx % 3) * X_STAGE; };
    load_stage(0, 0);
    load_a_async(a, 0);
    if (nk > 1) load_stage(X_BK, 1);
    if (nk > 2) load_stage(2 * X_BK, 2);
    __syncthreads();                                                      // (vmcnt(0): everything above has landed)
    landed(a);
    read_b(lds, 0);
    read_b(lds, 1);
    split_group(0);
    // One K step; DMA = false for the last three (nothing left to fetch).  No branch inside: a branch would end the scheduling
    // region the interleaving lives in, so past the last stage the LDS reads and the split simply work on a stale stage
    // whose limbs are never multiplied.  A workgroup never leaves with an LDS-DMA in flight (the write would land in the LDS
    // of whatever workgroup the CU runs next).
    auto step = [&](int kt, auto dma) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(LAB & 4)) load_a_async(an, min(kt + 1, nk - 1));            // fragments of step kt + 1
        products(0);
        split_group(1);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(dma)::value) {
            if constexpr (STRICT) __builtin_amdgcn_s_waitcnt(0x0070);
            else __builtin_amdgcn_s_waitcnt(0x007A);                      // vmcnt(10), lgkmcnt(0): stage kt + 1 has landed
            __builtin_amdgcn_s_barrier();                                 // ... in every wavefront; stage kt is in registers everywhere
            if (!(LAB & 4)) load_stage((STRAY ? min(kt + 3, nk - 1) : kt + 3) * X_BK, kt % 3);   // into the buffer stage kt lived in
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
        const char* nx = stage_of(kt + 1);
        read_b(nx, 0);
        products(1);
        split_group(0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        read_b(nx, 1);
        if constexpr (decltype(dma)::value && !STRICT) __builtin_amdgcn_s_waitcnt(0x0F74);   // vmcnt(4): the fragments of step kt + 1 are here
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
        if (!(LAB & 4)) {
            landed(an);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int i = 0; i < 3; ++i) a[mi][i] = an[mi][i];
        }
    };
    int kt = 0;
    for (; kt < nk - 3; ++kt) step(kt, std::true_type{});
    for (; kt < nk; ++kt) {
        if constexpr (STRAY) step(kt, std::true_type{});
        else step(kt, std::false_type{});
    }
    // register v of lane l: row (v/4)*8 + (l/32)*4 + v%4 of the 32x32 tile, column l%32 -> chunk column 128 h + 4 (l%32) + ni
    float* Mt = M + (int64_t)t * Cout * cols;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cl = 128 * h + 4 * l32;
        if (cl < cols_left) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row0 = co0 + wave * 64 + mi * 32 + g * 4;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = row0 + (v >> 2) * 8 + (v & 3);
                    if (row < Cout) {
                        f32x4 o = {acc[mi][4 * h + 0][v], acc[mi][4 * h + 1][v], acc[mi][4 * h + 2][v], acc[mi][4 * h + 3][v]};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Mt + (int64_t)row * cols + c0 + cl));
                    }
                }
            }
        }
    }
    if constexpr (!STRAY) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): nothing of this workgroup is in flight when its LDS is handed on
}


// ---- the two-wavefronts-per-SIMD shape: 256 rows x 128 columns per workgroup (4 wavefronts of 64 x 128 = 2 x 4 MFMA tiles, 128
// accumulator registers in the architectural file, <= 256 registers per lane), TWO workgroups per CU.  Why: inside ONE
// wavefront nothing hides under an MFMA on this chip the way the split needs it -- visit H, no memory instruction in the K
// loop: 96 MFMAs + 450 other instructions per step took the SUM of their issue times (48 cycles per MFMA group), interleaved
// or not -- so the overlap has to come from a partner wavefront on the same SIMD: each wavefront alternates a VALU burst (the
// split of two column tiles) with a burst of 4 NPROD back-to-back MFMAs, and while one splits the other multiplies.
constexpr int Y_BN = 128, Y_STAGE = X_BK * Y_BN * 4;      // 8 KB per stage, three stages per workgroup

// LATE (diagnostics): the fragment loads of step k + 1, which land IN PLACE in the registers the multiplies of step k read, are
// issued after the step's barrier instead of right behind the last multiply.
// PLAIN_STORE (diagnostics): M leaves through ordinary stores instead of nontemporal ones.
// COHERENT (diagnostics): the LDS-DMA loads of V carry sc0 sc1 (system scope: served past a possibly stale L2 line).
template <int NPROD, bool CLAIM = true, bool STRICT = false, bool LATE = false, bool PLAIN_STORE = false, bool COHERENT = false>
__global__ __launch_bounds__(X_NT, 2) void k_wino_gemm_bf16x3_o2(const uint16_t* __restrict__ Uf, const float* __restrict__ V,
                                                                 float* __restrict__ M, int T, int Cout, int Cin, int64_t cols,
                                                                 int ncc, int nrt) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [3][Y_STAGE]
    if constexpr (CLAIM) asm volatile("" ::: "v255");            // the whole 256-register share of the SIMD (see k_wino_gemm_bf16x3)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, g = lane >> 5;
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);  // an XCD works through a contiguous eighth of the chunk list
    const int t = wg / (nrt * ncc), rem = wg - t * (nrt * ncc), rt = rem / ncc, cc = rem - rt * ncc;
    const int co0 = rt * X_BM;
    const int64_t c0 = (int64_t)cc * Y_BN;
    const int cols_left = (int)min((int64_t)Y_BN, cols - c0);
    const int nk = Cin / X_BK, nrt32 = (Cout + 31) / 32;

    const uint16_t* a_base[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r32 = min(co0 / 32 + wave * 2 + mi, nrt32 - 1);
        a_base[mi] = Uf + (((int64_t)t * nrt32 + r32) * nk * 3 * 64 + lane) * 8;
    }
    bf16x8 a[2][3], b[2][3];
    auto load_a_async = [&](int kb) {          // straight into `a`: every MFMA that reads the old fragments has been issued
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                const uint16_t* src = a_base[mi] + ((int64_t)kb * 3 + i) * 64 * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(p.u) : "v"(src) : "memory");
                a[mi][i] = p.b;
            }
    };
    auto landed_a = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                p.b = a[mi][i];
                asm volatile("" : "+v"(p.u));
                a[mi][i] = p.b;
            }
    };
    // B: LDS-DMA granule q = tid + j * 256 (j = 0, 1): k = q >> 5, columns 4 (q & 31) ..; beyond the chunk: column 0
    const float* b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + j * X_NT, k = q >> 5, cg = (q & 31) * 4;
        b_src[j] = V + ((int64_t)t * Cin + k) * cols + c0 + (cg < cols_left ? cg : 0);
    }
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto load_stage = [&](int k0, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(b_src[j] + (int64_t)k0 * cols, (lds_ptr_t)(lds + buf * Y_STAGE + (wbase + j * X_NT) * 16), 16, 0,
                                             COHERENT ? 17 : 0);      // 17 = sc0 | sc1
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;

    f32x4 raw[8];                                  // this lane's 8 k (8 g ..) of columns 4 l32 .. + 3 (tile ni = column 4 l32 + ni)
    auto read_b = [&](const char* st) {
        const float* Bs = reinterpret_cast<const float*>(st) + (g * 8) * Y_BN + 4 * l32;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) raw[kk] = *reinterpret_cast<const f32x4*>(Bs + kk * Y_BN);
    };
    auto split_pair = [&](int h) {                 // tiles 2 h, 2 h + 1 -> b[0], b[1]: truncation split, exact
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            uint32_t u0[8], u1[8];
            float r1[8], r2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u0[e] = __float_as_uint(raw[e][2 * h + ni]) & 0xffff0000u;
#pragma unroll
            for (int e = 0; e < 8; ++e) r1[e] = raw[e][2 * h + ni] - __uint_as_float(u0[e]);
#pragma unroll
            for (int e = 0; e < 8; ++e) u1[e] = __float_as_uint(r1[e]) & 0xffff0000u;
#pragma unroll
            for (int e = 0; e < 8; ++e) r2[e] = r1[e] - __uint_as_float(u1[e]);
            Pack8 p0, p1, p2;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p0.u[kp] = __builtin_amdgcn_perm(u0[2 * kp + 1], u0[2 * kp], 0x07060302u);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p1.u[kp] = __builtin_amdgcn_perm(u1[2 * kp + 1], u1[2 * kp], 0x07060302u);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p2.u[kp] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * kp + 1]), __float_as_uint(r2[2 * kp]), 0x07060302u);
            b[ni][0] = p0.b, b[ni][1] = p1.b, b[ni][2] = p2.b;
        }
    };
    auto products = [&](int h) {                   // back to back: limb-product index outermost, four accumulators in turn
#pragma unroll
        for (int p = 9 - NPROD; p < 9; ++p)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][2 * h + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][X_ORDER9[p][0]], b[ni][X_ORDER9[p][1]],
                                                                                  acc[mi][2 * h + ni], 0, 0, 0);
    };
    // per wavefront, in issue order:  end of step s: 6 fragment loads of step s + 1, then (after the barrier) 2 DMA of stage s + 3
    //   top of step k needs the fragments of step k: all but the newest 2 (DMA of stage k + 2)                 -> vmcnt(2)
    //   barrier of step k needs stage k + 1: all but the newest 8 (2 DMA of stage k + 2, 6 loads of step k + 1) -> vmcnt(8)
    load_stage(0, 0);
    if (nk > 1) load_stage(X_BK, 1);
    if (nk > 2) load_stage(2 * X_BK, 2);
    load_a_async(0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    read_b(lds);
    auto step = [&](int kt, auto dma) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(dma)::value && !STRICT && !LATE) __builtin_amdgcn_s_waitcnt(0x0F72);   // vmcnt(2): this step's fragments are here
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        landed_a();
        split_pair(0);
        __builtin_amdgcn_sched_barrier(0);
        products(0);
        __builtin_amdgcn_sched_barrier(0);
        split_pair(1);
        __builtin_amdgcn_sched_barrier(0);
        products(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LATE) load_a_async(min(kt + 1, nk - 1));
        if constexpr (decltype(dma)::value) {
            if constexpr (STRICT) __builtin_amdgcn_s_waitcnt(0x0070);
            else if constexpr (LATE) __builtin_amdgcn_s_waitcnt(0x0072);   // (no fragment loads in flight: vmcnt(2) = all but the DMA of stage kt + 2)
            else __builtin_amdgcn_s_waitcnt(0x0078);                      // vmcnt(8), lgkmcnt(0): stage kt + 1 has landed
            __builtin_amdgcn_s_barrier();
            load_stage((kt + 3) * X_BK, kt % 3);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (LATE) {
            __builtin_amdgcn_sched_barrier(0);
            load_a_async(min(kt + 1, nk - 1));
            // (the last steps have no counted wait ahead of them: an asynchronous load must have landed before the registers
            // it writes are handed to the epilogue)
            if constexpr (!decltype(dma)::value) __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_b(lds + ((kt + 1) % 3) * Y_STAGE);
    };
    int kt = 0;
    for (; kt < nk - 3; ++kt) step(kt, std::true_type{});
    for (; kt < nk; ++kt) step(kt, std::false_type{});
    float* Mt = M + (int64_t)t * Cout * cols;
    const int cl = 4 * l32;
    if (cl < cols_left) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int row0 = co0 + wave * 64 + mi * 32 + g * 4;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = row0 + (v >> 2) * 8 + (v & 3);
                if (row < Cout) {
                    f32x4 o = {acc[mi][0][v], acc[mi][1][v], acc[mi][2][v], acc[mi][3][v]};
                    if constexpr (PLAIN_STORE) *reinterpret_cast<f32x4*>(Mt + (int64_t)row * cols + c0 + cl) = o;
                    else __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Mt + (int64_t)row * cols + c0 + cl));
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0): nothing of this workgroup is in flight when its LDS is handed on
}


// ---- the wave-specialised shape (round 6): ONE multiplying wavefront per SIMD, the split on wavefronts of its own.
// What round 6 measured first (tools/probes/mfma_rate.hip, profiles/r06b_mfma_rate.txt): a loop of NOTHING but
// v_mfma_f32_32x32x16_bf16 on limb operands sustains 1,782 TFLOP/s with one wavefront per SIMD (the chip clocks down to ~1.7 GHz
// under that load) and only 1,330 with two (on zeros: 2,446 against 1,651 -- the loss is the interleaving of two wavefronts' MFMAs on
// one matrix pipe, not power).  So the two shapes above were capped by their own occupancy: the o2 shape puts two multiplying
// wavefronts on every SIMD (ceiling 226.5 GFLOP / 1,330 T = 170 us for the T36 x 256^2 x 8000 launch, measured 219), the one-wave
// shape splits V in the multiplying wavefront itself, where nothing hides under an MFMA (6 VALU per MFMA).
// Here a workgroup is 8 wavefronts, two per SIMD, with different jobs:
//   wavefronts 0-3 (consumers): 64 rows x 128 columns each = 2 x 4 MFMA tiles, 128 accumulator registers.  Their loop holds MFMAs,
//     the LDS reads of the next column tile's three limb fragments (one tile = 12 MFMAs = 384 cycles ahead: the LDS latency is
//     always covered) and the six fragment loads of U for the next k block -- no VALU arithmetic at all;
//   wavefronts 4-7 (producers): wavefront w splits the V stage of the k blocks s = w (mod 4): eight 16-byte loads per lane a whole
//     ring ahead, the truncation split of the o2 shape (exact), twelve 16-byte LDS stores that leave the stage as MFMA B
//     fragments.  One wavefront splits for the whole workgroup (the o2 shape split every stage four times), and its VALU burst
//     runs beside the consumers' MFMAs on the other pipe.
// A ring of four LDS stages (12 KB each), one s_barrier per k block for all eight wavefronts:
//   the stage of k block s is written in interval s - 3 (after barrier s - 4: its slot was last read for block s - 4) and first
//   read at the end of interval s - 1.
// Same arithmetic as the shapes above, bit for bit: k blocks ascending, X_ORDER9 inside a block, float32 accumulation -- the
// results do not depend on the shape (asserted by tests/test_codec_gpu.py::test_bf16x3_gemm_shapes_agree_bitwise).
// Registers: consumers ~200, the kernel is compiled for two wavefronts per SIMD with at most 224 each, so that a 64-register
// coder wavefront of the other chain group still fits on the SIMD (the o2 shape's 2 x 256 left no room: its GEMMs pushed the
// serial pops out of their hiding place).
constexpr int W_BN = 128, W_STAGE = 4 * 3 * 1024, W_NS = 4;
// The producers' half of both wave-specialised kernels.  Producer w (0 .. 3) owns the global steps G = w (mod 4) of its workgroup;
// src_of(G) = this lane's first 16 bytes of the step's V rows (row kk of its eight at + kk * cols).  NS = ring stages (4 or 8):
// NS / 4 load sets per producer are in flight, each requested NS steps before its stage is read.  Step G is written right after
// barrier G - NS (its slot was last read for step G - NS), i.e. in interval G - NS + 1.  Barriers: one before the loop + one per
// global step, in every wavefront of the workgroup.
// LAB (lab builds): 1 = load nothing, 8 = neither split nor write, 32 = no barrier in the loop.
template <int NS, int LAB, int NP, typename SrcOf>
__device__ __forceinline__ void ws_produce(char* lds, int lane, int w, int gtot, int64_t cols, SrcOf&& src_of) {
    // NP producer wavefronts per workgroup (4, 2 or 1): producer w owns the steps G = w (mod NP) and keeps NS / NP (1 or 2) load
    // sets in flight.  Fewer producers = fewer 224-register wavefronts that only need 90: a SIMD without one has 288 registers
    // left for the other chain group's table / transform wavefronts (the allocation of a kernel is uniform over its wavefronts)
    constexpr int DEPTH = NS / NP;
    static_assert(DEPTH == 1 || DEPTH == 2, "one or two load sets per producer");
    f32x4 set0[8], set1[8];                          // (set1 is dead code with a ring of 4)
    auto issue = [&](int G, f32x4 (&raw)[8]) {
        if constexpr (LAB & 1) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(raw[kk]));
            return;
        }
        // ORDINARY loads, waited for by the compiler's own s_waitcnt in front of their first use (write_stage, an LDS ring
        // later).  The first version used asynchronous inline-asm loads with hand-counted waits, like the consumers' fragment
        // loads: once the producers' control flow grew a second register set and a loop of barriers, the compiler -- which
        // cannot see that such a register is not written yet -- copied the "results" through a phi right behind the asm
        // (v_mov of a register whose load had not landed) and the data arrived in registers that by then held addresses:
        // memory faults on every launch (r06l).  s_barrier and the "memory" asm below keep these loads where they are issued.
        const float* src0 = src_of(G);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) raw[kk] = *reinterpret_cast<const f32x4*>(src0 + (int64_t)kk * cols);
        asm volatile("" ::: "memory");
    };
    auto write_stage = [&](int G, f32x4 (&raw)[8]) {
        if constexpr (LAB & 8) return;
        char* st = lds + (G & (NS - 1)) * W_STAGE + lane * 16;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {            // truncation split into three bf16 limbs: exact (as the o2 shape's split_pair)
            uint32_t u0[8], u1[8];
            float r1[8], r2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u0[e] = __float_as_uint(raw[e][ni]) & 0xffff0000u;
#pragma unroll
            for (int e = 0; e < 8; ++e) r1[e] = raw[e][ni] - __uint_as_float(u0[e]);
#pragma unroll
            for (int e = 0; e < 8; ++e) u1[e] = __float_as_uint(r1[e]) & 0xffff0000u;
#pragma unroll
            for (int e = 0; e < 8; ++e) r2[e] = r1[e] - __uint_as_float(u1[e]);
            Pack8 p0, p1, p2;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p0.u[kp] = __builtin_amdgcn_perm(u0[2 * kp + 1], u0[2 * kp], 0x07060302u);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p1.u[kp] = __builtin_amdgcn_perm(u1[2 * kp + 1], u1[2 * kp], 0x07060302u);
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) p2.u[kp] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * kp + 1]), __float_as_uint(r2[2 * kp]), 0x07060302u);
            *reinterpret_cast<u32x4*>(st + (ni * 3 + 0) * 1024) = p0.u;
            *reinterpret_cast<u32x4*>(st + (ni * 3 + 1) * 1024) = p1.u;
            *reinterpret_cast<u32x4*>(st + (ni * 3 + 2) * 1024) = p2.u;
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): what this wavefront wrote is in LDS (its loads stay in flight)
        __builtin_amdgcn_s_barrier();
    };
    // the first NS steps: requested together, written as they land
    if (w < gtot) issue(w, set0);
    if (DEPTH == 2 && w + NP < gtot) issue(w + NP, set1);
    if (w < gtot) {
        write_stage(w, set0);
        if (w + NS < gtot) issue(w + NS, set0);
    }
    if (DEPTH == 2 && w + NP < gtot) {
        write_stage(w + NP, set1);
        if (w + NP + NS < gtot) issue(w + NP + NS, set1);
    }
    barrier();
    int passed = 0;                                    // loop barriers this wavefront has gone through
    auto upto = [&](int b) {
        while (passed < b) {
            if constexpr (LAB & 32) __builtin_amdgcn_s_waitcnt(0xC07F);
            else barrier();
            ++passed;
        }
    };
    for (int G = w + NS; G < gtot; G += NP * DEPTH) {
        upto(G - NS + 1);
        write_stage(G, set0);
        if (G + NS < gtot) issue(G + NS, set0);
        if (DEPTH == 2 && G + NP < gtot) {
            upto(G + NP - NS + 1);
            write_stage(G + NP, set1);
            if (G + NP + NS < gtot) issue(G + NP + NS, set1);
        }
    }
    upto(gtot);
}

// Issue priority of the wave-specialised kernels' wavefronts (s_setprio: priority first, then age, decides which wavefront of
// a SIMD issues).  In the pipeline a GEMM of one chain group starts beside a table or transform kernel of the other group that is
// OLDER, so at equal priority it gets the issue slots those leave.  BS_GEMM_PRIO_CONS / BS_GEMM_PRIO_PROD (0..3; the serial coder
// kernels run at 3) are build-time knobs for that experiment (visit r06u).
#ifndef BS_GEMM_PRIO_CONS
#define BS_GEMM_PRIO_CONS 0
#endif
#ifndef BS_GEMM_PRIO_PROD
#define BS_GEMM_PRIO_PROD 0
#endif
__device__ __forceinline__ void ws_priority(int wave) {
    if (BS_GEMM_PRIO_CONS == BS_GEMM_PRIO_PROD) {
        if (BS_GEMM_PRIO_CONS) __builtin_amdgcn_s_setprio(BS_GEMM_PRIO_CONS);
    } else if (wave < 4) __builtin_amdgcn_s_setprio(BS_GEMM_PRIO_CONS);
    else __builtin_amdgcn_s_setprio(BS_GEMM_PRIO_PROD);
}

// LAB != 0 (-DBS_GEMM_LAB builds only; WRONG results, timing experiments): 1 = the producers load nothing (they split what is in
// their registers), 2 = the consumers store nothing, 4 = the consumers load no U fragments, 8 = the producers neither split nor
// write LDS, 16 = one MFMA per tile instead of 12, 32 = no s_barrier inside the loop
template <int NPROD, int LAB = 0, int NS = W_NS, int NP = 4>
__global__ __launch_bounds__(256 + 64 * NP)
void k_wino_gemm_bf16x3_ws(const uint16_t* __restrict__ Uf, const float* __restrict__ V, float* __restrict__ M, int T, int Cout,
                           int Cin, int64_t cols, int ncc, int nrt) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [W_NS][4 tiles][3 limbs][64 lanes][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ws_priority(wave);
    const int l32 = lane & 31, g = lane >> 5;
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);  // an XCD works through a contiguous eighth of the chunk list
    const int t = wg / (nrt * ncc), rem = wg - t * (nrt * ncc), rt = rem / ncc, cc = rem - rt * ncc;
    const int co0 = rt * X_BM;
    const int64_t c0 = (int64_t)cc * W_BN;
    const int cols_left = (int)min((int64_t)W_BN, cols - c0);
    const int nk = Cin / X_BK, nrt32 = (Cout + 31) / 32;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const float* vsrc = V + ((int64_t)t * Cin + 8 * g) * cols + c0 + (4 * l32 < cols_left ? 4 * l32 : 0);
        ws_produce<NS, LAB, NP>(lds, lane, wave - 4, nk, cols, [&](int s) { return vsrc + (int64_t)s * X_BK * cols; });
        return;
    }
    // ---------------------------------------------------------------------- consumers
    const uint16_t* a_base[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r32 = min(co0 / 32 + wave * 2 + mi, nrt32 - 1);
        a_base[mi] = Uf + (((int64_t)t * nrt32 + r32) * nk * 3 * 64 + lane) * 8;
    }
    // The consumers' fragment loads stay asynchronous inline-asm loads with hand-placed waits (load_a / landed_a), unlike the
    // producers' (see ws_produce): with ordinary loads the compiler's own s_waitcnt is exact in the first block of the unrolled pair
    // (vmcnt(11) .. vmcnt(6)) but conservative in the second (vmcnt(5) .. vmcnt(0): it waits for the six loads it has just issued,
    // ~500 cycles per pair of blocks; round 6, measured on the ISA).  There is no phi on this path -- the double buffer is indexed
    // at compile time -- and every build is held to bitwise equality with the other launch shapes by the GPU suite.
    bf16x8 a[2][2][3], bq[2][3];
    auto load_a = [&](int kb, auto P) {
        constexpr int q = decltype(P)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                const uint16_t* src = a_base[mi] + ((int64_t)kb * 3 + i) * 64 * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(p.u) : "v"(src) : "memory");
                a[q][mi][i] = p.b;
            }
    };
    auto landed_a = [&](auto P) {
        constexpr int q = decltype(P)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                p.b = a[q][mi][i];
                asm volatile("" : "+v"(p.u));
                a[q][mi][i] = p.b;
            }
    };
    auto read_tile = [&](int s, int ni, int dst) {     // the three limb fragments of column tile ni of k block s
        const char* st = lds + (s & (NS - 1)) * W_STAGE + lane * 16 + ni * 3 * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Pack8 p;
            p.u = *reinterpret_cast<const u32x4*>(st + i * 1024);
            bq[dst][i] = p.b;
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;

    load_a(0, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();                      // stages 0 .. 3 are in LDS
    read_tile(0, 0, 0);
    auto step = [&](int b, auto P) {
        constexpr int q = decltype(P)::value;
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < nk && !(LAB & 4)) {
            load_a(b + 1, std::integral_constant<int, q ^ 1>{});
            __builtin_amdgcn_s_waitcnt(0x0F76);        // vmcnt(6): all but the six just issued -> the fragments of block b are here
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        landed_a(P);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            __builtin_amdgcn_sched_barrier(0);
            if (ni < 3) read_tile(b, ni + 1, (ni + 1) & 1);
            else if (b + 1 < nk) read_tile(b + 1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = ((LAB & 16) ? 8 : 9 - NPROD); p < 9; ++p)
#pragma unroll
                for (int mi = 0; mi < ((LAB & 16) ? 1 : 2); ++mi)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][mi][X_ORDER9[p][0]], bq[ni & 1][X_ORDER9[p][1]], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(LAB & 32)) __builtin_amdgcn_s_barrier();   // block b is read everywhere: its slot goes back to the producers
    };
    int b = 0;
    for (; b + 1 < nk; b += 2) {
        step(b, std::integral_constant<int, 0>{});
        step(b + 1, std::integral_constant<int, 1>{});
    }
    if (b < nk) step(b, std::integral_constant<int, 0>{});
    float* Mt = M + (int64_t)t * Cout * cols;
    const int cl = 4 * l32;
    if ((LAB & 2) && acc[0][0][0] != 12345.678f) return;
    if (cl < cols_left) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int row0 = co0 + wave * 64 + mi * 32 + g * 4;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = row0 + (v >> 2) * 8 + (v & 3);
                if (row < Cout) {
                    f32x4 o = {acc[mi][0][v], acc[mi][1][v], acc[mi][2][v], acc[mi][3][v]};
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Mt + (int64_t)row * cols + c0 + cl));
                }
            }
        }
    }
}


// ---- the same, PERSISTENT: one workgroup per CU works through a contiguous range of (t, row tile, 128-column chunk) units without
// ever draining its LDS ring.  Why: alone, the one-unit-per-workgroup kernel above runs the T36 x 256^2 x 8000 launch in 227 us --
// 43k cycles per unit of which 24.6k are MFMA.  A workgroup of 8 x 218 registers fills the CU, so nothing overlaps its prologue
// (the first V stages come from HBM: ~4k cycles) or its epilogue (128 KB of M per unit at the CU's share of HBM: ~9k cycles, and a
// wavefront only retires when its stores are acknowledged).  Here the producers simply keep going -- the stage of global step
// G = unit * nk + k block belongs to whichever unit that is -- and a consumer issues its 32 stores and carries on with the next
// unit's first block, whose operands are already in LDS: the stores drain under the next 1,536 cycles of MFMAs.
// Needs an even number of k blocks (the U fragments are double-buffered by block parity); otherwise the kernel above runs.
// Same arithmetic, same bits.
struct WsUnit {
    int t, rt, cc;
};
__device__ __forceinline__ WsUnit ws_unit(int u, int nrt, int ncc) {
    WsUnit r;
    r.t = u / (nrt * ncc);
    const int rem = u - r.t * (nrt * ncc);
    r.rt = rem / ncc;
    r.cc = rem - r.rt * ncc;
    return r;
}
template <int NPROD, int NS = W_NS, int NP = 4>
__global__ __launch_bounds__(256 + 64 * NP)
void k_wino_gemm_bf16x3_wsp(const uint16_t* __restrict__ Uf, const float* __restrict__ V, float* __restrict__ M, int T, int Cout,
                            int Cin, int64_t cols, int ncc, int nrt, int nunits) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [W_NS][4 tiles][3 limbs][64 lanes][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ws_priority(wave);
    const int l32 = lane & 31, g = lane >> 5;
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);  // an XCD works through a contiguous eighth of the unit list
    const int u0 = (int)(((int64_t)wg * nunits) / nwg), u1 = (int)(((int64_t)(wg + 1) * nunits) / nwg);
    const int nk = Cin / X_BK, nrt32 = (Cout + 31) / 32;
    const int gtot = (u1 - u0) * nk;                   // global steps of this workgroup
    if (gtot == 0) return;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        ws_produce<NS, 0, NP>(lds, lane, wave - 4, gtot, cols, [&](int G) {
            const int du = G / nk, kb = G - du * nk;
            const WsUnit un = ws_unit(u0 + du, nrt, ncc);
            const int64_t c0 = (int64_t)un.cc * W_BN;
            const int cols_left = (int)min((int64_t)W_BN, cols - c0);
            return V + ((int64_t)un.t * Cin + (int64_t)kb * X_BK + 8 * g) * cols + c0 + (4 * l32 < cols_left ? 4 * l32 : 0);
        });
        return;
    }
    // ---------------------------------------------------------------------- consumers
    const uint16_t* a_base[2];
    auto set_a_base = [&](const WsUnit& un) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int r32 = min(un.rt * (X_BM / 32) + wave * 2 + mi, nrt32 - 1);
            a_base[mi] = Uf + (((int64_t)un.t * nrt32 + r32) * nk * 3 * 64 + lane) * 8;
        }
    };
    bf16x8 a[2][2][3], bq[2][3];
    auto load_a = [&](int kb, auto P) {
        constexpr int q = decltype(P)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                const uint16_t* src = a_base[mi] + ((int64_t)kb * 3 + i) * 64 * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(p.u) : "v"(src) : "memory");
                a[q][mi][i] = p.b;
            }
    };
    auto landed_a = [&](auto P) {
        constexpr int q = decltype(P)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pack8 p;
                p.b = a[q][mi][i];
                asm volatile("" : "+v"(p.u));
                a[q][mi][i] = p.b;
            }
    };
    auto read_tile = [&](int G, int ni, int dst) {     // the three limb fragments of column tile ni of global step G
        const char* st = lds + (G & (NS - 1)) * W_STAGE + lane * 16 + ni * 3 * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Pack8 p;
            p.u = *reinterpret_cast<const u32x4*>(st + i * 1024);
            bq[dst][i] = p.b;
        }
    };
    f32x16 acc[2][4];
    WsUnit un = ws_unit(u0, nrt, ncc);
    set_a_base(un);
    load_a(0, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();                      // stages 0 .. 3 are in LDS
    read_tile(0, 0, 0);
    // One k block.  P: parity of the block (which fragment buffer).  waited (wave-uniform): the block right behind the stores
    // of the previous unit -- its fragments were waited for BEFORE those stores were issued (see the unit loop), so that the
    // wait here does not have to reach across them.  (The first version counted: vmcnt(38) = the newest 6 loads + 32 stores.  A
    // wavefront whose rows or columns lie partly outside the matrix issues FEWER than 32 stores -- T7 x 700 x 64 x 40000: the
    // unit behind (t, row tile 2) multiplied fragments that had not landed.  Counting instructions that may not be issued is
    // not a wait.)
    auto step = [&](int G, int kb_next, bool more, bool waited, auto P) {
        constexpr int q = decltype(P)::value;
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_a(kb_next, std::integral_constant<int, q ^ 1>{});
        if (!waited) {
            if (more) __builtin_amdgcn_s_waitcnt(0x0F76);                   // vmcnt(6): all but the six just issued
            else __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
        }
        landed_a(P);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            __builtin_amdgcn_sched_barrier(0);
            if (ni < 3) read_tile(G, ni + 1, (ni + 1) & 1);
            else if (G + 1 < gtot) read_tile(G + 1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 9 - NPROD; p < 9; ++p)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][mi][X_ORDER9[p][0]], bq[ni & 1][X_ORDER9[p][1]], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // the block is read everywhere: its slot goes back to the producers
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    int G = 0;
    for (int u = u0; u < u1; ++u) {
        const bool last_unit = u + 1 == u1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
        for (int b = 0; b < nk; b += 2) {               // blocks b (even), b + 1 (odd): nk is even
            step(G, b + 1, true, b == 0 && u != u0, C0{});
            ++G;
            const bool tail = b + 2 == nk;              // the unit's last block: its prefetch is block 0 of the NEXT unit
            if (tail) set_a_base(ws_unit(last_unit ? u : u + 1, nrt, ncc));
            step(G, tail ? 0 : b + 2, !(tail && last_unit), false, C1{});
            ++G;
        }
        if (!last_unit) {       // block 0 of the next unit: requested a whole block ago -- wait for it HERE, in front of the stores
            __builtin_amdgcn_s_waitcnt(0x0F70);
            landed_a(C0{});
        }
        const int co0 = un.rt * X_BM;
        const int64_t c0 = (int64_t)un.cc * W_BN;
        const int cols_left = (int)min((int64_t)W_BN, cols - c0);
        float* Mt = M + (int64_t)un.t * Cout * cols;
        const int cl = 4 * l32;
        if (cl < cols_left) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row0 = co0 + wave * 64 + mi * 32 + g * 4;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = row0 + (v >> 2) * 8 + (v & 3);
                    if (row < Cout) {
                        f32x4 o = {acc[mi][0][v], acc[mi][1][v], acc[mi][2][v], acc[mi][3][v]};
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(Mt + (int64_t)row * cols + c0 + cl));
                    }
                }
            }
        }
        if (!last_unit) un = ws_unit(u + 1, nrt, ncc);
    }
}


// (A third shape -- V split ONCE per workgroup, each thread 8 floats fetched straight from global memory two steps ahead, the
// limbs written to LDS in MFMA B-fragment layout: 44 VALU per wavefront and K step instead of 176 -- was built and measured in
// round 4, visit N: 232-250 us against 221-239 us for the shape above on the 36 x 256 x 256 x 8000 launch, with or without a
// second register buffer for the U fragments.  The matrix pipe's idle share is not the split; LABNOTES.md has the numbers.
// The kernel is not kept.)

}  // namespace

extern "C" int bs_wino_gemm_bf16x3(const uint16_t* U_frags, const float* V, float* M, int T, int Cout, int Cin, int64_t cols,
                                   int nprod, void* stream) {
    if (!U_frags || !V || !M || T < 0 || T > 65535 || Cout < 1 || Cin < 1 || cols < 0 || (nprod != 6 && nprod != 9)) return BS_EINVAL;
    if (Cin % X_BK != 0 || cols % 4 != 0) return BS_EUNSUPPORTED;
    if (((uintptr_t)U_frags | (uintptr_t)V | (uintptr_t)M) & 15u) return BS_EINVAL;
    if (T == 0 || cols == 0) return BS_OK;
    const int64_t ncc = (cols + X_BN - 1) / X_BN, nrt = (Cout + X_BM - 1) / X_BM;
    const int64_t wgs = (int64_t)T * nrt * ncc;
    if (wgs > 0x7fffffff || (int64_t)Cin * cols > 0x7fffffff) return BS_EUNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    size_t shm = 3 * (size_t)X_STAGE;
    const long lds_pad = [] { const char* e = getenv("BITSWAP_BF16X3_LDS_PAD"); return e ? atol(e) : 0L; }();   // diagnostics
    shm += (size_t)lds_pad;
    if (shm > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
    }
#ifdef BS_GEMM_LAB
    if (const char* e = getenv("BITSWAP_BF16X3_LAB")) {
        const int lab = atoi(e);
        if (lab == 4) hipLaunchKernelGGL((k_wino_gemm_bf16x3<6, 4>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
        else if (lab == 6) hipLaunchKernelGGL((k_wino_gemm_bf16x3<6, 6>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
        else if (lab == 1) hipLaunchKernelGGL((k_wino_gemm_bf16x3<6, 1>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
        else if (lab == 2) hipLaunchKernelGGL((k_wino_gemm_bf16x3<6, 2>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
        else hipLaunchKernelGGL((k_wino_gemm_bf16x3<6, 3>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
        return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
    }
#endif
    // BITSWAP_BF16X3_SHAPE=3 (default since round 6: wave-specialised), =2 (two 256 x 128 workgroups per CU: rounds 4-5),
    // =1 (the one-workgroup-per-CU shape) and BITSWAP_BF16X3_DIAG (noclaim | noclaim_strict: the kernels
    // WITHOUT the whole-register-share claim, with counted or with vmcnt(0) waits) exist for the co-residency diagnostics
    // only: the unclaimed shape-1 kernel gave wrong products beside a small wavefront of another kernel (LABNOTES r04/r05)
    const char* shape_env = getenv("BITSWAP_BF16X3_SHAPE");
    const char* diag = getenv("BITSWAP_BF16X3_DIAG");
    const int shape = shape_env ? atoi(shape_env) : (diag ? 2 : 3);     // default: the wave-specialised shape (diagnostics: o2)
    const int dg = !diag ? 0 : !strcmp(diag, "noclaim") ? 1 : !strcmp(diag, "noclaim_strict") ? 2 : !strcmp(diag, "stray_exit") ? 3
                   : !strcmp(diag, "noclaim_late") ? 4 : !strcmp(diag, "noclaim_plainstore") ? 5 : !strcmp(diag, "noclaim_coherent") ? 6 : -1;
    if (dg < 0 || (shape != 1 && shape != 2 && shape != 3) || (dg && nprod != 6) || (dg && shape == 3) || (dg == 3 && shape != 1) || (dg >= 4 && shape != 2)) return BS_EINVAL;
    if (shape == 3) {                     // wave-specialised: one multiplying wavefront per SIMD, the split on wavefronts of its own
        const int64_t ncc3 = (cols + W_BN - 1) / W_BN, wgs3 = (int64_t)T * nrt * ncc3;
        if (wgs3 > 0x7fffffff) return BS_EUNSUPPORTED;
        // ring stages: 4 (48 KB) or 8 (96 KB: two load sets per producer in flight, each requested 8 k blocks ahead)
        const char* re = getenv("BITSWAP_BF16X3_RING");
        const int ring = re ? atoi(re) : W_NS;
        if (ring != 4 && ring != 8) return BS_EINVAL;
        const size_t shm3 = (size_t)ring * W_STAGE;
        if (ring == 8) {
            static bool raised3 = false;
            if (!raised3) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3_ws<6, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3_ws<9, 0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3_wsp<6, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3_wsp<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                raised3 = true;
            }
        }
        const int nk3 = Cin / X_BK;
        // producer wavefronts per workgroup: 4 (one per SIMD) or 2 (nprod 6, ring of 4 only) -- BITSWAP_BF16X3_PRODUCERS
        const char* npe = getenv("BITSWAP_BF16X3_PRODUCERS");
        const int nprw = npe ? atoi(npe) : 4;
        if (nprw != 4 && nprw != 2) return BS_EINVAL;
#ifdef BS_GEMM_LAB
        if (const char* e = getenv("BITSWAP_BF16X3_WSLAB")) {
            const int lab = atoi(e);
#define BS_WSLAB(L) case L: hipLaunchKernelGGL((k_wino_gemm_bf16x3_ws<6, L>), dim3((unsigned)wgs3), dim3(512), shm3, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc3, (int)nrt); break;
            switch (lab) {
                BS_WSLAB(0) BS_WSLAB(1) BS_WSLAB(2) BS_WSLAB(3) BS_WSLAB(4) BS_WSLAB(7) BS_WSLAB(8) BS_WSLAB(9) BS_WSLAB(15) BS_WSLAB(16) BS_WSLAB(23) BS_WSLAB(32) BS_WSLAB(47)
                default: return BS_EINVAL;
            }
#undef BS_WSLAB
            return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
        }
#endif
        static const int cus = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        // Persistent workgroups (one per CU, a contiguous range of units each), one unit per workgroup, or something between?  Alone
        // the persistent kernel is 3-5 % faster at every size (profiles/r06f_gemm_times.txt); in the two-group pipeline of 1000 chains
        // it is SLOWER (161-165 ms per step against 159: a persistent grid holds every CU for the whole launch, the other chain
        // group's table and transform kernels -- which do not fit beside it -- wait for its end instead of slipping in between
        // 16-microsecond workgroups), at 100 chains per call faster (21.9 against 22.4 ms).  Between the two: workgroups of TWO
        // consecutive units (the persistent kernel on wgs / 2 workgroups) -- every second prologue is hidden behind the ring, the
        // grid still has thousands of workgroups to slip in between: alone no faster than one unit per workgroup (ragged tail), in
        // the pipeline 150.5 against 152.9 ms per step at 1000 chains, imagenet4 119.1 against 120.5; three and four units the same,
        // a fixed 2 .. 6 workgroups per CU -0.5 % (profiles/r06M_*, r06N_*).  So: one workgroup per CU up to four units per CU, two
        // units per workgroup above.  BITSWAP_BF16X3_PERSISTENT = 0 / 1 forces one unit per workgroup / one workgroup per CU,
        // BITSWAP_BF16X3_UNITS = k forces k units per workgroup (a multiple of 8 workgroups keeps the XCD mapping),
        // BITSWAP_BF16X3_WGS_PER_CU = r forces r workgroups per CU in all (read per launch: tests flip them).  Same bits every way.
        const char* pe = getenv("BITSWAP_BF16X3_PERSISTENT");
        const char* ke = getenv("BITSWAP_BF16X3_UNITS");
        const char* re2 = getenv("BITSWAP_BF16X3_WGS_PER_CU");
        const int per_cu = re2 ? atoi(re2) : 0;
        const bool small = wgs3 <= 4 * (int64_t)cus;
        const int kunits = ke ? atoi(ke) : (pe || per_cu > 0 || small) ? 0 : 2;
        const int persistent = (kunits > 1 || per_cu > 0) ? 1 : pe ? atoi(pe) : small;
        if (persistent && nk3 >= 2 && nk3 % 2 == 0) {
            int nwg = (int)(wgs3 < cus ? wgs3 : cus);
            if (kunits > 1) {
                const int64_t want = (wgs3 + kunits - 1) / kunits;
                nwg = (int)(want < 8 ? want : (want + 7) / 8 * 8);
                if (nwg > wgs3) nwg = (int)wgs3;
            } else if (per_cu > 0) {
                const int64_t want = (int64_t)cus * per_cu;
                nwg = (int)(want < wgs3 ? want : wgs3);
            }
#define BS_WSP(NPR, RING, NPW) hipLaunchKernelGGL((k_wino_gemm_bf16x3_wsp<NPR, RING, NPW>), dim3((unsigned)nwg), dim3(256 + 64 * NPW), shm3, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc3, (int)nrt, (int)wgs3)
            if (nprod == 9) { if (ring == 8) BS_WSP(9, 8, 4); else BS_WSP(9, 4, 4); }
            else if (ring == 8) BS_WSP(6, 8, 4);
            else if (nprw == 2) BS_WSP(6, 4, 2);
            else BS_WSP(6, 4, 4);
#undef BS_WSP
            return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
        }
#define BS_WS(NPR, RING, NPW) hipLaunchKernelGGL((k_wino_gemm_bf16x3_ws<NPR, 0, RING, NPW>), dim3((unsigned)wgs3), dim3(256 + 64 * NPW), shm3, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc3, (int)nrt)
        if (nprod == 9) { if (ring == 8) BS_WS(9, 8, 4); else BS_WS(9, 4, 4); }
        else if (ring == 8) BS_WS(6, 8, 4);
        else if (nprw == 2) BS_WS(6, 4, 2);
        else BS_WS(6, 4, 4);
#undef BS_WS
        return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
    }
#define BS_X3_O2(NP, CL, ST) hipLaunchKernelGGL((k_wino_gemm_bf16x3_o2<NP, CL, ST>), dim3((unsigned)wgs2), dim3(X_NT), shm2, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc2, (int)nrt)
#define BS_X3_O1(NP, CL, ST) hipLaunchKernelGGL((k_wino_gemm_bf16x3<NP, 0, CL, ST>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt)
    if (shape == 2) {                     // two workgroups of 256 x 128 per CU (default)
        const int64_t ncc2 = (cols + Y_BN - 1) / Y_BN, wgs2 = (int64_t)T * nrt * ncc2;
        if (wgs2 > 0x7fffffff) return BS_EUNSUPPORTED;
        const size_t shm2 = 3 * (size_t)Y_STAGE;
        if (dg == 1) BS_X3_O2(6, false, false);
        else if (dg == 2) BS_X3_O2(6, false, true);
        else if (dg == 4) hipLaunchKernelGGL((k_wino_gemm_bf16x3_o2<6, false, false, true>), dim3((unsigned)wgs2), dim3(X_NT), shm2, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc2, (int)nrt);
        else if (dg == 5) hipLaunchKernelGGL((k_wino_gemm_bf16x3_o2<6, false, false, false, true>), dim3((unsigned)wgs2), dim3(X_NT), shm2, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc2, (int)nrt);
        else if (dg == 6) hipLaunchKernelGGL((k_wino_gemm_bf16x3_o2<6, false, false, false, false, true>), dim3((unsigned)wgs2), dim3(X_NT), shm2, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc2, (int)nrt);
        else if (nprod == 9) BS_X3_O2(9, true, false);
        else BS_X3_O2(6, true, false);
        return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
    }
    if (dg) {
        static bool raised_d = false;
        if (!raised_d) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3<6, 0, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3<6, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wino_gemm_bf16x3<6, 0, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised_d = true;
        }
        if (dg == 1) BS_X3_O1(6, false, false);
        else if (dg == 3) hipLaunchKernelGGL((k_wino_gemm_bf16x3<6, 0, false, false, true>), dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
        else BS_X3_O1(6, false, true);
        return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
    }
    if (nprod == 9)
        hipLaunchKernelGGL(k_wino_gemm_bf16x3<9>, dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
    else
        hipLaunchKernelGGL(k_wino_gemm_bf16x3<6>, dim3((unsigned)wgs), dim3(X_NT), shm, st, U_frags, V, M, T, Cout, Cin, cols, (int)ncc, (int)nrt);
    return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH;
}
