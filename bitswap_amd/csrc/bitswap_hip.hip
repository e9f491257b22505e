// bitswap_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the Bit-Swap / BB-ANS hot path.
//
// What runs where (see DESIGN.md for the roofline of each kernel):
//   k_logistic<NPL,PT,MODE>    one 64-lane wavefront per (latent dim d, group of chains); the
//                              K-1 float64 bin endpoints of dim d stay in registers (NPL = K/64
//                              consecutive bins per lane) and are reused for every chain of the
//                              group; per chain: NPL deterministic float64 sigmoids per lane,
//                              adjacent difference, trunc-multiply, wave-wide sum / first-argmax
//                              (DPP + scalar unit), remnant bump, exclusive scan (DPP) -> integer cdf
//                              row (decode flavour; wave-native rows go through an LDS transpose and
//                              leave as streaming 1-KB stores) or the (f, c) pair of one symbol
//                              (encode flavour).
//   k_table_rows / _generic    the same integer tail for caller-supplied float64 pmf rows
//                              (bit-exact ANS.__init__).
//   k_rans_pop_wave<NR,PF>     BS_LAYOUT_WAVE rows, one wavefront per chain: register-pivot two-ballot
//                              search, PF rows in flight through buffer loads, the 64-bit head on the
//                              scalar unit.
//   k_rans_pop<ROW,PF> /       the reference's linear rows (drop-in ANS class, tests): 16-byte loads +
//   k_rans_pop_generic         popcount of ballots / any K, any alignment.
//   k_rans_push / _table       one wavefront per chain run as a 64-lane systolic array: lane i owns
//                              symbol i of a 64-symbol chunk, the head moves up one lane per step.
//
// Reference lines are cited in include/bitswap_hip.h next to each entry point.
#include <hip/hip_runtime.h>

// issue priority of the serial coder kernels' wavefronts (one per chain) against co-resident bulk kernels (0 .. 3)
#ifndef BS_SERIAL_PRIO
#define BS_SERIAL_PRIO 3
#endif
// rows of pivots + anchors k_rans_pop_pivot keeps in flight: the fewer registers the coder wavefronts hold, the less they
// cost the bulk kernels they sit beside -- 16 rows: 125 registers, step 188.4 ms; 8: 92, 181.5; 4: 76,
// 180.0 (A/B on one box each; 4 rows are still ~3 us ahead of their use)
#ifndef BS_POP_PF
#define BS_POP_PF 4
#endif
#include <stdint.h>

#include "../../include/bitswap_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// wave64 primitives (DPP: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143)
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t v) {
    // lanes whose source is out of range, or whose row is masked off, read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
    v += dpp_or0<0x111, 0xf>(v);
    v += dpp_or0<0x112, 0xf>(v);
    v += dpp_or0<0x114, 0xf>(v);
    v += dpp_or0<0x118, 0xf>(v);
    v += dpp_or0<0x142, 0xa>(v);
    v += dpp_or0<0x143, 0xc>(v);
    return v;
}

// maximum over the 64 lanes, returned in every lane (wave-uniform)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_or0<0x111, 0xf>(v));
    v = max(v, dpp_or0<0x112, 0xf>(v));
    v = max(v, dpp_or0<0x114, 0xf>(v));
    v = max(v, dpp_or0<0x118, 0xf>(v));
    v = max(v, dpp_or0<0x142, 0xa>(v));
    v = max(v, dpp_or0<0x143, 0xc>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ double lane_shift_up_f64(double v) {
    // value of lane-1 (lane 0 receives garbage; caller overrides)
    return __shfl_up(v, 1, 64);
}

// ------------------------------------------------------------------------------------------
// Deterministic float64 sigmoid -- BS_CDF_SPEC 1 (DESIGN.md "Deterministic logistic CDF").
// IEEE-754 binary64 sub/mul/min/max/rint/fma/ldexp/add + a correctly rounded reciprocal only;
// compiled with -ffp-contract=off so nothing else is fused.  oracle/bitswap_oracle.c carries
// an independent C restatement that must agree bit for bit.
// ------------------------------------------------------------------------------------------
// Correctly rounded 1/x for x in [1, 2^1011): hardware seed (|rel err| <= 2^-24.4 measured), ONE cubic
// Newton step (-> 2^-73), one residual correction.  The spec demands RN(1/x), whatever the seed: this
// sequence agrees with IEEE division on 1.7e10 arguments incl. the all-ones-mantissa hard cases
// (tools/probes/recip_check.hip), and parity with the oracle's `1.0 / x` is asserted bit for bit
// (tests/test_hip_parity.py::test_sigmoid_bit_exact_vs_oracle).  v_rcp_f64 issues at quarter rate
// (tools/probes/instr_rate.hip): 4 + 5 issue slots here against 4 + 9 for hipcc's generic f64 division.
__device__ __forceinline__ double recip_1_to_huge(double x) {
    double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    const double t = fma(e, e, e);
    y = fma(y, t, y);
    const double r = fma(-x, y, 1.0);
    return fma(r, y, y);
}

// exp(a) for the clamped argument: the exponential half of the deterministic sigmoid
__device__ __forceinline__ double det_exp(double a) {
    a = fmin(fmax(a, -700.0), 700.0);
    const double kd = rint(a * 0x1.71547652b82fep+0);
    double r = fma(-kd, 0x1.62e42fee00000p-1, a);
    r = fma(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.af631e4ea6521p-26;
    p = fma(p, r, 0x1.28b4068ef93d2p-22);
    p = fma(p, r, 0x1.71ddf573e8618p-19);
    p = fma(p, r, 0x1.a01991ab61789p-16);
    p = fma(p, r, 0x1.a01a01b143bc8p-13);
    p = fma(p, r, 0x1.6c16c187fc4dep-10);
    p = fma(p, r, 0x1.111111110f224p-7);
    p = fma(p, r, 0x1.555555554f0ccp-5);
    p = fma(p, r, 0x1.555555555555ap-3);
    p = fma(p, r, 0x1.0000000000011p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}

__device__ __forceinline__ double det_sigmoid(double t) { return recip_1_to_huge(1.0 + det_exp(-t)); }

// RN(1 / scale): the same Newton sequence.  It is invariant under scaling by powers of two as long as x and 1/x stay
// normal, which every positive finite scale a model head can emit satisfies by hundreds of binades; zero, negative,
// Inf, NaN and subnormal scales come out as NaN / Inf / a negative number and the caller's validity check (scale > 0,
// rs > 0) flags the chain, as it would after IEEE division.  Saves the v_div_scale / v_div_fmas / v_div_fixup
// scaffolding per row.
__device__ __forceinline__ double recip_scale(double x) { return recip_1_to_huge(x); }

// ------------------------------------------------------------------------------------------
// integer tail shared by the table kernels.  A lane holds t[i] = trunc(pmf * M) of NPL consecutive bins
// (the reference's frequency is f = t + 1, mnist_compress.py:30,33; the +1 is folded into the sums and
// into the running cdf `c = c + t + 1`, one v_add3_u32).  On return the first maximal bin has absorbed the
// remnant 2^bits - sum f (first max wins like torch.argmax, :36) and the result is the lane's starting
// cumulative value.  Per bin this costs 1/2 add3 + 1/2 max3 + 1 compare on the VALU; which bin of the
// winning lane is maximal is resolved on the scalar unit, and the bump is ONE scalar-indexed register add.
// ------------------------------------------------------------------------------------------
template <int NPL>
struct Bins {
    typedef uint32_t vec_t __attribute__((ext_vector_type(NPL)));
    vec_t t;
};
template <>
struct Bins<1> {
    struct vec_t {
        uint32_t x;
        __device__ __forceinline__ uint32_t& operator[](int) { return x; }
        __device__ __forceinline__ const uint32_t& operator[](int) const { return x; }
    };
    vec_t t;
};

template <int NPL>
__device__ __forceinline__ uint32_t bump_and_scan(Bins<NPL>& bn, int lane, int bits, bool& bad, uint32_t* bumped_bin = nullptr,
                                                  uint32_t* remnant = nullptr) {
    uint32_t tsum = NPL, best = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        tsum += bn.t[i];
        best = max(best, bn.t[i]);
    }
    const uint32_t incl0 = wave_incl_scan_add(tsum);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
    const uint32_t mx = wave_max_u32(best);
    const int first = __ffsll((long long)__ballot(best == mx)) - 1;  // wave-uniform
    // first maximal bin inside lane `first`: bit `first` of the 64-lane compare masks, descending so the
    // smallest index is the one that sticks
    int barg = 0;
#pragma unroll
    for (int i = NPL - 1; i >= 0; --i) {
        const unsigned long long eq = __ballot(bn.t[i] == mx);
        barg = ((eq >> first) & 1ull) ? i : barg;
    }
    barg = __builtin_amdgcn_readfirstlane(barg);
    const uint32_t rem = (1u << bits) - total;  // two's complement: may be "negative"
    const bool mine = lane == first;
    bn.t[barg] += mine ? rem : 0u;
    bad = mine && ((int32_t)(mx + 1u + rem) < 1);
    if (bumped_bin) *bumped_bin = (uint32_t)(first * NPL + barg);   // wave-uniform: which bin took the remnant, and how much
    if (remnant) *remnant = rem;
    // exclusive prefix of the bumped per-lane sums: lanes after `first` shift by rem
    uint32_t excl = incl0 - tsum;
    if (lane > first) excl += rem;
    return excl;
}

__device__ __forceinline__ uint32_t trunc_u32(double x) { return (uint32_t)(int32_t)x; }

// ------------------------------------------------------------------------------------------
// k_logistic: fused logistic CDF -> integer table, NPL = K/64 bins per lane
// ------------------------------------------------------------------------------------------
// output modes of k_logistic
enum { M_ENCODE = 0, M_LINEAR = 1, M_LINEAR_VEC = 2, M_WAVE = 3, M_PIVOT = 4 };

// BS_LAYOUT_WAVE: dword offset of cdf entry j (K = 64*NPL entries) inside a row.  Register r = j/64 of
// the popping wavefront holds entries 64r..64r+63 across its lanes; uint4 load i of lane l returns
// registers 4i..4i+3, so entry j sits at ((r/4)*64 + l)*4 + r%4 with l = j%64.
__device__ __forceinline__ int wave_offset(int j) {
    const int r = j >> 6, l = j & 63;
    return (((r >> 2) << 6) + l) * 4 + (r & 3);
}

// UNI: BS_CDF_SPEC 2 for rows of uniform-width bins (bin width step[d]): one exponential per lane (the anchor
// A = exp(-t) of its first bin) and one per row (the geometric factors Q_b = exp(-b*h/scale), b < NPL, computed by
// lane b < NPL of every 16-lane row), then per bin 1 + E = fma(Q_b, A*(1 - eps), 1), eps = r_b / scale, where
// r_b = e_b - (e_0 + b*h) ~ 1e-16 is how far the stored endpoint sits from the ideal progression (computed once per
// wave: the registers that held the endpoints hold the residuals).  Q_b reaches every lane INSIDE the multiply-add
// (v_fmac_f64 with DPP row_newbcast:b, the one DPP control CDNA offers for 64-bit operands), so a bin costs
// 3 + 9 (correctly rounded reciprocal) + 3 (difference, scale, truncate) float64 issue slots instead of 35.
// oracle/bitswap_oracle.c::det2_row_cdf is the C restatement.
template <int B_>
__device__ __forceinline__ double fma_rowbcast(double q, double u) {
    // fma(q[lane b of this lane's 16-lane row], u, 1.0); s_nop: a DPP read needs 2 wait states after the VALU
    // write of its source, which the compiler cannot see through inline asm
    double d = 1.0;
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(q), "v"(u), "n"(B_));
    return d;
}
template <int NPL, int I>
__device__ __forceinline__ double one_plus_e(double qb, double u) {
    if constexpr (NPL <= 16) return fma_rowbcast<I>(qb, u);
    else return fma(readlane_f64(qb, I), u, 1.0);
}
template <int NPL, int I>
__device__ __forceinline__ void uni_bins(const double (&e)[NPL], double rs, double A, double qb, double M, int lane,
                                         double& prev, Bins<NPL>& bn) {
    if constexpr (I < NPL) {
        const double eps = e[I] * rs;
        const double u = fma(-A, eps, A);
        double c = recip_1_to_huge(one_plus_e<NPL, I>(qb, u));
        if (I == NPL - 1 && lane == 63) c = 1.0;
        bn.t[I] = trunc_u32((c - prev) * M);
        prev = c;
        uni_bins<NPL, I + 1>(e, rs, A, qb, M, lane, prev, bn);
    }
}

// One (chain, dim) row: bn.t[i] = trunc(pmf * M) of this lane's NPL bins (the reference's f - 1).  `e` holds the
// lane's endpoints (spec 1) or its anchor + residuals (spec 2, see k_logistic).
// Returns false when the row leaves the domain of CDF spec 2: NPL * h / scale < 650.  det_exp clamps its argument to +-700.
// A clamped ANCHOR is harmless on its own: beyond +700 every bin of the lane is exactly 1, beyond -700 the lane's bins
// come out as e^-(700 - b h/scale) <= e^-50 -- too large, but still truncated to the same f = 1 as the true values, so the
// table is the exact one.  What must not be clamped is the geometric factor Q_b = exp(-b h/scale): with h/scale in the
// hundreds (a scale tiny against the bin width: scale < 5e-5 for the pixel bins, 20x below the reference's floor of
// 2/255/8, mnist_train.py:411; reachable only through the C ABI) a lane with a clamped anchor would put 0.5 where the
// cdf is 1e-18, and the cdf would step DOWN into the next lane.  Such rows are not coded: the caller flags
// BS_ST_BADTABLE (oracle/bitswap_oracle.c::layer_in_domain applies the same test); CDF spec 1 takes any scale.
template <int NPL, bool UNI>
__device__ __forceinline__ bool logistic_row(const double (&e)[NPL], double hstep, double m_, double rs, double M, int lane,
                                             Bins<NPL>& bn) {
    double c0, prev;
    bool in_domain = true;
    if (UNI) {
        const double hr = hstep * rs;
        const double qb = det_exp(-((double)(lane & (NPL - 1)) * hr));   // lane b < NPL: Q_b
        const double ta = (e[0] - m_) * rs;
        in_domain = (double)NPL * fabs(hr) < 650.0;
        const double A = det_exp(-ta);
        c0 = recip_1_to_huge(1.0 + A);
        prev = c0;
        uni_bins<NPL, 1>(e, rs, A, qb, M, lane, prev, bn);
    } else {
        c0 = det_sigmoid((e[0] - m_) * rs);
        if (NPL == 1 && lane == 63) c0 = 1.0;
        prev = c0;
#pragma unroll
        for (int i = 1; i < NPL; ++i) {
            double c = det_sigmoid((e[i] - m_) * rs);
            if (i == NPL - 1 && lane == 63) c = 1.0;
            bn.t[i] = trunc_u32((c - prev) * M);
            prev = c;
        }
    }
    const double below = lane_shift_up_f64(prev);
    // reference: pmf[0] = cdf[0] (no subtraction), mnist_compress.py:185
    const double p0 = (lane == 0) ? c0 : c0 - below;
    bn.t[0] = trunc_u32(p0 * M);
    return in_domain;
}

template <int NPL, typename PT, int MODE, bool UNI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(UNI ? 4 : 7, 8))) void k_logistic(const double* __restrict__ endpoints, int64_t e_stride,
                                                  const double* __restrict__ step,
                                                  const PT* __restrict__ mu, const PT* __restrict__ scale,
                                                  const int32_t* __restrict__ sym, int B, int D, int bits,
                                                  int quantbits, int nb, uint32_t* __restrict__ out0,
                                                  uint32_t* __restrict__ out1, int64_t ld,
                                                  int32_t* __restrict__ status) {
    constexpr int K = NPL * 64;
    __shared__ uint32_t stage[MODE == M_WAVE ? 4 * 64 * (NPL + 1) : 1];   // (M_PIVOT needs no transpose)
    const int lane = threadIdx.x & 63;
    const int d = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (d >= D) return;
    const int b0 = blockIdx.y * nb;
    const int b1 = min(B, b0 + nb);

    // this lane's NPL upper bin boundaries (the last bin of lane 63 has none: C = 1)
    double e[NPL];
    const double* er = endpoints + (int64_t)d * e_stride + lane * NPL;
#pragma unroll
    for (int i = 0; i < NPL; ++i) e[i] = (lane * NPL + i < K - 1) ? er[i] : 0.0;

    const double M = (double)((1ll << bits) - (1ll << quantbits));
    const double hstep = UNI ? step[d] : 0.0;
    if (UNI) {
#pragma unroll
        for (int i = 1; i < NPL; ++i) e[i] = e[i] - fma((double)i, hstep, e[0]);   // residuals r_i; e[0] stays the anchor
    }
    // (mu, scale) of the next chain are fetched while the current one is computed: the row is
    // wave-uniform, so these are scalar loads whose latency would otherwise sit in front of every row
    PT mu_n = mu[(int64_t)b0 * D + d], sc_n = scale[(int64_t)b0 * D + d];
    int sym_n = 0;
    if (MODE == M_ENCODE) sym_n = sym[(int64_t)b0 * D + d];
    for (int b = b0; b < b1; ++b) {
        const int64_t row = (int64_t)b * D + d;
        const double m_ = (double)mu_n;
        const double rs = recip_scale((double)sc_n);
        // NaN / Inf / non-positive parameters (a broken checkpoint) would still produce a well-formed table of
        // garbage: flag the chain instead (first error sticks; later launches skip it)
        const bool okp = ((double)sc_n > 0.0) && (rs > 0.0) && (fabs(m_) < __builtin_huge_val());
        const int64_t nrow = (int64_t)min(b + 1, b1 - 1) * D + d;
        mu_n = mu[nrow];
        sc_n = scale[nrow];
        const int sym_c = sym_n;
        if (MODE == M_ENCODE) sym_n = sym[nrow];

        Bins<NPL> bn;
        const bool dom = logistic_row<NPL, UNI>(e, hstep, m_, rs, M, lane, bn);

        bool bad;
        uint32_t bumped = 0, rem = 0;
        uint32_t c = bump_and_scan<NPL>(bn, lane, bits, bad, MODE == M_PIVOT ? &bumped : nullptr, MODE == M_PIVOT ? &rem : nullptr);
        bad = bad || !dom;
        if (status && (__ballot(bad) != 0ull || !okp) && lane == 0 && status[b] == BS_ST_OK) status[b] = BS_ST_BADTABLE;

        if (MODE == M_PIVOT) {
            // BS_LAYOUT_PIVOT: the hand-off to k_rans_pop_pivot is ONE 8-byte word per lane -- the cumulative value at
            // the lane's first bin (remnant included for the lanes behind the bumped one) and, in lanes 0 and 1, which bin
            // took the remnant and how much.  The popping wavefront rebuilds the NPL bins of the one group its symbol falls
            // into with the arithmetic of logistic_row: 512 B per row cross HBM instead of 4 (K + 64).
            uint2 v;
            v.x = c;
            v.y = lane == 0 ? bumped : lane == 1 ? rem : 0u;
            *reinterpret_cast<uint2*>(out0 + row * ld + lane * 2) = v;
        } else if (MODE == M_WAVE) {
            // wave-native rows for k_rans_pop_wave: the K entries permuted as wave_offset(), then 64 pivot
            // words at [K, K+64) (see below).  The permutation is a 64 x NPL transpose:
            // it goes through a wave-private LDS tile (entry j at j + j/NPL: conflict-free writes, reads with
            // one 2-way conflict) so that the row leaves as NPL/4 fully coalesced 1-KB stores instead of
            // NPL scattered dword stores (16 cache lines each).  No barrier: one wave, and the LDS queue of
            // a wave is served in order.
            uint32_t* sw = stage + (threadIdx.x >> 6) * (64 * (NPL + 1));
            uint32_t* o = out0 + row * ld;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                sw[lane * (NPL + 1) + i] = c;
                c += bn.t[i] + 1u;
            }
            asm volatile("" ::: "memory");
            const uint32_t* sr = sw + lane + lane / NPL;
#pragma unroll
            for (int i = 0; i < NPL / 4; ++i) {
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                v4u v;
                v.x = sr[(64 + 64 / NPL) * (4 * i + 0)];
                v.y = sr[(64 + 64 / NPL) * (4 * i + 1)];
                v.z = sr[(64 + 64 / NPL) * (4 * i + 2)];
                v.w = sr[(64 + 64 / NPL) * (4 * i + 3)];
                // streaming store: the row is read once, much later, by the pop kernel -- keep it from evicting
                // the endpoint rows that the other chain groups of this XCD are about to re-read from L2
                __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(o) + i * 64 + lane);
            }
            // pivots: lane r < NPL gets c_{64r} (the first entry of register r), lane NPL gets c_K = 2^bits,
            // the rest never compare <= m
            uint32_t pv = 0xffffffffu;
            if (lane < NPL) pv = sw[(64 + 64 / NPL) * lane];
            if (lane == NPL) pv = 1u << bits;
            __builtin_nontemporal_store(pv, o + K + lane);
            asm volatile("" ::: "memory");
        } else if (MODE != M_ENCODE) {
            uint32_t* o = out0 + row * ld + lane * NPL;
            if (MODE == M_LINEAR_VEC) {
#pragma unroll
                for (int i = 0; i < NPL; i += 4) {
                    uint4 v;
                    v.x = c; c += bn.t[i] + 1u;
                    v.y = c; c += bn.t[i + 1] + 1u;
                    v.z = c; c += bn.t[i + 2] + 1u;
                    v.w = c; c += bn.t[i + 3] + 1u;
                    *reinterpret_cast<uint4*>(o + i) = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPL; ++i) { o[i] = c; c += bn.t[i] + 1u; }
            }
            if (lane == 63) out0[row * ld + K] = 1u << bits;
        } else {
            // the symbol is wave-uniform (one row per wave): its lane and bin are scalars, so (f_s, c_s)
            // come out of the registers by scalar index instead of a per-bin select
            // (fetched a row ahead, like mu and scale: a scalar load whose latency would otherwise sit at the end of every row)
            const int s = __builtin_amdgcn_readfirstlane(sym_c);
            const bool ok = (s >= 0) && (s < K);
            if (!ok && lane == 0 && status[b] == BS_ST_OK) status[b] = BS_ST_BADSYMBOL;  // first error sticks
            const int ss = ok ? s : 0;
            const int idx = ss % NPL;
            // c_s = the lane's first cumulative value + the idx bins in front of the symbol: idx is a scalar, so this is a
            // scalar branch to the one prefix that is needed (idx adds) instead of all NPL prefixes and two selects
            uint32_t fs = 0, cs = c;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                if (idx == k) {
                    uint32_t a = c;
#pragma unroll
                    for (int i = 0; i < k; ++i) a += bn.t[i] + 1u;
                    cs = a;
                    fs = bn.t[k] + 1u;
                }
            }
            if (lane == ss / NPL) {
                out0[row] = fs;
                out1[row] = cs;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_table_rows: ANS.__init__ on caller-supplied pmf rows, K = 64*NPL, one wave per row
// ------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(256) void k_table_rows(const double* __restrict__ pmf, int64_t rows, int bits,
                                                    int quantbits, uint32_t* __restrict__ f_out,
                                                    uint32_t* __restrict__ cdf_out, int64_t ld,
                                                    int32_t* __restrict__ status) {
    constexpr int K = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double M = (double)((1ll << bits) - (1ll << quantbits));
    const double* p = pmf + row * K + lane * NPL;
    Bins<NPL> bn;
#pragma unroll
    for (int i = 0; i < NPL; ++i) bn.t[i] = trunc_u32(p[i] * M);
    bool bad;
    uint32_t c = bump_and_scan<NPL>(bn, lane, bits, bad);
    if (status && __ballot(bad) != 0ull && lane == 0) status[row] = BS_ST_BADTABLE;
    uint32_t* co = cdf_out + row * ld + lane * NPL;
    uint32_t* fo = f_out ? f_out + row * K + lane * NPL : nullptr;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        co[i] = c;
        if (fo) fo[i] = bn.t[i] + 1u;
        c += bn.t[i] + 1u;
    }
    if (lane == 63) cdf_out[row * ld + K] = 1u << bits;
}

// any K >= 1: bins strided over the lanes (j = it*64 + lane), two passes over the row
__global__ __launch_bounds__(256) void k_table_rows_generic(const double* __restrict__ pmf, int64_t rows, int K,
                                                            int bits, int quantbits, uint32_t* __restrict__ f_out,
                                                            uint32_t* __restrict__ cdf_out, int64_t ld,
                                                            int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double M = (double)((1ll << bits) - (1ll << quantbits));
    const double* p = pmf + row * K;
    // pass 1: sum, maximum, first index of the maximum
    uint32_t fsum = 0, best = 0;
    int barg = 0x7fffffff;
    for (int j = lane; j < K; j += 64) {
        const uint32_t fj = (uint32_t)((int32_t)(p[j] * M) + 1);
        fsum += fj;
        if (fj > best) { best = fj; barg = j; }
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_add(fsum), 63);
    const uint32_t mx = wave_max_u32(best);
    // smallest bin index among the lanes holding the maximum
    uint32_t cand = (best == mx) ? (uint32_t)(0x7fffffff - barg) : 0u;
    const int arg = 0x7fffffff - (int)wave_max_u32(cand);
    const uint32_t rem = (1u << bits) - total;
    if (status && lane == 0 && (int32_t)(mx + rem) < 1) status[row] = BS_ST_BADTABLE;
    // pass 2: exclusive prefix, 64 bins at a time
    uint32_t carry = 0;
    for (int j0 = 0; j0 < K; j0 += 64) {
        const int j = j0 + lane;
        uint32_t fj = 0;
        if (j < K) {
            fj = (uint32_t)((int32_t)(p[j] * M) + 1);
            if (j == arg) fj += rem;
        }
        const uint32_t incl = wave_incl_scan_add(fj);
        if (j < K) {
            cdf_out[row * ld + j] = carry + incl - fj;
            if (f_out) f_out[row * K + j] = fj;
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) cdf_out[row * ld + K] = 1u << bits;
}

// ------------------------------------------------------------------------------------------
// k_rans_pop: one wavefront per chain.  NV = uint4 loads per lane per row (K = 256*NV), rows
// 16-byte aligned.  Rows are streamed PF deep into registers (the table lives in HBM: at B=100,
// Z=2048, K=1024 it is 0.84 GB, far beyond L2), the symbol is the popcount of 4*NV 64-wide ballots,
// c_s / c_{s+1} come out of the row registers by scalar-indexed VGPR read + v_readlane (no dependent
// memory access), the next two stack words wait in scalar registers, and the 64-bit head never
// leaves the scalar unit.
// ------------------------------------------------------------------------------------------
template <int NV>
struct RowRegs {
    static constexpr int K = NV * 256;
    typedef uint32_t vec_t __attribute__((ext_vector_type(4 * NV)));
    vec_t v;
    __device__ __forceinline__ void find(uint32_t m, int bits, int& s, uint32_t& cs, uint32_t& cs1) const {
        s = count_le(m) - 1;  // c_0 = 0 <= m always, so s >= 0
        cs = entry(s);
        cs1 = (s + 1 < K) ? entry(s + 1) : (1u << bits);
    }
    __device__ __forceinline__ void load(const uint32_t* row, int lane) {
        const uint4* r = reinterpret_cast<const uint4*>(row);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint4 t = r[i * 64 + lane];
            v[4 * i + 0] = t.x;
            v[4 * i + 1] = t.y;
            v[4 * i + 2] = t.z;
            v[4 * i + 3] = t.w;
        }
    }
    // entry j of the row: uint4 index q = j/4 lives in lane q%64, load i = q/64, component j%4
    __device__ __forceinline__ uint32_t entry(int j) const {
        const int e = ((j >> 8) << 2) | (j & 3);
        return (uint32_t)__builtin_amdgcn_readlane((int)v[e], (j >> 2) & 63);
    }
    __device__ __forceinline__ int count_le(uint32_t m) const {
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 4 * NV; ++e) cnt += __popcll(__ballot(v[e] <= m));
        return cnt;
    }
};

template <class ROW, int PF>
__global__ __launch_bounds__(64) void k_rans_pop(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                 int32_t* __restrict__ len, int64_t cap,
                                                 const uint32_t* __restrict__ cdf, int64_t chain_stride, int64_t ld,
                                                 int D, int bits, int32_t* __restrict__ sym_out,
                                                 const double* __restrict__ centres, int64_t c_stride,
                                                 float* __restrict__ centre_out, int32_t* __restrict__ status) {
    // D is a multiple of 64 here (host dispatch).  The main loop contains NO conditional memory
    // operation: row prefetches are unconditional (clamped addresses), stack words are fetched one
    // 64-row chunk ahead, decoded symbols go to LDS and are written out in a coalesced epilogue.
    // That keeps hipcc's s_waitcnt vmcnt(N) counted (PF-1 rows stay in flight) instead of vmcnt(0).
    extern __shared__ int32_t sh_sym[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[b] != BS_ST_OK) {  // failed chain: skipped, but its outputs stay well-defined
        for (int dd = lane; dd < D; dd += 64) {
            sym_out[(int64_t)b * D + dd] = 0;
            if (centres) centre_out[(int64_t)b * D + dd] = 0.0f;
        }
        return;
    }
    // latency-critical serial wave: win instruction-issue arbitration against co-resident bulk
    // kernels (the convs of another chain group run concurrently on other streams)
    __builtin_amdgcn_s_setprio(BS_SERIAL_PRIO);
    uint64_t h = head[b];
    int n = len[b];
    const uint32_t* stk = stack + (int64_t)b * cap;
    const uint32_t* tab = cdf + (int64_t)b * chain_stride;
    const uint64_t mask = (1ull << bits) - 1;
    int st = BS_ST_OK;

    auto stack_window = [&](int top, int off) -> uint32_t {  // lane l <- stk[top-1-off-l] (0 if below the stack)
        const int i = top - 1 - off - lane;
        return stk[max(i, 0)];
    };
    // words this chunk may consume (at most 64): loaded against `wtop`, the word count at load time
    int wtop = n;
    uint32_t wa = stack_window(wtop, 0), wb = stack_window(wtop, 64);
    // materialise the first window now (one exposed latency per launch): otherwise its loads count as
    // 'possibly still in flight' at every window read of the main loop and turn the counted waits into ~vmcnt(0)
    asm volatile("" : "+v"(wa), "+v"(wb));

    ROW buf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        buf[u].load(tab + (int64_t)max(D - 1 - u, 0) * ld, lane);
        // keep issue order == consumption order: the counted vmcnt of the main loop must also be valid
        // on the first trip, when these loads (not the in-loop refills) are the ones in flight
        __builtin_amdgcn_sched_barrier(0);
    }

    for (int c64 = D / 64 - 1; c64 >= 0; --c64) {
        // fetch the window the NEXT chunk will read; it has a whole chunk to arrive
        const int ntop = n;
        const uint32_t na = stack_window(ntop, 0), nb = stack_window(ntop, 64);
        int mysym = 0;
        for (int g = 64 / PF - 1; g >= 0; --g) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int d = c64 * 64 + g * PF + (PF - 1 - u);
                const uint32_t m = (uint32_t)(h & mask);
                int s;
                uint32_t cs, cs1;
                buf[u].find(m, bits, s, cs, cs1);
                // this row's registers are free again: fetch the row PF steps ahead (clamped, unconditional)
                buf[u].load(tab + (int64_t)max(d - PF, 0) * ld, lane);
                const uint64_t f = (uint64_t)(cs1 - cs);
                h = f * (h >> bits) + (uint64_t)(m - cs);
                if (h < (1ull << 32)) {
                    if (n <= 0) {
                        st = BS_ST_UNDERFLOW;  // keep going on garbage (reads stay in bounds); reported below
                    } else {
                        const int o = wtop - n;  // 0..127 within this chunk's window
                        const uint32_t w = (o < 64) ? (uint32_t)__builtin_amdgcn_readlane((int)wa, o & 63)
                                                    : (uint32_t)__builtin_amdgcn_readlane((int)wb, o & 63);
                        h = (h << 32) | (uint64_t)w;
                        --n;
                    }
                }
                mysym = (lane == (d & 63)) ? s : mysym;
            }
        }
        sh_sym[c64 * 64 + lane] = mysym;
        wtop = ntop;
        wa = na;
        wb = nb;
    }
    if (lane == 0) {
        head[b] = h;
        len[b] = n;
        if (st != BS_ST_OK) status[b] = st;
    }
    __syncthreads();
    for (int dd = lane; dd < D; dd += 64) {
        const int sy = sh_sym[dd];
        const int64_t o = (int64_t)b * D + dd;
        sym_out[o] = sy;
        if (centres) centre_out[o] = (float)centres[(int64_t)dd * c_stride + sy];
    }
}

// ------------------------------------------------------------------------------------------
// k_rans_pop_wave: BS_LAYOUT_WAVE rows, one wavefront per chain.
//
// A lone wavefront issues about one instruction every 3-4 ns whatever it is (tools/probes/instr_latency.hip),
// so the step is written for instruction count.  A row is NR = K/64 registers (register r, lane l =
// c_{64r+l}) plus one pivot register (lane r = c_{64r}, lane NR = 2^bits, other lanes 0xffffffff):
//   ballot(pivot <= m)      -> which register holds the symbol (scalar-indexed VGPR read)
//   ballot(R[r] <= m)       -> its lane; the entries are strictly increasing, so the popcount IS the
//                              position, no shifting or masking of the ballot
//   c_s, c_{s+1}            -> two v_readlane of that same register (the pivot of the next register when
//                              the symbol sits in lane 63)
// Rows arrive through buffer loads whose only per-row address arithmetic is one scalar subtract; PF rows
// stay in flight (counted vmcnt).  Stack words for a 64-symbol chunk wait in ONE register (lane k = the k-th
// word the chunk will consume), realigned once per chunk with ds_bpermute, so a renormalisation is one
// v_readlane.  The 64-bit head never leaves the scalar unit; decoded symbols go to a lane of a register
// (one select per symbol), to LDS once per chunk, and to HBM with the centre gather in a coalesced epilogue.
// ------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NR>
struct WaveRow2 {
    typedef uint32_t vec_t __attribute__((ext_vector_type(NR)));
    vec_t R;
    uint32_t pivot;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, uint32_t voff_row, uint32_t voff_piv, uint32_t soff) {
#pragma unroll
        for (int i = 0; i < NR / 4; ++i) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_row + i * 1024, soff, 2);  // nt: read once
            R[4 * i + 0] = t.x;
            R[4 * i + 1] = t.y;
            R[4 * i + 2] = t.z;
            R[4 * i + 3] = t.w;
        }
        pivot = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_piv, soff, 2);
    }
};

template <int NR, int PF>
__global__ __launch_bounds__(64) void k_rans_pop_wave(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                      int32_t* __restrict__ len, int64_t cap,
                                                      const uint32_t* __restrict__ cdf, int64_t chain_stride,
                                                      int64_t ld, int D, int bits, int32_t* __restrict__ sym_out,
                                                      const double* __restrict__ centres, int64_t c_stride,
                                                      float* __restrict__ centre_out, int32_t* __restrict__ status) {
    // host dispatch guarantees: D % 64 == 0, rows 16-byte aligned, D * ld * 4 < 2^31
    constexpr int K = NR * 64;
    extern __shared__ int32_t sh_sym[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[b] != BS_ST_OK) {  // failed chain: skipped, but its outputs stay well-defined
        for (int dd = lane; dd < D; dd += 64) {
            sym_out[(int64_t)b * D + dd] = 0;
            if (centres) centre_out[(int64_t)b * D + dd] = 0.0f;
        }
        return;
    }
    // latency-critical serial wave: win instruction-issue arbitration against co-resident bulk kernels
    __builtin_amdgcn_s_setprio(BS_SERIAL_PRIO);
    uint64_t h = head[b];
    int n = len[b];
    const uint32_t* stk = stack + (int64_t)b * cap;
    const uint32_t ld4 = (uint32_t)ld * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(cdf + (int64_t)b * chain_stride), 0, (int)((uint32_t)D * ld4), 0x00020000);
    const uint32_t voff_row = (uint32_t)lane * 16u, voff_piv = (uint32_t)K * 4u + (uint32_t)lane * 4u;
    const uint32_t mask = (1u << bits) - 1u;
    int st = BS_ST_OK;

    auto stack_window = [&](int top, int off) -> uint32_t {  // lane l <- stk[top-1-off-l] (clamped at the bottom)
        const int i = top - 1 - off - lane;
        return stk[max(i, 0)];
    };
    // the 128 words below `wtop`: whatever the previous chunk consumed (<= 64), the next 64 are in here
    int wtop = n;
    uint32_t wa = stack_window(wtop, 0), wb = stack_window(wtop, 64);
    asm volatile("" : "+v"(wa), "+v"(wb));  // see k_rans_pop: keep these loads out of the counted waits

    WaveRow2<NR> buf[PF];
    uint32_t soff = (uint32_t)(D - 1) * ld4;  // byte offset of the row the NEXT refill fetches
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        buf[u].load(rs, voff_row, voff_piv, soff);
        soff = (uint32_t)max((int)(soff - ld4), 0);  // clamped: the last PF refills re-read row 0, unused
        __builtin_amdgcn_sched_barrier(0);           // issue order == consumption order (counted vmcnt)
    }

    for (int c64 = D / 64 - 1; c64 >= 0; --c64) {
        // this chunk's words: realign the 128-word window by what the previous chunk consumed
        const int idx = (wtop - n) + lane;  // 0..127
        const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute((idx & 63) << 2, (int)wa);
        const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute((idx & 63) << 2, (int)wb);
        const uint32_t win = idx < 64 ? pa : pb;
        // and fetch the window the NEXT chunk will realign; it has a whole chunk to arrive
        const int ntop = n;
        const uint32_t na = stack_window(ntop, 0), nb = stack_window(ntop, 64);
        int o = 0;  // words consumed in this chunk
        // the chunk's symbols: lane i of (symr, symp) = (register, position) of symbol 64 c64 + i.  Both are scalars the
        // search has in hand, the lane is a compile-time constant of the unrolled chunk: two v_writelane per symbol (round 3
        // spent eight instructions per symbol on `(lane == d % 64) ? 64 r + p : mysym`)
        uint32_t symr = 0, symp = 0;
#pragma unroll
        for (int g = 64 / PF - 1; g >= 0; --g) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const uint32_t m = (uint32_t)h & mask;
                const int r1 = __popcll(__ballot(buf[u].pivot <= m));  // 1..NR (c_0 = 0 <= m)
                const uint32_t x = buf[u].R[r1 - 1];
                const int pos = __popcll(__ballot(x <= m));            // 1..64
                const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)x, pos - 1);
                const uint32_t cin = (uint32_t)__builtin_amdgcn_readlane((int)x, pos & 63);
                const uint32_t cnx = (uint32_t)__builtin_amdgcn_readlane((int)buf[u].pivot, r1);
                const uint32_t f = (pos == 64 ? cnx : cin) - cs;
                asm("v_writelane_b32 %0, %1, %2" : "+v"(symr) : "s"(r1 - 1), "n"(g * PF + (PF - 1 - u)));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(symp) : "s"(pos - 1), "n"(g * PF + (PF - 1 - u)));
                // this row's registers are free again: fetch the row PF steps ahead
                buf[u].load(rs, voff_row, voff_piv, soff);
                soff = (uint32_t)max((int)(soff - ld4), 0);
                h = (uint64_t)f * (h >> bits) + (uint64_t)(m - cs);
                uint32_t hhi = (uint32_t)(h >> 32);
#if !__has_feature(address_sanitizer)   // (the ASan build, bitswap_amd/build.py --asan, keeps the head in vector registers)
                asm("" : "+s"(hhi));  // keep this a 32-bit scalar compare (hipcc otherwise builds a 64-bit VALU one)
#endif
                if (hhi == 0u) {  // h < 2^32, mnist_compress.py:65
                    h = (h << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)win, o);
                    ++o;
                }
            }
        }
        sh_sym[c64 * 64 + lane] = (int32_t)(symr * 64u + symp);
        n -= o;
        if (n < 0) {  // popped below the bottom: garbage from here on (reads stay in bounds), reported below
            st = BS_ST_UNDERFLOW;
            n = 0;
        }
        wtop = ntop;
        wa = na;
        wb = nb;
    }
    if (lane == 0) {
        head[b] = h;
        len[b] = n;
        if (st != BS_ST_OK) status[b] = st;
    }
    __syncthreads();
    for (int dd = lane; dd < D; dd += 64) {
        const int sy = sh_sym[dd];
        const int64_t oo = (int64_t)b * D + dd;
        sym_out[oo] = sy;
        if (centres) centre_out[oo] = (float)centres[(int64_t)dd * c_stride + sy];
    }
}

// ------------------------------------------------------------------------------------------
// k_rans_pop_pivot: BS_LAYOUT_PIVOT rows (uniform-width bins, CDF spec 2), one wavefront per chain.
//
// The table kernel hands over 64 cumulative values per row (one per group of NPL bins) plus which bin took the remnant
// and how much.  Per symbol: ballot(pivot <= m) names the group L; lanes 0 .. NPL-1 rebuild the cdf of its NPL bins and
// lane NPL the last cdf of group L-1, each with exactly the operations logistic_row spends on that bin (own anchor
// exponential, own geometric factor, residual of the stored endpoint, correctly rounded reciprocal) -- the truncated
// differences are therefore the table's, and a 6-step scan on top of the pivot gives c_s and f_s.  About 2.5x the
// instructions of k_rans_pop_wave per symbol, but 512 B of HBM traffic per row instead of 4352 B: at 400 chains the
// row-reading pop kernel ran at the HBM roof (3.57 GB per launch in 0.57 ms) and nothing overlapped with it
// (profiles/r03m_overlap2.txt); this one touches the L2-resident endpoint table and little else.
// Endpoints of the group are fetched AFTER the group is known (data dependent) and consumed after the two exponentials
// that do not need them; pivots and the 64 anchor endpoints of a row are prefetched PF rows ahead like the rows of
// k_rans_pop_wave; (mu, scale, bin width) wait in registers per 64-symbol chunk.
// ------------------------------------------------------------------------------------------
// lane i <- lane i-1 of the whole wavefront (DPP wave_shr:1; lane 0 keeps its value)
__device__ __forceinline__ double wave_shr1_f64(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)u, (int)(uint32_t)u, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(u >> 32), (int)(uint32_t)(u >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

template <int NPL, typename PT, int PF>
__global__ __launch_bounds__(64) void k_rans_pop_pivot(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                       int32_t* __restrict__ len, int64_t cap,
                                                       const uint32_t* __restrict__ piv, int64_t ld,
                                                       const double* __restrict__ endpoints, int64_t e_stride,
                                                       const double* __restrict__ step, const PT* __restrict__ mu,
                                                       const PT* __restrict__ scale, int D, int bits, int quantbits,
                                                       int32_t* __restrict__ sym_out, const double* __restrict__ centres,
                                                       int64_t c_stride, float* __restrict__ centre_out,
                                                       int32_t* __restrict__ status) {
    constexpr int K = NPL * 64;
    constexpr bool ONE_EXP = NPL <= 16;      // both exponentials of a symbol in ONE instruction stream (lower / upper half-wave)
    extern __shared__ int32_t sh_sym[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[b] != BS_ST_OK) {  // failed chain (a bad table among them): skipped, outputs well-defined
        for (int dd = lane; dd < D; dd += 64) {
            sym_out[(int64_t)b * D + dd] = 0;
            if (centres) centre_out[(int64_t)b * D + dd] = 0.0f;
        }
        return;
    }
    __builtin_amdgcn_s_setprio(BS_SERIAL_PRIO);
    uint64_t h = head[b];
    int n = len[b];
    const uint32_t* stk = stack + (int64_t)b * cap;
    const uint32_t mask = (1u << bits) - 1u;
    const double M = (double)((1ll << bits) - (1ll << quantbits));
    int st = BS_ST_OK;
    const int64_t ld2 = ld / 2;
    // this lane's role in the rebuild.  Lanes 0 .. NPL-1: bin `bi` of the symbol's group; lane NPL: the last bin of the group
    // below it.  With ONE_EXP the upper half-wave evaluates the geometric factors Q_b = exp(-b h/scale) in the same
    // instructions in which the lower half evaluates the anchors exp(-t_a); lane 32 + k serves lane k.
    const bool is_bin = lane < NPL;
    const int role = ONE_EXP ? (lane & 31) : lane;
    const int bi = role < NPL ? role : NPL - 1;
    const bool q_lane = ONE_EXP && lane >= 32;

    auto stack_window = [&](int top, int off) -> uint32_t {
        const int i = top - 1 - off - lane;
        return stk[max(i, 0)];
    };
    int wtop = n;
    uint32_t wa = stack_window(wtop, 0), wb = stack_window(wtop, 64);

    uint2 pv[PF];
    double anc[PF];
    const uint2* pp = reinterpret_cast<const uint2*>(piv + (int64_t)b * D * ld) + (int64_t)(D - 1) * ld2 + lane;   // row of the next refill
    const double* ap = endpoints + (int64_t)(D - 1) * e_stride + lane * NPL;
    int dl = D - 1;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        pv[u] = *pp;
        anc[u] = *ap;
        if (dl > 0) { pp -= ld2; ap -= e_stride; }
        --dl;
    }
    const double* erow = endpoints + (int64_t)(D - 1) * e_stride;   // endpoint row of the symbol being popped

    int d = D - 1;
    for (int c64 = D / 64 - 1; c64 >= 0; --c64) {
        const int idx = (wtop - n) + lane;  // 0..127
        const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute((idx & 63) << 2, (int)wa);
        const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute((idx & 63) << 2, (int)wb);
        const uint32_t win = idx < 64 ? pa : pb;
        const int ntop = n;
        const uint32_t na = stack_window(ntop, 0), nb = stack_window(ntop, 64);
        // parameters of this chunk's 64 rows: lane k <- row c64*64 + k
        const int64_t prm = (int64_t)b * D + c64 * 64 + lane;
        const double mu_l = (double)mu[prm], h_l = step[c64 * 64 + lane];
        const double rs_l = recip_scale((double)scale[prm]);
        const double hr_l = h_l * rs_l;
        int o = 0;
        uint32_t mysym = 0;
        for (int g = 64 / PF - 1; g >= 0; --g) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int dk = d & 63;
                const uint32_t m = (uint32_t)h & mask;
                const int L = __popcll(__ballot(pv[u].x <= m)) - 1;          // group of the symbol: 0..63 (c_0 = 0 <= m)
                const int Lb = max(L - 1, 0);
                const int j = (is_bin ? L : Lb) * NPL + bi;                  // this lane's bin (lanes 0 .. NPL)
                // its upper endpoint: data dependent, requested first, used last
                const double e_j = erow[min(j, K - 2)];
                erow -= d > 0 ? e_stride : 0;
                const double m_ = readlane_f64(mu_l, dk), rs = readlane_f64(rs_l, dk), hstep = readlane_f64(h_l, dk);
                const double hr = readlane_f64(hr_l, dk);
                const double eL = readlane_f64(anc[u], L), eLb = readlane_f64(anc[u], Lb);
                const double e_a = is_bin ? eL : eLb;
                const uint32_t piv_L = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].x, L);
                const uint32_t bumped = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].y, 0);
                const uint32_t rem = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].y, 1);
                // refill: the row PF steps ahead
                pv[u] = *pp;
                anc[u] = *ap;
                if (dl > 0) { pp -= ld2; ap -= e_stride; }
                --dl;
                // logistic_row, one bin per lane
                double A, Q;
                if (ONE_EXP) {
                    const double x = det_exp(q_lane ? -((double)bi * hr) : -((e_a - m_) * rs));
                    A = x;
                    Q = __shfl(x, lane | 32, 64);                            // lane k < 32 <- lane 32 + k
                } else {
                    A = det_exp(-((e_a - m_) * rs));
                    Q = det_exp(-((double)bi * hr));
                }
                const double r = e_j - fma((double)bi, hstep, e_a);
                const double eps = r * rs;
                const double uu = fma(-A, eps, A);
                double c = recip_1_to_huge(fma(Q, uu, 1.0));
                if (j == K - 1) c = 1.0;                                     // the last bin has no upper endpoint
                // cdf of the bin below: lane-1 within the group, lane NPL for bin 0, nothing for the very first bin
                double below = wave_shr1_f64(c);
                const double c_grp_below = readlane_f64(c, NPL);
                if (lane == 0) below = L == 0 ? 0.0 : c_grp_below;
                uint32_t f = trunc_u32((c - below) * M) + 1u;
                if ((uint32_t)j == bumped) f += rem;
                if (!is_bin) f = 0u;
                uint32_t incl = f;                                           // inclusive scan over the NPL bins
                incl += dpp_or0<0x111, 0xf>(incl);
                incl += dpp_or0<0x112, 0xf>(incl);
                if (NPL > 4) incl += dpp_or0<0x114, 0xf>(incl);
                if (NPL > 8) incl += dpp_or0<0x118, 0xf>(incl);
                if (NPL > 16) incl += dpp_or0<0x142, 0xa>(incl);
                const uint32_t cst = piv_L + incl - f;                       // c of this lane's bin
                const int pos = __popcll(__ballot(is_bin && cst <= m));      // 1..NPL
                const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)cst, pos - 1);
                const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)f, pos - 1);
                mysym = (lane == dk) ? (uint32_t)(L * NPL + pos - 1) : mysym;
                h = (uint64_t)fs * (h >> bits) + (uint64_t)(m - cs);
                if ((uint32_t)(h >> 32) == 0u) {  // h < 2^32, mnist_compress.py:65
                    h = (h << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)win, o);
                    ++o;
                }
                --d;
            }
        }
        sh_sym[c64 * 64 + lane] = (int32_t)mysym;
        n -= o;
        if (n < 0) {
            st = BS_ST_UNDERFLOW;
            n = 0;
        }
        wtop = ntop;
        wa = na;
        wb = nb;
    }
    if (lane == 0) {
        head[b] = h;
        len[b] = n;
        if (st != BS_ST_OK) status[b] = st;
    }
    __syncthreads();
    for (int dd = lane; dd < D; dd += 64) {
        const int sy = sh_sym[dd];
        const int64_t oo = (int64_t)b * D + dd;
        sym_out[oo] = sy;
        if (centres) centre_out[oo] = (float)centres[(int64_t)dd * c_stride + sy];
    }
}

// any K / any alignment (reference layout ld = K+1): scalar strided loads, no prefetch
__global__ __launch_bounds__(64) void k_rans_pop_generic(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                         int32_t* __restrict__ len, int64_t cap,
                                                         const uint32_t* __restrict__ cdf, int64_t chain_stride,
                                                         int64_t ld, int D, int K, int bits,
                                                         int32_t* __restrict__ sym_out,
                                                         const double* __restrict__ centres, int64_t c_stride,
                                                         float* __restrict__ centre_out,
                                                         int32_t* __restrict__ status) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[b] != BS_ST_OK) {
        for (int dd = lane; dd < D; dd += 64) {
            sym_out[(int64_t)b * D + dd] = 0;
            if (centres) centre_out[(int64_t)b * D + dd] = 0.0f;
        }
        return;
    }
    uint64_t h = head[b];
    int n = len[b];
    const uint32_t* stk = stack + (int64_t)b * cap;
    const uint32_t* tab = cdf + (int64_t)b * chain_stride;
    const uint64_t mask = (1ull << bits) - 1;
    int st = BS_ST_OK;
    for (int d = D - 1; d >= 0; --d) {
        const uint32_t* row = tab + (int64_t)d * ld;
        const uint32_t m = (uint32_t)(h & mask);
        int cnt = 0;
        for (int j0 = 0; j0 < K; j0 += 64) {
            const int j = j0 + lane;
            const bool le = (j < K) && (row[j] <= m);
            cnt += __popcll(__ballot(le));
        }
        const int s = cnt - 1;
        const uint32_t cs = row[s];
        const uint32_t cs1 = row[s + 1];
        const uint64_t f = (uint64_t)(cs1 - cs);
        h = f * (h >> bits) + (uint64_t)(m - cs);
        if (h < (1ull << 32)) {
            if (n <= 0) { st = BS_ST_UNDERFLOW; break; }
            h = (h << 32) | (uint64_t)stk[--n];
        }
        if (lane == 0) {
            const int64_t o = (int64_t)b * D + d;
            sym_out[o] = s;
            if (centres) centre_out[o] = (float)centres[(int64_t)d * c_stride + s];
        }
    }
    if (lane == 0) {
        head[b] = h;
        len[b] = n;
        if (st != BS_ST_OK) status[b] = st;
    }
}

// ------------------------------------------------------------------------------------------
// k_rans_push: one wavefront per chain.  Per 64-symbol chunk the lanes fetch (f, c) of 64 symbols
// with one coalesced load each (or gather them from cdf rows) and compute the 64 reciprocals 1/f
// lane-parallel, one chunk ahead of use; the serial part then runs wave-uniform: the 64-bit head and
// all integer work sit on the scalar unit, only the float64 quotient estimate touches the VALU.
// Emitted words collect in a register window and leave as coalesced 256-byte stores.
// ------------------------------------------------------------------------------------------
struct FcSource {  // (f, c) arrays produced by k_logistic<M_ENCODE>
    const uint32_t* f;
    const uint32_t* c;
    __device__ __forceinline__ bool fetch(int d, int D, uint32_t& fv, uint32_t& cv) const {
        fv = 1u;
        cv = 0u;
        if (d < D) { fv = f[d]; cv = c[d]; }
        return true;
    }
};

struct TableSource {  // cdf rows + symbols (drop-in ANS.encode, shared prior table)
    const uint32_t* tab;
    const int32_t* sym;
    int64_t ld;
    int layout, K, bits;
    __device__ __forceinline__ bool fetch(int d, int D, uint32_t& fv, uint32_t& cv, int& err) const {
        fv = 1u;
        cv = 0u;
        if (d >= D) return true;
        const int s = sym[d];
        if (s < 0 || s >= K) { err = BS_ST_BADSYMBOL; return false; }
        const uint32_t* row = tab + (int64_t)d * ld;
        uint32_t c0, c1;
        if (layout == BS_LAYOUT_WAVE) {
            c0 = row[wave_offset(s)];
            c1 = (s + 1 < K) ? row[wave_offset(s + 1)] : (1u << bits);
        } else {
            c0 = row[s];
            c1 = row[s + 1];
        }
        if (c1 <= c0) { err = BS_ST_BADTABLE; return false; }
        fv = c1 - c0;
        cv = c0;
        return true;
    }
};


// serial part shared by both sources.  h / f through a float64 reciprocal: after the renormalisation
// h < 2^(64-bits) * f, so q = h / f < 2^33; RN(h) * RN(1/f) is within 3 ulp of h / f (< 3e-6 absolute),
// hence trunc() is q-1, q or q+1 and one remainder check repairs it.
template <bool TABLE>
__device__ __forceinline__ void push_chain(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                           int32_t* __restrict__ len, int64_t cap, const FcSource& fc,
                                           const TableSource& ts, int D, int bits, int32_t* __restrict__ status) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[b] != BS_ST_OK) return;
    __builtin_amdgcn_s_setprio(BS_SERIAL_PRIO);
    uint64_t h = head[b];
    int n = len[b];
    uint32_t* stk = stack + (int64_t)b * cap;
    int st = BS_ST_OK;
    uint32_t wbuf = 0;  // pending output words, word k in lane k
    int wpos = 0;

    uint32_t fv, cv;
    int err = BS_ST_OK;
    if (TABLE) ts.fetch(lane, D, fv, cv, err); else fc.fetch(lane, D, fv, cv);
    const int nchunks = (D + 63) >> 6;
    for (int ck = 0; ck < nchunks; ++ck) {
        uint32_t fn, cn;
        int errn = BS_ST_OK;
        if (TABLE) ts.fetch((ck + 1) * 64 + lane, D, fn, cn, errn); else fc.fetch((ck + 1) * 64 + lane, D, fn, cn);
        if (TABLE) {
            const unsigned long long bad = __ballot(err != BS_ST_OK);
            if (bad) {  // first offending symbol of the chunk decides; nothing of this chunk is coded
                st = __builtin_amdgcn_readlane(err, __ffsll((long long)bad) - 1);
                break;
            }
        }
        const double rf = 1.0 / (double)fv;  // 64 reciprocals at once
        const int cnt = min(64, D - ck * 64);
        for (int i = 0; i < cnt; ++i) {
            const uint64_t f = (uint32_t)__builtin_amdgcn_readlane((int)fv, i);
            const uint64_t c = (uint32_t)__builtin_amdgcn_readlane((int)cv, i);
            const double rfi = readlane_f64(rf, i);
            if ((h >> (64 - bits)) >= f) {  // h >= ((2^32 >> bits) << 32) * f, mnist_compress.py:52
                wbuf = (lane == wpos) ? (uint32_t)h : wbuf;
                h >>= 32;
                if (++wpos == 64) {
                    if (n + 64 > cap) { st = BS_ST_OVERFLOW; break; }
                    stk[n + lane] = wbuf;
                    n += 64;
                    wpos = 0;
                }
            }
            uint64_t q = (uint64_t)((double)h * rfi);
            int64_t r = (int64_t)(h - q * f);
            if (r < 0) { --q; r += (int64_t)f; }
            else if (r >= (int64_t)f) { ++q; r -= (int64_t)f; }
            h = (q << bits) + (uint64_t)r + c;
        }
        if (st != BS_ST_OK) break;
        fv = fn;
        cv = cn;
        err = errn;
    }
    if (st == BS_ST_OK && wpos > 0) {
        if (n + wpos > cap) st = BS_ST_OVERFLOW;
        else {
            if (lane < wpos) stk[n + lane] = wbuf;
            n += wpos;
        }
    }
    if (lane == 0) {
        if (st == BS_ST_OK) {
            head[b] = h;
            len[b] = n;
        } else {
            status[b] = st;
        }
    }
}

// value of lane-1 (DPP wave_shr:1); lane 0 keeps `keep`
__device__ __forceinline__ uint32_t from_lane_below(uint32_t keep, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x138, 0xf, 0xf, false);
}

// Serial part for bits >= 28 (the reference uses 31), written as a 64-lane SYSTOLIC array.
//
// A lone wavefront issues one instruction every ~3.7 ns whatever the instruction is (tools/
// instr_latency.hip), so the time per symbol is the number of instructions in the step and nothing else.
// Lane i owns symbol i of a 64-symbol chunk: its (f, c, 1/f) never leave the lane.  Every step all lanes
// apply their own symbol to the head held by the lane below (one DPP shift of the 64-bit head per step);
// lane 0's input is the chunk's input and never changes, so lane i's output is final from step i on and
// after 64 steps every lane holds the exact head after its symbol.  No v_readlane, no scalar unit, no
// branch in the step: 24 VALU instructions against ~45 for the broadcast formulation.
//
// Division.  After the renormalisation h < f * 2^(64-bits), so q = h / f < 2^36.  With
// rfb = RN(RN(1/f) * (1 - 2^-40)), hd = RN(h) and the single-rounding z = fma(hd, rfb, 2^52 - 0.5), the
// integer in z's low mantissa bits is q_est = RN(x - 0.5) for an x with h/f - 2^-3 < x < h/f (three
// roundings of relative size 2^-53 against a relative bias of 2^-40), hence q_est is floor(h/f) or
// floor(h/f) - 1, never above: r_est = h - q_est * f lies in [0, 2f) and only needs the low 32 bits of the
// product.  One compare repairs it.  Emitted words (the head's low half before a renormalising step) are
// compacted by ballot rank and leave as one store per chunk.
template <bool TABLE>
__device__ __forceinline__ void push_chain_fast(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                int32_t* __restrict__ len, int64_t cap, const FcSource& fc,
                                                const TableSource& ts, int D, int bits,
                                                int32_t* __restrict__ status) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (status[b] != BS_ST_OK) return;
    __builtin_amdgcn_s_setprio(BS_SERIAL_PRIO);
    const uint64_t h0 = head[b];
    uint32_t in_lo = (uint32_t)h0, in_hi = (uint32_t)(h0 >> 32);  // this lane's input head (lane 0: the chunk's)
    int n = len[b];
    uint32_t* stk = stack + (int64_t)b * cap;
    int st = BS_ST_OK;
    const int sh = 32 - bits;  // (h >> (64 - bits)) >= f  <=>  (hi >> sh) >= f, mnist_compress.py:52

    uint32_t f, c;
    int err = BS_ST_OK;
    if (TABLE) ts.fetch(lane, D, f, c, err); else fc.fetch(lane, D, f, c);
    const int nchunks = (D + 63) >> 6;
    for (int ck = 0; ck < nchunks; ++ck) {
        uint32_t fn, cn;
        int errn = BS_ST_OK;
        if (TABLE) ts.fetch((ck + 1) * 64 + lane, D, fn, cn, errn); else fc.fetch((ck + 1) * 64 + lane, D, fn, cn);
        if (TABLE) {
            const unsigned long long bad = __ballot(err != BS_ST_OK);
            if (bad) {  // first offending symbol of the chunk decides; nothing of this chunk is coded
                st = __builtin_amdgcn_readlane(err, __ffsll((long long)bad) - 1);
                break;
            }
        }
        const double rfb = recip_1_to_huge((double)f) * (1.0 - 0x1p-40);
        const uint32_t nf = (1u << bits) - f;  // a repaired quotient adds 2^bits - f to the low word
        const int cnt = min(64, D - ck * 64);
        uint32_t out_lo = 0, out_hi = 0;
        for (int t = 0; t < cnt; ++t) {
            if (t) {  // lane i's input <- lane i-1's output of the previous step
                in_lo = from_lane_below(in_lo, out_lo);
                in_hi = from_lane_below(in_hi, out_hi);
            }
            const bool ren = (in_hi >> sh) >= f;
            const uint32_t a_lo = ren ? in_hi : in_lo;
            const uint32_t a_hi = ren ? 0u : in_hi;
            const double hd = fma((double)a_hi, 0x1p32, (double)a_lo);  // RN(h), one rounding
            double z;  // = fma(hd, rfb, 2^52 - 0.5): three-operand form, the constant stays in scalar registers
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(z) : "v"(hd), "v"(rfb), "s"(0x1p52 - 0.5));
            const uint64_t zb = (uint64_t)__double_as_longlong(z);
            const uint32_t q_lo = (uint32_t)zb, q_hi = (uint32_t)(zb >> 32);  // q_hi: 0x43300000 | (q_est >> 32)
            const uint32_t r = a_lo - q_lo * f;  // r_est, exact in [0, 2f): the low 32 bits are all of it
            // head' = ((q_est + ge) << bits) + (r - ge * f) + c = (q_est << bits) + [r + c + ge * (2^bits - f)]
            const uint32_t w = r + c + ((r >= f) ? nf : 0u);  // < 2^(bits+1): may carry into the high word
            const uint32_t l = q_lo << bits;
            out_lo = l + w;
            // (q_est >> sh): alignbit only looks at the low `sh` bits of q_hi, the exponent bits fall out
            out_hi = __builtin_amdgcn_alignbit(q_hi, q_lo, (uint32_t)sh) + (out_lo < l ? 1u : 0u);
        }
        const bool ren = (in_hi >> sh) >= f;  // of the final inputs
        // words: lane i emitted the low half of its input iff it renormalised
        const unsigned long long emit = __ballot(ren && lane < cnt);
        const int nw = __popcll(emit);
        if (nw) {
            if ((int64_t)n + nw > cap) { st = BS_ST_OVERFLOW; break; }
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(emit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)emit, 0u));
            if (ren && lane < cnt) stk[n + rank] = in_lo;
            n += nw;
        }
        // the chunk's output is the next chunk's lane-0 input
        in_lo = (uint32_t)__builtin_amdgcn_readlane((int)out_lo, cnt - 1);
        in_hi = (uint32_t)__builtin_amdgcn_readlane((int)out_hi, cnt - 1);
        f = fn;
        c = cn;
        err = errn;
    }
    if (lane == 0) {
        if (st == BS_ST_OK) {
            head[b] = ((uint64_t)in_hi << 32) | in_lo;
            len[b] = n;
        } else {
            status[b] = st;
        }
    }
}

__global__ __launch_bounds__(64) void k_rans_push(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                  int32_t* __restrict__ len, int64_t cap,
                                                  const uint32_t* __restrict__ fs, const uint32_t* __restrict__ cs,
                                                  int B, int D, int bits, int32_t* __restrict__ status) {
    const FcSource fc{fs + (int64_t)blockIdx.x * D, cs + (int64_t)blockIdx.x * D};
    const TableSource none{nullptr, nullptr, 0, 0, 0, 0};
    if (bits >= 28) push_chain_fast<false>(head, stack, len, cap, fc, none, D, bits, status);
    else push_chain<false>(head, stack, len, cap, fc, none, D, bits, status);
}

__global__ __launch_bounds__(64) void k_rans_push_table(uint64_t* __restrict__ head, uint32_t* __restrict__ stack,
                                                        int32_t* __restrict__ len, int64_t cap,
                                                        const uint32_t* __restrict__ cdf, int64_t chain_stride,
                                                        int64_t ld, int layout, const int32_t* __restrict__ sym, int B,
                                                        int D, int K, int bits, int32_t* __restrict__ status) {
    const FcSource none{nullptr, nullptr};
    const TableSource ts{cdf + (int64_t)blockIdx.x * chain_stride, sym + (int64_t)blockIdx.x * D, ld, layout, K, bits};
    if (bits >= 28) push_chain_fast<true>(head, stack, len, cap, none, ts, D, bits, status);
    else push_chain<true>(head, stack, len, cap, none, ts, D, bits, status);
}


// ------------------------------------------------------------------------------------------
// BS_FORMAT_WAVE64 -- the opt-in 64-state stream format (no reference counterpart).
//
// A chain owns 64 independent rANS states (head, stack, length); symbol d of every coding operation goes to
// state d % 64, in the reference's order within that state (pushes ascending d, pops descending d,
// mnist_compress.py:50,60).  That removes the one serial dependence of the reference format -- a single 64-bit
// head per chain -- and with it the serial kernels and the cdf-row hand-off through HBM: ONE wavefront owns
// (residue j, a few chains), builds the integer table row of dim d = 64 i + j in its registers exactly as
// k_logistic does, and codes the symbol right there: the row never exists outside the register file.
//   pop : m = head & mask;  lane L = last lane whose first entry is <= m (ballot);  bin inside the lane by a
//         log2(NPL)-step binary search on scalar-indexed register reads;  head update on the scalar unit.
//   push: (f, c) of the given symbol by scalar lane / register index;  head / f through the float64 reciprocal
//         with one correction, as k_rans_push's generic path.
// The price: 64 flushes (heads) and 64 x the initial bits per chain instead of one -- 64 x 64 bits per chain,
// 0.02 bits/dim over 100 blocks.  oracle/backend.py::Oracle64Backend restates the format on the oracle's
// single-state primitives.
// ------------------------------------------------------------------------------------------
constexpr int W64_NB = 4;  // chains per wavefront (the endpoint registers are shared between them)

template <int NPL, typename PT, bool UNI, bool PUSH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_layer64(
    const double* __restrict__ endpoints, int64_t e_stride, const double* __restrict__ step, const PT* __restrict__ mu,
    const PT* __restrict__ scale, int64_t p_stride, const int32_t* __restrict__ sym_in, int32_t* __restrict__ sym_out,
    const double* __restrict__ centres, int64_t c_stride, float* __restrict__ centre_out, uint64_t* __restrict__ head,
    uint32_t* __restrict__ stack, int32_t* __restrict__ len, int64_t cap, int B, int D, int bits, int quantbits, int nb,
    int32_t* __restrict__ status) {
    constexpr int K = NPL * 64;
    const int lane = threadIdx.x & 63;
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));  // residue = state index
    if (j >= D) return;
    const int nrows = (D - j + 63) >> 6;  // dims j, j + 64, ...; host guarantees nrows <= 64
    const int b0 = blockIdx.y * nb;
    const int nc = min(B - b0, nb);
    const double M = (double)((1ll << bits) - (1ll << quantbits));
    const uint32_t mask = (1u << bits) - 1u;

    // per-chain state of residue j (wave-uniform: scalar registers)
    uint64_t h[W64_NB];
    int n[W64_NB], st[W64_NB];
    uint32_t wnext[W64_NB], symreg[W64_NB];
    const uint32_t* stk[W64_NB];
#pragma unroll
    for (int c = 0; c < W64_NB; ++c) {
        const int b = min(b0 + c, B - 1);
        const int64_t sj = (int64_t)b * 64 + j;
        h[c] = head[sj];
        n[c] = len[sj];
        stk[c] = stack + sj * cap;
        st[c] = (c < nc) ? status[b] : -1;  // -1: no such chain; > 0: failed earlier, skipped
        wnext[c] = PUSH ? 0u : stk[c][max(n[c] - 1, 0)];
        symreg[c] = 0u;
    }

    for (int ii = 0; ii < nrows; ++ii) {
        const int i = PUSH ? ii : nrows - 1 - ii;  // pushes ascending, pops descending (mnist_compress.py:50,60)
        const int d = 64 * i + j;
        double e[NPL];
        const double* er = endpoints + (int64_t)d * e_stride + lane * NPL;
#pragma unroll
        for (int k = 0; k < NPL; ++k) e[k] = (lane * NPL + k < K - 1) ? er[k] : 0.0;
        const double hstep = UNI ? step[d] : 0.0;
        if (UNI) {
#pragma unroll
            for (int k = 1; k < NPL; ++k) e[k] = e[k] - fma((double)k, hstep, e[0]);
        }
#pragma unroll
        for (int c = 0; c < W64_NB; ++c) {
            if (st[c] != BS_ST_OK) continue;
            const int b = b0 + c;
            const int64_t prow = (int64_t)b * p_stride + d;
            const double m_ = (double)mu[prow];
            const double sc_ = (double)scale[prow];
            const double rs = recip_scale(sc_);
            const bool okp = (sc_ > 0.0) && (rs > 0.0) && (fabs(m_) < __builtin_huge_val());
            Bins<NPL> bn;
            const bool dom = logistic_row<NPL, UNI>(e, hstep, m_, rs, M, lane, bn);
            bool bad;
            uint32_t cstart = bump_and_scan<NPL>(bn, lane, bits, bad);
            if (__ballot(bad || !dom) != 0ull || !okp) { st[c] = BS_ST_BADTABLE; continue; }
            Bins<NPL> cum;  // cum[k] = c of this lane's k-th bin
            {
                uint32_t cc = cstart;
#pragma unroll
                for (int k = 0; k < NPL; ++k) { cum.t[k] = cc; cc += bn.t[k] + 1u; }
            }
            if (!PUSH) {
                const uint32_t m = (uint32_t)h[c] & mask;
                const int L = __popcll(__ballot(cstart <= m)) - 1;  // c_0 = 0 <= m: L >= 0
                int lo = 0;
#pragma unroll
                for (int stp = NPL / 2; stp > 0; stp >>= 1) {
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cum.t[lo + stp], L);
                    lo = (v <= m) ? lo + stp : lo;
                }
                lo = __builtin_amdgcn_readfirstlane(lo);
                const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)cum.t[lo], L);
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)bn.t[lo], L) + 1u;
                uint64_t hh = (uint64_t)f * (h[c] >> bits) + (uint64_t)(m - cs);
                if (hh < (1ull << 32)) {  // mnist_compress.py:65-66
                    if (n[c] <= 0) { st[c] = BS_ST_UNDERFLOW; continue; }
                    hh = (hh << 32) | (uint64_t)wnext[c];
                    --n[c];
                    wnext[c] = stk[c][max(n[c] - 1, 0)];
                }
                h[c] = hh;
                const uint32_t s = (uint32_t)(L * NPL + lo);
                symreg[c] = (lane == i) ? s : symreg[c];  // lane i keeps the symbol of row i
            } else {
                const int s = __builtin_amdgcn_readfirstlane(sym_in[(int64_t)b * D + d]);
                if (s < 0 || s >= K) { st[c] = BS_ST_BADSYMBOL; continue; }
                const int L = s / NPL, lo = s % NPL;
                const uint64_t cs = (uint32_t)__builtin_amdgcn_readlane((int)cum.t[lo], L);
                const uint64_t f = (uint32_t)__builtin_amdgcn_readlane((int)bn.t[lo], L) + 1u;
                uint64_t hh = h[c];
                if ((hh >> (64 - bits)) >= f) {  // head >= 2^(64-bits) * f, mnist_compress.py:52-54
                    if ((int64_t)n[c] >= cap) { st[c] = BS_ST_OVERFLOW; continue; }
                    if (lane == 0) const_cast<uint32_t*>(stk[c])[n[c]] = (uint32_t)hh;
                    ++n[c];
                    hh >>= 32;
                }
                // head // f, head % f (:55): the quotient is below 2^(64-bits+1); float64 estimate, one repair
                uint64_t q = (uint64_t)((double)hh * (1.0 / (double)f));
                int64_t r = (int64_t)(hh - q * f);
                if (r < 0) { --q; r += (int64_t)f; }
                else if (r >= (int64_t)f) { ++q; r -= (int64_t)f; }
                hh = (q << bits) + (uint64_t)r + cs;
                h[c] = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hh) |
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hh >> 32)) << 32);
            }
        }
    }

#pragma unroll
    for (int c = 0; c < W64_NB; ++c) {
        if (st[c] < 0) continue;
        const int b = b0 + c;
        const int64_t sj = (int64_t)b * 64 + j;
        if (!PUSH) {  // symbols (and their bin centres) of this residue: lane i <-> dim 64 i + j
            if (lane < nrows) {
                const int d = 64 * lane + j;
                const int64_t o = (int64_t)b * D + d;
                const uint32_t sy = (st[c] == BS_ST_OK) ? symreg[c] : 0u;
                sym_out[o] = (int32_t)sy;
                if (centres) centre_out[o] = (st[c] == BS_ST_OK) ? (float)centres[(int64_t)d * c_stride + sy] : 0.0f;
            }
        }
        if (lane == 0) {
            if (st[c] == BS_ST_OK) {
                head[sj] = h[c];
                len[sj] = n[c];
            } else if (status[b] == BS_ST_OK) {
                status[b] = st[c];  // first error sticks (any residue of the chain may report it)
            }
        }
    }
}

__global__ void k_gather_centres(const double* __restrict__ centres, int64_t c_stride,
                                 const int32_t* __restrict__ sym, int64_t total, int D, int K,
                                 float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    int s = sym[i];
    s = min(max(s, 0), K - 1);
    out[i] = (float)centres[(int64_t)d * c_stride + s];
}

__global__ void k_sigmoid(const double* __restrict__ t, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = det_sigmoid(t[i]);
}

// ------------------------------------------------------------------------------------------
// self test of the wave primitives against shuffle-based restatements
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_selftest(unsigned long long* failures) {
    const int lane = threadIdx.x;
    unsigned long long bad = 0;
    uint32_t x = (uint32_t)(lane * 2654435761u + blockIdx.x * 40503u) >> 8;
    // reference inclusive scan through LDS
    __shared__ uint32_t sh[64];
    sh[lane] = x;
    __syncthreads();
    uint32_t ref = 0, rmax = 0;
    for (int i = 0; i <= lane; ++i) ref += sh[i];
    for (int i = 0; i < 64; ++i) rmax = max(rmax, sh[i]);
    if (wave_incl_scan_add(x) != ref) bad++;
    if (wave_max_u32(x) != rmax) bad++;
    const double dv = (double)x * 0.25;
    const double up = lane_shift_up_f64(dv);
    if (lane > 0 && up != (double)sh[lane - 1] * 0.25) bad++;
    // DPP wave_shr:1 hand-over of the systolic push: lane i <- lane i-1, lane 0 keeps its own value
    const uint32_t below = from_lane_below(1000u + (uint32_t)lane, x);
    if (below != (lane == 0 ? 1000u : sh[lane - 1])) bad++;
    // bump_and_scan on a 4-bins-per-lane row whose maximum repeats
    uint32_t f[4];
    uint32_t tot = 0;
    for (int i = 0; i < 4; ++i) f[i] = 1 + ((x >> (3 * i)) & 7u);
    if (lane == 9 || lane == 40) f[2] = 100;  // tie: first (lane 9, bin 2) must win
    if (lane == 9) f[3] = 100;                // ... also against a later bin of the same lane
    __shared__ uint32_t fall[256];
    for (int i = 0; i < 4; ++i) fall[lane * 4 + i] = f[i];
    __syncthreads();
    for (int i = 0; i < 256; ++i) tot += fall[i];
    bool badrow;
    Bins<4> bn;
    for (int i = 0; i < 4; ++i) bn.t[i] = f[i] - 1u;
    uint32_t c = bump_and_scan<4>(bn, lane, 20, badrow);
    uint32_t rc = 0;
    for (int i = 0; i < lane * 4; ++i) rc += fall[i] + ((i == 9 * 4 + 2) ? ((1u << 20) - tot) : 0u);
    if (c != rc) bad++;
    if (lane == 9 && bn.t[2] + 1u != 100 + ((1u << 20) - tot)) bad++;
    if (lane == 9 && bn.t[3] + 1u != 100) bad++;
    if (lane == 40 && bn.t[2] + 1u != 100) bad++;
    if (bad) atomicAdd(failures, bad);
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_rc() { return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int NPL, typename PT>
int launch_logistic(int mode, const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale,
                    const int32_t* sym, int B, int D, int bits, int quantbits, uint32_t* out0, uint32_t* out1,
                    int64_t ld, int32_t* status, hipStream_t st) {
    // chains per wavefront: amortise the endpoint fetch, but keep >= ~8 waves per SIMD in flight
    int nb = 4;
    while (nb > 1 && (int64_t)((D + 3) / 4) * ((B + nb - 1) / nb) < 4096) nb >>= 1;
    dim3 grid((D + 3) / 4, (B + nb - 1) / nb), block(256);
    const PT* m = static_cast<const PT*>(mu);
    const PT* s = static_cast<const PT*>(scale);
#define BS_LAUNCH(MODE, UNI)                                                                                     \
    hipLaunchKernelGGL((k_logistic<NPL, PT, MODE, UNI>), grid, block, 0, st, endpoints, e_stride, step, m, s, sym, B, D, \
                       bits, quantbits, nb, out0, out1, ld, status)
    if (step) {  // CDF spec 2 (uniform bins); host dispatch guarantees NPL >= 4
        if (mode == M_PIVOT) BS_LAUNCH((NPL >= 4 ? M_PIVOT : M_LINEAR), (NPL >= 4));
        else if (mode == M_WAVE) BS_LAUNCH((NPL >= 4 ? M_WAVE : M_LINEAR), (NPL >= 4));
        else if (mode == M_LINEAR_VEC) BS_LAUNCH((NPL >= 4 ? M_LINEAR_VEC : M_LINEAR), (NPL >= 4));
        else if (mode == M_ENCODE) BS_LAUNCH(M_ENCODE, (NPL >= 4));
        else BS_LAUNCH(M_LINEAR, (NPL >= 4));
    } else if (mode == M_WAVE && NPL >= 4) BS_LAUNCH((NPL >= 4 ? M_WAVE : M_LINEAR), false);
    else if (mode == M_LINEAR_VEC && NPL >= 4) BS_LAUNCH((NPL >= 4 ? M_LINEAR_VEC : M_LINEAR), false);
    else if (mode == M_ENCODE) BS_LAUNCH(M_ENCODE, false);
    else BS_LAUNCH(M_LINEAR, false);
#undef BS_LAUNCH
    return launch_rc();
}

template <typename PT>
int dispatch_logistic(int K, int mode, const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale,
                      const int32_t* sym, int B, int D, int bits, int quantbits, uint32_t* out0, uint32_t* out1,
                      int64_t ld, int32_t* status, hipStream_t st) {
#define BS_CASE(NPL)                                                                                             \
    case 64 * NPL:                                                                                               \
        return launch_logistic<NPL, PT>(mode, endpoints, e_stride, step, mu, scale, sym, B, D, bits, quantbits, out0, \
                                        out1, ld, status, st)
    if (step && K < 256) return BS_EUNSUPPORTED;  // CDF spec 2 is defined for K >= 256 (groups of K/64 >= 4 bins)
    switch (K) {
        BS_CASE(1);
        BS_CASE(2);
        BS_CASE(4);
        BS_CASE(8);
        BS_CASE(16);
        BS_CASE(32);
        default:
            return BS_EUNSUPPORTED;
    }
#undef BS_CASE
}


template <typename PT, bool PUSH>
int dispatch_layer64(int K, const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale,
                     int64_t p_stride, const int32_t* sym_in, int32_t* sym_out, const double* centres, int64_t c_stride,
                     float* centre_out, uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, int B, int D, int bits,
                     int quantbits, int32_t* status, hipStream_t st) {
    // chains per wavefront: share the endpoint registers, but keep the chip full (64 residues x B / nb waves)
    int nb = W64_NB;
    while (nb > 1 && (int64_t)64 * ((B + nb - 1) / nb) < 8192) nb >>= 1;
    dim3 grid(16, (B + nb - 1) / nb), block(256);
    const PT* m = static_cast<const PT*>(mu);
    const PT* s = static_cast<const PT*>(scale);
#define BS_L64(NPL, UNI)                                                                                          \
    hipLaunchKernelGGL((k_layer64<NPL, PT, UNI, PUSH>), grid, block, 0, st, endpoints, e_stride, step, m, s, p_stride, \
                       sym_in, sym_out, centres, c_stride, centre_out, head, stack, len, cap, B, D, bits, quantbits, nb, status)
    switch (K) {
        case 256: if (step) BS_L64(4, true); else BS_L64(4, false); break;
        case 512: if (step) BS_L64(8, true); else BS_L64(8, false); break;
        case 1024: if (step) BS_L64(16, true); else BS_L64(16, false); break;
        default: return BS_EUNSUPPORTED;
    }
#undef BS_L64
    return launch_rc();
}

}  // namespace

template <typename PT>
int dispatch_pop_pivot(uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, const uint32_t* piv, int64_t ld,
                       const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale, int B,
                       int D, int K, int bits, int quantbits, int32_t* sym_out, const double* centres, int64_t c_stride,
                       float* centre_out, int32_t* status, hipStream_t st) {
    dim3 grid(B), block(64);
    const PT* m = static_cast<const PT*>(mu);
    const PT* s = static_cast<const PT*>(scale);
#define BS_POPP(NPL, PF)                                                                                              \
    hipLaunchKernelGGL((k_rans_pop_pivot<NPL, PT, PF>), grid, block, (size_t)D * 4, st, head, stack, len, cap, piv, ld, endpoints, \
                       e_stride, step, m, s, D, bits, quantbits, sym_out, centres, c_stride, centre_out, status)
    if (K == 256) BS_POPP(4, BS_POP_PF);
    else if (K == 512) BS_POPP(8, BS_POP_PF);
    else if (K == 1024) BS_POPP(16, BS_POP_PF);
    else if (K == 2048) BS_POPP(32, BS_POP_PF);
    else return BS_EUNSUPPORTED;
#undef BS_POPP
    return launch_rc();
}

extern "C" {

int bs_abi_version(void) { return BS_ABI_VERSION; }
int bs_cdf_spec(void) { return BS_CDF_SPEC; }

const char* bs_strerror(int code) {
    switch (code) {
        case BS_OK: return "ok";
        case BS_EINVAL: return "invalid argument";
        case BS_EUNSUPPORTED: return "alphabet size not supported by the fused logistic kernels (need K = 64*2^n <= 2048)";
        case BS_ELAUNCH: return "HIP kernel launch failed";
        case BS_ST_UNDERFLOW: return "stack underflow (too few initial bits)";
        case BS_ST_OVERFLOW: return "stack overflow (capacity exhausted)";
        case BS_ST_BADTABLE: return "table invariant violated (cdf[K] != 2^bits or zero frequency)";
        case BS_ST_BADSYMBOL: return "symbol out of range";
        default: return "unknown code";
    }
}

int bs_table_rows_f64(const double* pmf, int64_t rows, int K, int bits, int quantbits, uint32_t* f_out,
                      uint32_t* cdf_out, int64_t ld, int32_t* status, void* stream) {
    if (!pmf || !cdf_out || rows < 0 || K < 1 || ld < K + 1 || bits < 1 || bits > 31 || quantbits < 0 ||
        quantbits >= bits)
        return BS_EINVAL;
    if (rows == 0) return BS_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = S(stream);
    switch (K) {
        case 256:
            hipLaunchKernelGGL(k_table_rows<4>, grid, block, 0, st, pmf, rows, bits, quantbits, f_out, cdf_out, ld, status);
            break;
        case 1024:
            hipLaunchKernelGGL(k_table_rows<16>, grid, block, 0, st, pmf, rows, bits, quantbits, f_out, cdf_out, ld, status);
            break;
        default:
            hipLaunchKernelGGL(k_table_rows_generic, grid, block, 0, st, pmf, rows, K, bits, quantbits, f_out, cdf_out, ld,
                               status);
    }
    return launch_rc();
}

int bs_logistic_tables(const double* endpoints, int64_t e_stride, const double* bin_step, const void* mu,
                       const void* scale, int param_dtype, int B, int D, int K, int bits, int quantbits,
                       uint32_t* cdf_out, int64_t ld, int layout, int32_t* status, void* stream) {
    if (!endpoints || !mu || !scale || !cdf_out || B < 0 || D < 0 || bits < 1 || bits > 31 || quantbits < 0 ||
        quantbits >= bits || e_stride < 0)
        return BS_EINVAL;
    int mode;
    if (layout == BS_LAYOUT_LINEAR) {
        if (ld < K + 1) return BS_EINVAL;
        mode = (aligned16(cdf_out) && (ld % 4 == 0)) ? M_LINEAR_VEC : M_LINEAR;
    } else if (layout == BS_LAYOUT_WAVE) {
        if (ld < K + 64 || K < 256) return BS_EINVAL;
        mode = M_WAVE;
    } else if (layout == BS_LAYOUT_PIVOT) {      // 64 x (cumulative value, aux) per row: uniform bins (spec 2) only
        if (ld < 128 || ld % 2 || K < 256 || !bin_step || (reinterpret_cast<uintptr_t>(cdf_out) & 7u)) return BS_EINVAL;
        mode = M_PIVOT;
    } else {
        return BS_EINVAL;
    }
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_logistic<float>(K, mode, endpoints, e_stride, bin_step, mu, scale, nullptr, B, D, bits, quantbits,
                                        cdf_out, nullptr, ld, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_logistic<double>(K, mode, endpoints, e_stride, bin_step, mu, scale, nullptr, B, D, bits, quantbits,
                                         cdf_out, nullptr, ld, status, S(stream));
    return BS_EINVAL;
}

int bs_logistic_fc(const double* endpoints, int64_t e_stride, const double* bin_step, const void* mu, const void* scale,
                   int param_dtype, const int32_t* sym, int B, int D, int K, int bits, int quantbits, uint32_t* f_out,
                   uint32_t* c_out, int32_t* status, void* stream) {
    if (!endpoints || !mu || !scale || !sym || !f_out || !c_out || !status || B < 0 || D < 0 || bits < 1 ||
        bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0)
        return BS_EINVAL;
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_logistic<float>(K, M_ENCODE, endpoints, e_stride, bin_step, mu, scale, sym, B, D, bits, quantbits,
                                        f_out, c_out, 0, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_logistic<double>(K, M_ENCODE, endpoints, e_stride, bin_step, mu, scale, sym, B, D, bits, quantbits,
                                         f_out, c_out, 0, status, S(stream));
    return BS_EINVAL;
}

int bs_rans_push(uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, const uint32_t* f, const uint32_t* c,
                 int B, int D, int bits, int32_t* status, void* stream) {
    if (!head || !stack || !len || !f || !c || !status || B < 0 || D < 0 || cap < 0 || bits < 1 || bits > 31)
        return BS_EINVAL;
    if (B == 0 || D == 0) return BS_OK;
    hipLaunchKernelGGL(k_rans_push, dim3(B), dim3(64), 0, S(stream), head, stack, len, cap, f, c, B, D, bits, status);
    return launch_rc();
}

int bs_rans_push_table(uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, const uint32_t* cdf,
                       int64_t chain_stride, int64_t ld, int layout, const int32_t* sym, int B, int D, int K, int bits,
                       int32_t* status, void* stream) {
    if (!head || !stack || !len || !cdf || !sym || !status || B < 0 || D < 0 || cap < 0 || K < 1 ||
        chain_stride < 0 || bits < 1 || bits > 31)
        return BS_EINVAL;
    if (layout == BS_LAYOUT_LINEAR ? ld < K + 1 : (layout != BS_LAYOUT_WAVE || ld < K + 64 || K % 256 != 0))
        return BS_EINVAL;
    if (B == 0 || D == 0) return BS_OK;
    hipLaunchKernelGGL(k_rans_push_table, dim3(B), dim3(64), 0, S(stream), head, stack, len, cap, cdf, chain_stride,
                       ld, layout, sym, B, D, K, bits, status);
    return launch_rc();
}

int bs_rans_pop(uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, const uint32_t* cdf, int64_t chain_stride,
                int64_t ld, int layout, int B, int D, int K, int bits, int32_t* sym_out, const double* centres,
                int64_t c_stride, float* centre_out, int32_t* status, void* stream) {
    if (!head || !stack || !len || !cdf || !sym_out || !status || B < 0 || D < 0 || cap < 0 || K < 1 ||
        chain_stride < 0 || bits < 1 || bits > 31 || (centres && !centre_out) || c_stride < 0)
        return BS_EINVAL;
    if (layout == BS_LAYOUT_LINEAR ? ld < K + 1 : (layout != BS_LAYOUT_WAVE || ld < K + 64)) return BS_EINVAL;
    if (B == 0 || D == 0) return BS_OK;
    hipStream_t st = S(stream);
    // fast paths: 16-byte aligned rows, whole 64-row chunks, symbols of one chain fit in LDS
    const bool fast = aligned16(cdf) && (ld % 4 == 0) && (chain_stride % 4 == 0) && (D % 64 == 0) && (D <= 16384);
    dim3 grid(B), block(64);
#define BS_POP(ROW, PF)                                                                                          \
    hipLaunchKernelGGL((k_rans_pop<ROW, PF>), grid, block, (size_t)D * 4, st, head, stack, len, cap, cdf,        \
                       chain_stride, ld, D, bits, sym_out, centres, c_stride, centre_out, status)
    if (layout == BS_LAYOUT_WAVE) {
        if (!fast) return BS_EINVAL;  // the wave layout only exists for the fast path
        if ((int64_t)D * ld * 4 >= (1ll << 31)) return BS_EINVAL;  // one chain's rows must fit a 32-bit buffer offset
#define BS_POPW(NR, PF)                                                                                          \
    hipLaunchKernelGGL((k_rans_pop_wave<NR, PF>), grid, block, (size_t)D * 4, st, head, stack, len, cap, cdf,    \
                       chain_stride, ld, D, bits, sym_out, centres, c_stride, centre_out, status)
        if (K == 256) BS_POPW(4, 32);
        else if (K == 512) BS_POPW(8, 16);
        else if (K == 1024) BS_POPW(16, 8);
        else if (K == 2048) BS_POPW(32, 8);
        else return BS_EUNSUPPORTED;
#undef BS_POPW
    } else if (fast && K == 256) BS_POP(RowRegs<1>, 8);
    else if (fast && K == 512) BS_POP(RowRegs<2>, 8);
    else if (fast && K == 1024) BS_POP(RowRegs<4>, 8);
    else if (fast && K == 2048) BS_POP(RowRegs<8>, 4);
    else
        hipLaunchKernelGGL(k_rans_pop_generic, grid, block, 0, st, head, stack, len, cap, cdf, chain_stride, ld, D, K,
                           bits, sym_out, centres, c_stride, centre_out, status);
#undef BS_POP
    return launch_rc();
}

int bs_rans_pop_pivot(uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, const uint32_t* pivots, int64_t ld,
                      const double* endpoints, int64_t e_stride, const double* bin_step, const void* mu, const void* scale,
                      int param_dtype, int B, int D, int K, int bits, int quantbits, int32_t* sym_out, const double* centres,
                      int64_t c_stride, float* centre_out, int32_t* status, void* stream) {
    if (!head || !stack || !len || !pivots || !endpoints || !bin_step || !mu || !scale || !sym_out || !status || B < 0 ||
        D < 0 || cap < 0 || bits < 1 || bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || c_stride < 0 ||
        (centres && !centre_out) || ld < 128 || ld % 2 || (reinterpret_cast<uintptr_t>(pivots) & 7u))
        return BS_EINVAL;
    if (D % 64 != 0 || D > 16384) return BS_EUNSUPPORTED;      // whole 64-symbol chunks; a chain's symbols fit in LDS
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_pop_pivot<float>(head, stack, len, cap, pivots, ld, endpoints, e_stride, bin_step, mu, scale, B, D, K,
                                         bits, quantbits, sym_out, centres, c_stride, centre_out, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_pop_pivot<double>(head, stack, len, cap, pivots, ld, endpoints, e_stride, bin_step, mu, scale, B, D, K,
                                          bits, quantbits, sym_out, centres, c_stride, centre_out, status, S(stream));
    return BS_EINVAL;
}

int bs_layer_pop64(uint64_t* head64, uint32_t* stack64, int32_t* len64, int64_t cap, const double* endpoints,
                   int64_t e_stride, const double* bin_step, const void* mu, const void* scale, int64_t p_stride,
                   int param_dtype, int B, int D, int K, int bits, int quantbits, int32_t* sym_out, const double* centres,
                   int64_t c_stride, float* centre_out, int32_t* status, void* stream) {
    if (!head64 || !stack64 || !len64 || !endpoints || !mu || !scale || !sym_out || !status || B < 0 || D < 0 || cap < 0 ||
        bits < 1 || bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || p_stride < 0 || c_stride < 0 ||
        (centres && !centre_out))
        return BS_EINVAL;
    if (D > 4096) return BS_EUNSUPPORTED;  // one lane per row of a residue
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_layer64<float, false>(K, endpoints, e_stride, bin_step, mu, scale, p_stride, nullptr, sym_out, centres,
                                              c_stride, centre_out, head64, stack64, len64, cap, B, D, bits, quantbits,
                                              status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_layer64<double, false>(K, endpoints, e_stride, bin_step, mu, scale, p_stride, nullptr, sym_out, centres,
                                               c_stride, centre_out, head64, stack64, len64, cap, B, D, bits, quantbits,
                                               status, S(stream));
    return BS_EINVAL;
}

int bs_layer_push64(uint64_t* head64, uint32_t* stack64, int32_t* len64, int64_t cap, const double* endpoints,
                    int64_t e_stride, const double* bin_step, const void* mu, const void* scale, int64_t p_stride,
                    int param_dtype, const int32_t* sym, int B, int D, int K, int bits, int quantbits, int32_t* status,
                    void* stream) {
    if (!head64 || !stack64 || !len64 || !endpoints || !mu || !scale || !sym || !status || B < 0 || D < 0 || cap < 0 ||
        bits < 1 || bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || p_stride < 0)
        return BS_EINVAL;
    if (D > 4096) return BS_EUNSUPPORTED;
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_layer64<float, true>(K, endpoints, e_stride, bin_step, mu, scale, p_stride, sym, nullptr, nullptr, 0,
                                             nullptr, head64, stack64, len64, cap, B, D, bits, quantbits, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_layer64<double, true>(K, endpoints, e_stride, bin_step, mu, scale, p_stride, sym, nullptr, nullptr, 0,
                                              nullptr, head64, stack64, len64, cap, B, D, bits, quantbits, status, S(stream));
    return BS_EINVAL;
}

int bs_gather_centres(const double* centres, int64_t c_stride, const int32_t* sym, int B, int D, int K,
                      float* centre_out, void* stream) {
    if (!centres || !sym || !centre_out || B < 0 || D < 0 || K < 1 || c_stride < 0) return BS_EINVAL;
    const int64_t total = (int64_t)B * D;
    if (total == 0) return BS_OK;
    hipLaunchKernelGGL(k_gather_centres, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), centres,
                       c_stride, sym, total, D, K, centre_out);
    return launch_rc();
}

int bs_sigmoid_f64(const double* t, int64_t n, double* out, void* stream) {
    if (!t || !out || n < 0) return BS_EINVAL;
    if (n == 0) return BS_OK;
    hipLaunchKernelGGL(k_sigmoid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), t, n, out);
    return launch_rc();
}

int bs_selftest(int64_t* failures_host, void* stream) {
    if (!failures_host) return BS_EINVAL;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, sizeof(unsigned long long)) != hipSuccess) return BS_ELAUNCH;
    hipStream_t st = S(stream);
    if (hipMemsetAsync(d, 0, sizeof(unsigned long long), st) != hipSuccess) { (void)hipFree(d); return BS_ELAUNCH; }
    hipLaunchKernelGGL(k_selftest, dim3(64), dim3(64), 0, st, d);
    unsigned long long h = ~0ull;
    hipError_t e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) return BS_ELAUNCH;
    *failures_host = (int64_t)h;
    return BS_OK;
}

}  // extern "C"
