// bitswap_dev.h -- device helpers shared by the translation units of libbitswap_hip.so (tables.hip, pop.hip, push.hip,
// layer64.hip, bitswap_hip.hip): wave64 DPP primitives, the deterministic float64 exponential / sigmoid of BS_CDF_SPEC,
// the integer tail of ANS.__init__ (bump_and_scan) and logistic_row, the per-row CDF evaluation every table-building
// kernel runs (k_logistic, k_layer64, and the group rebuild of k_rans_pop_pivot must produce the same bits: one source).
// Everything here sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
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

// fma(a, b, c) with the constant c in a scalar register pair, as ONE three-source instruction.  Left to itself hipcc keeps
// the eleven polynomial constants of det_exp in VECTOR registers (22 of them, live across the whole row loop of the table
// kernels) and writes every Horner step as v_mov_b64 + v_fmac_f64 -- 9 extra issue slots per exponential.  Same IEEE
// operation either way.
__device__ __forceinline__ double fma_sconst(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// exp(a) for the argument clamped to [-700, hi]: the exponential half of the deterministic sigmoid (hi = 700; the anchors
// of CDF spec 3 clamp at 41)
__device__ __forceinline__ double det_exp_hi(double a, double hi) {
    a = fmin(fmax(a, -700.0), hi);
    const double kd = rint(a * 0x1.71547652b82fep+0);
    double r = fma(-kd, 0x1.62e42fee00000p-1, a);
    r = fma(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.af631e4ea6521p-26;
    p = fma_sconst(p, r, 0x1.28b4068ef93d2p-22);
    p = fma_sconst(p, r, 0x1.71ddf573e8618p-19);
    p = fma_sconst(p, r, 0x1.a01991ab61789p-16);
    p = fma_sconst(p, r, 0x1.a01a01b143bc8p-13);
    p = fma_sconst(p, r, 0x1.6c16c187fc4dep-10);
    p = fma_sconst(p, r, 0x1.111111110f224p-7);
    p = fma_sconst(p, r, 0x1.555555554f0ccp-5);
    p = fma_sconst(p, r, 0x1.555555555555ap-3);
    p = fma_sconst(p, r, 0x1.0000000000011p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kd);
}

__device__ __forceinline__ double det_exp(double a) { return det_exp_hi(a, 700.0); }

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

// ------------------------------------------------------------------------------------------
// BS_CDF_SPEC 3: spec 2's denominators x_b = 1 + E_b, inverted per block of N <= 16 bins with ONE correctly rounded
// reciprocal -- a balanced product tree up, the root inverted, one multiplication per node down (heap order: node k has
// the children 2k and 2k+1, the leaves are N .. 2N-1).  oracle/bitswap_oracle.c::det3_row_cdf is the C restatement.
// A quotient costs 3 multiplications + 9/N slots instead of 9: 54 instead of 144 issue slots per 16 bins.
// ------------------------------------------------------------------------------------------
#define BS_SPEC3_FAST_HR 8.0      // rows with NPL * |h / scale| below this take the batch inversion, the others spec 2's arithmetic
#define BS_SPEC3_ANCHOR_HI 41.0   // clamp of the anchor's exponent: x <= 1 + e^41 < 2^59.2, a root of 16 stays below 2^947
// A node of the product tree over N leaves, walked depth-first so that the quotients come out in bin order and a node's
// value dies as soon as both of its children have their inverse (the table kernels are register-bound).
template <int N>
struct InvTree {
    InvTree<N / 2> l, r;
    double t;
    __device__ __forceinline__ void up(const double* x) {
        l.up(x);
        r.up(x + N / 2);
        t = l.t * r.t;
    }
    template <typename F>
    __device__ __forceinline__ void down(double inv, int i0, F&& leaf) {
        l.down(inv * r.t, i0, leaf);
        r.down(inv * l.t, i0 + N / 2, leaf);
    }
};
template <>
struct InvTree<1> {
    double t;
    __device__ __forceinline__ void up(const double* x) { t = x[0]; }
    template <typename F>
    __device__ __forceinline__ void down(double inv, int i0, F&& leaf) { leaf(i0, inv, t); }
};
// leaf(i, ~1 / x[i], x[i]) for i = 0 .. N-1 in order (spec 3 uses the quotient as it comes; spec 4 corrects it with x[i])
template <int N, typename F>
__device__ __forceinline__ void tree_inverse(const double* x, int i0, F&& leaf) {
    static_assert(N == 2 || N == 4 || N == 8 || N == 16, "block of a power of two, at most 16 bins");
    InvTree<N> tr;
    tr.up(x);
    tr.down(recip_1_to_huge(tr.t), i0, leaf);
}
// BS_CDF_SPEC 4 (round 6): spec 3 with blocks of at most 8 bins and ONE residual correction per quotient,
//   c <- fma(fma(-x, c, 1), c, c),
// which squares the tree's accumulated rounding error (2 log2 n + 1 roundings) away: c is RN(1 / x) in all but ~2^-50 of the cases,
// i.e. spec 2's tables (0.03 ppm of entries off torch.sigmoid's, against 0.2 ppm for spec 3; the reference's own 100-block streams
// reproduced to the word like spec 2, tests/test_oracle.py::HORIZON) for 92 instead of 144 issue slots per 16 bins (spec 3: 54).
// Blocks of 8: half the live tree registers of spec 3's blocks of 16 -- the table kernel fits 5 wavefronts per SIMD instead of 4.
// oracle/bitswap_oracle.c::det4_row_cdf is the C restatement.
template <int SPEC, int NPL>
struct SpecBlock {
    static constexpr int MAXN = SPEC == 4 ? 8 : 16;
    static constexpr int N = NPL < MAXN ? NPL : MAXN;
};
__device__ __forceinline__ double newton_correct(double x, double c) {
    double r;
    // two three-source instructions (left to itself hipcc writes v_mov_b64 + v_fmac_f64 for each: 34 extra moves per row)
    double d;
    asm("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(r) : "v"(x), "v"(c));       // fma(-x, c, 1)
    asm("v_fma_f64 %0, %1, %2, %2" : "=v"(d) : "v"(r), "v"(c));         // fma(r, c, c)
    return d;
}
// fma(-a, b, c) as ONE three-source instruction: left to itself hipcc writes v_mov_b64 + v_fmac_f64 whenever c lives on
__device__ __forceinline__ double fnma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// denominators x[I .. END) of this lane's bins (x[0] = 1 + A is the caller's: no residual, no geometric factor)
template <int NPL, int I, int END>
__device__ __forceinline__ void spec3_denoms(const double (&e)[NPL], double Ars, double A, double qb, double (&x)[NPL]) {
    if constexpr (I < END) {
        x[I] = one_plus_e<NPL, I>(qb, fnma3(Ars, e[I], A));      // e[I]: the residual r_I of the stored endpoint
        spec3_denoms<NPL, I + 1, END>(e, Ars, A, qb, x);
    }
}
// blocks [I0, I0 + NB), [I0 + NB, ...) of a row in order: a block's denominators are built right in front of its tree, so only
// one block's worth of them (and of tree nodes) is live at a time -- with spec 4's blocks of 8 that is what lets the table
// kernel run 5 wavefronts per SIMD
template <int NPL, int SPEC, int I0, typename F>
__device__ __forceinline__ void spec3_blocks(const double (&e)[NPL], double Ars, double A, double qb, double (&x)[NPL], F&& leaf) {
    constexpr int NB = SpecBlock<SPEC, NPL>::N;
    if constexpr (I0 < NPL) {
        if constexpr (I0 == 0) x[0] = 1.0 + A;
        spec3_denoms<NPL, (I0 == 0 ? 1 : I0), I0 + NB>(e, Ars, A, qb, x);
        tree_inverse<NB>(x + I0, I0, leaf);
        spec3_blocks<NPL, SPEC, I0 + NB>(e, Ars, A, qb, x, leaf);
    }
}

// One (chain, dim) row: bn.t[i] = trunc(pmf * M) of this lane's NPL bins (the reference's f - 1).  `e` holds the
// lane's endpoints (spec 1) or its anchor + residuals (specs 2 and 3, see k_logistic; the residual of lane 63's last,
// virtual endpoint is 0).  SPEC: the CDF specification, 1 = generic bins, 2 / 3 = uniform-width bins.
// Returns false when the row leaves the domain of CDF specs 2 / 3: NPL * h / scale < 650.  det_exp clamps its argument to +-700.
// A clamped ANCHOR is harmless on its own: beyond +700 every bin of the lane is exactly 1, beyond -700 the lane's bins
// come out as e^-(700 - b h/scale) <= e^-50 -- too large, but still truncated to the same f = 1 as the true values, so the
// table is the exact one.  What must not be clamped is the geometric factor Q_b = exp(-b h/scale): with h/scale in the
// hundreds (a scale tiny against the bin width: scale < 5e-5 for the pixel bins, 20x below the reference's floor of
// 2/255/8, mnist_train.py:411; reachable only through the C ABI) a lane with a clamped anchor would put 0.5 where the
// cdf is 1e-18, and the cdf would step DOWN into the next lane.  Such rows are not coded: the caller flags
// BS_ST_BADTABLE (oracle/bitswap_oracle.c::layer_in_domain applies the same test); CDF spec 1 takes any scale.
template <int NPL, int SPEC>
__device__ __forceinline__ bool logistic_row(const double (&e)[NPL], double hstep, double m_, double rs, double M, int lane,
                                             Bins<NPL>& bn) {
    double c0, prev;
    bool in_domain = true;
    if constexpr (SPEC >= 2) {
        const double hr = hstep * rs;
        const double qb = det_exp(-((double)(lane & (NPL - 1)) * hr));   // lane b < NPL: Q_b
        const double ta = (e[0] - m_) * rs;
        const double span = (double)NPL * fabs(hr);
        in_domain = span < 650.0;
        bool batch = false;
        if constexpr (SPEC >= 3 && NPL >= 4) batch = span < BS_SPEC3_FAST_HR;      // wave-uniform (mu, scale, h are scalars)
        if (batch) {
            if constexpr (SPEC >= 3 && NPL >= 4) {
                const double A = det_exp_hi(-ta, BS_SPEC3_ANCHOR_HI);
                const double Ars = A * rs;
                double x[NPL];
                c0 = prev = 0.0;
                spec3_blocks<NPL, SPEC, 0>(e, Ars, A, qb, x, [&](int i, double ci, double xi) {
                    if constexpr (SPEC == 4) ci = newton_correct(xi, ci);
                    if (i == 0) {
                        c0 = ci;
                    } else {
                        if (i == NPL - 1 && lane == 63) ci = 1.0;
                        bn.t[i] = trunc_u32((ci - prev) * M);
                    }
                    prev = ci;
                });
            }
        } else {
            const double A = det_exp(-ta);
            c0 = recip_1_to_huge(1.0 + A);
            prev = c0;
            uni_bins<NPL, 1>(e, rs, A, qb, M, lane, prev, bn);
        }
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

// value of lane-1 (DPP wave_shr:1); lane 0 keeps `keep`
__device__ __forceinline__ uint32_t from_lane_below(uint32_t keep, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x138, 0xf, 0xf, false);
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_rc() { return hipGetLastError() == hipSuccess ? BS_OK : BS_ELAUNCH; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace
