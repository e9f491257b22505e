// layer64.hip -- BS_FORMAT_WAVE64: table + coding step in one launch (k_layer64)
// (one of the translation units of libbitswap_hip.so; shared device helpers: bitswap_dev.h; entry points: include/bitswap_hip.h)
#include "bitswap_dev.h"

namespace {

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

template <int NPL, typename PT, int SPEC, bool PUSH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_layer64(
    const double* __restrict__ endpoints, int64_t e_stride, const double* __restrict__ step, const PT* __restrict__ mu,
    const PT* __restrict__ scale, int64_t p_stride, const int32_t* __restrict__ sym_in, int32_t* __restrict__ sym_out,
    const double* __restrict__ centres, int64_t c_stride, float* __restrict__ centre_out, uint64_t* __restrict__ head,
    uint32_t* __restrict__ stack, int32_t* __restrict__ len, int64_t cap, int B, int D, int bits, int quantbits, int nb,
    int32_t* __restrict__ status) {
    constexpr int K = NPL * 64;
    constexpr bool UNI = SPEC >= 2;      // CDF specs 2 and 3: rows of uniform-width bins
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
            if (lane == 63) e[NPL - 1] = 0.0;   // the K-th, virtual endpoint: on the progression (k_logistic does the same)
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
            const bool dom = logistic_row<NPL, SPEC>(e, hstep, m_, rs, M, lane, bn);
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

template <typename PT, bool PUSH>
int dispatch_layer64(int K, int spec, const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale,
                     int64_t p_stride, const int32_t* sym_in, int32_t* sym_out, const double* centres, int64_t c_stride,
                     float* centre_out, uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, int B, int D, int bits,
                     int quantbits, int32_t* status, hipStream_t st) {
    // chains per wavefront: share the endpoint registers, but keep the chip full (64 residues x B / nb waves)
    int nb = W64_NB;
    while (nb > 1 && (int64_t)64 * ((B + nb - 1) / nb) < 8192) nb >>= 1;
    dim3 grid(16, (B + nb - 1) / nb), block(256);
    const PT* m = static_cast<const PT*>(mu);
    const PT* s = static_cast<const PT*>(scale);
#define BS_L64(NPL, SPEC)                                                                                         \
    hipLaunchKernelGGL((k_layer64<NPL, PT, SPEC, PUSH>), grid, block, 0, st, endpoints, e_stride, step, m, s, p_stride, \
                       sym_in, sym_out, centres, c_stride, centre_out, head, stack, len, cap, B, D, bits, quantbits, nb, status)
    switch (K) {
        case 256: if (spec == 4) BS_L64(4, 4); else if (spec == 3) BS_L64(4, 3); else if (spec == 2) BS_L64(4, 2); else BS_L64(4, 1); break;
        case 512: if (spec == 4) BS_L64(8, 4); else if (spec == 3) BS_L64(8, 3); else if (spec == 2) BS_L64(8, 2); else BS_L64(8, 1); break;
        case 1024: if (spec == 4) BS_L64(16, 4); else if (spec == 3) BS_L64(16, 3); else if (spec == 2) BS_L64(16, 2); else BS_L64(16, 1); break;
        default: return BS_EUNSUPPORTED;
    }
#undef BS_L64
    return launch_rc();
}

}  // namespace

extern "C" {

int bs_layer_pop64(uint64_t* head64, uint32_t* stack64, int32_t* len64, int64_t cap, const double* endpoints,
                   int64_t e_stride, const double* bin_step, int cdf_spec, const void* mu, const void* scale, int64_t p_stride,
                   int param_dtype, int B, int D, int K, int bits, int quantbits, int32_t* sym_out, const double* centres,
                   int64_t c_stride, float* centre_out, int32_t* status, void* stream) {
    const int spec = cdf_spec == 1 ? 1 : (cdf_spec >= 2 && cdf_spec <= 4 && bin_step) ? cdf_spec : 0;
    if (!spec) return BS_EINVAL;
    if (spec == 1) bin_step = nullptr;
    if (!head64 || !stack64 || !len64 || !endpoints || !mu || !scale || !sym_out || !status || B < 0 || D < 0 || cap < 0 ||
        bits < 1 || bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || p_stride < 0 || c_stride < 0 ||
        (centres && !centre_out))
        return BS_EINVAL;
    if (D > 4096) return BS_EUNSUPPORTED;  // one lane per row of a residue
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_layer64<float, false>(K, spec, endpoints, e_stride, bin_step, mu, scale, p_stride, nullptr, sym_out, centres,
                                              c_stride, centre_out, head64, stack64, len64, cap, B, D, bits, quantbits,
                                              status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_layer64<double, false>(K, spec, endpoints, e_stride, bin_step, mu, scale, p_stride, nullptr, sym_out, centres,
                                               c_stride, centre_out, head64, stack64, len64, cap, B, D, bits, quantbits,
                                               status, S(stream));
    return BS_EINVAL;
}

int bs_layer_push64(uint64_t* head64, uint32_t* stack64, int32_t* len64, int64_t cap, const double* endpoints,
                    int64_t e_stride, const double* bin_step, int cdf_spec, const void* mu, const void* scale, int64_t p_stride,
                    int param_dtype, const int32_t* sym, int B, int D, int K, int bits, int quantbits, int32_t* status,
                    void* stream) {
    const int spec = cdf_spec == 1 ? 1 : (cdf_spec >= 2 && cdf_spec <= 4 && bin_step) ? cdf_spec : 0;
    if (!spec) return BS_EINVAL;
    if (spec == 1) bin_step = nullptr;
    if (!head64 || !stack64 || !len64 || !endpoints || !mu || !scale || !sym || !status || B < 0 || D < 0 || cap < 0 ||
        bits < 1 || bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || p_stride < 0)
        return BS_EINVAL;
    if (D > 4096) return BS_EUNSUPPORTED;
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_layer64<float, true>(K, spec, endpoints, e_stride, bin_step, mu, scale, p_stride, sym, nullptr, nullptr, 0,
                                             nullptr, head64, stack64, len64, cap, B, D, bits, quantbits, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_layer64<double, true>(K, spec, endpoints, e_stride, bin_step, mu, scale, p_stride, sym, nullptr, nullptr, 0,
                                              nullptr, head64, stack64, len64, cap, B, D, bits, quantbits, status, S(stream));
    return BS_EINVAL;
}

}  // extern "C"
