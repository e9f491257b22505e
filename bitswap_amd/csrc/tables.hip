// tables.hip -- k_logistic (fused logistic CDF -> integer table: decode rows, pivot hand-off, encode (f, c)) and k_table_rows (ANS.__init__ on given pmfs)
// (one of the translation units of libbitswap_hip.so; shared device helpers: bitswap_dev.h; entry points: include/bitswap_hip.h)
#include "bitswap_dev.h"

// wavefronts per SIMD the spec 4 table kernels are compiled for (blocks of 8 bins: half the live tree registers of spec 3)
#ifndef BS_SPEC4_WAVES
#define BS_SPEC4_WAVES 5
#endif

namespace {

template <int NPL, typename PT, int MODE, int SPEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPEC == 4 ? BS_SPEC4_WAVES : SPEC >= 2 ? 4 : 7, 8))) void k_logistic(const double* __restrict__ endpoints, int64_t e_stride,
                                                  const double* __restrict__ step,
                                                  const PT* __restrict__ mu, const PT* __restrict__ scale,
                                                  const int32_t* __restrict__ sym, int B, int D, int bits,
                                                  int quantbits, int nb, uint32_t* __restrict__ out0,
                                                  uint32_t* __restrict__ out1, int64_t ld,
                                                  int32_t* __restrict__ status) {
#ifdef BS_TABLE_PRIO
    __builtin_amdgcn_s_setprio(BS_TABLE_PRIO);       // build-time experiment knob (visit r06y); default: priority 0
#endif
    constexpr int K = NPL * 64;
    constexpr bool UNI = SPEC >= 2;      // CDF specs 2 and 3: rows of uniform-width bins
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
        if (NPL > 1 && lane == 63) e[NPL - 1] = 0.0;   // the K-th, virtual endpoint (its cdf is 1 by definition): on the progression
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
        const bool dom = logistic_row<NPL, SPEC>(e, hstep, m_, rs, M, lane, bn);

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

template <int NPL, typename PT>
int launch_logistic(int mode, int spec, const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale,
                    const int32_t* sym, int B, int D, int bits, int quantbits, uint32_t* out0, uint32_t* out1,
                    int64_t ld, int32_t* status, hipStream_t st) {
    // chains per wavefront: amortise the endpoint fetch (8 KB of L2 reads per row at K = 1024: with one chain per wavefront the
    // launch is bound by them), but keep ~8 wavefronts per SIMD in flight.  Round 4, visit W (alone, us, z layer; nb = 1 / 4 / 16):
    // 13 chains 34.3 / 27.5 / - (encode flavour), 40.0 / 29.2 / - (rows); 500 chains - / 792 / 714 and - / 790 / 748 (pivot).
    // In the two-group pipeline of 1000 chains (same box, ms per step): nb = 4 / 8 / 16 = 180.1 / 178.3 / 183.7 -- sixteen rows per
    // wavefront make the launch's tail too coarse when it shares the chip.  Round 6, with the spec 4 kernel at five wavefronts per
    // SIMD (profiles/r06I_table_nb.txt, three repetitions): 1000 chains nb = 8 / 12 / 16 / 32 = 156.7 / 155.9 / 154.8 / 154.8 ms, 100 chains
    // nb = 2 / 4 / 8 / 16 = 23.3 / 22.6 / 22.2 / 22.0 -- sixteen; the rule below still shrinks it for few chains (13 chains: 4).
    int nb = 16;
    while (nb > 1 && (int64_t)((D + 3) / 4) * ((B + nb - 1) / nb) < 2048) nb >>= 1;
    const char* nb_env = getenv("BITSWAP_TABLE_NB");   // tuning only; read per launch (tools flip it between launches)
    if (nb_env && atoi(nb_env) > 0) nb = atoi(nb_env);
    dim3 grid((D + 3) / 4, (B + nb - 1) / nb), block(256);
    const PT* m = static_cast<const PT*>(mu);
    const PT* s = static_cast<const PT*>(scale);
#define BS_LAUNCH(MODE, SPEC)                                                                                     \
    hipLaunchKernelGGL((k_logistic<NPL, PT, MODE, SPEC>), grid, block, 0, st, endpoints, e_stride, step, m, s, sym, B, D, \
                       bits, quantbits, nb, out0, out1, ld, status)
    // specs 2 / 3 (uniform bins): host dispatch guarantees NPL >= 4; the layouts of the pop kernels need NPL >= 4 as well
    constexpr int S2 = NPL >= 4 ? 2 : 1, S3 = NPL >= 4 ? 3 : 1, S4 = NPL >= 4 ? 4 : 1;
    constexpr int MP = NPL >= 4 ? M_PIVOT : M_LINEAR, MW = NPL >= 4 ? M_WAVE : M_LINEAR, MV = NPL >= 4 ? M_LINEAR_VEC : M_LINEAR;
#define BS_BY_MODE(SPEC)                                   \
    do {                                                   \
        if (mode == M_PIVOT) BS_LAUNCH(MP, SPEC);          \
        else if (mode == M_WAVE) BS_LAUNCH(MW, SPEC);      \
        else if (mode == M_LINEAR_VEC) BS_LAUNCH(MV, SPEC); \
        else if (mode == M_ENCODE) BS_LAUNCH(M_ENCODE, SPEC); \
        else BS_LAUNCH(M_LINEAR, SPEC);                    \
    } while (0)
    if (spec == 4) BS_BY_MODE(S4);
    else if (spec == 3) BS_BY_MODE(S3);
    else if (spec == 2) BS_BY_MODE(S2);
    else if (mode == M_WAVE && NPL >= 4) BS_LAUNCH(MW, 1);
    else if (mode == M_LINEAR_VEC && NPL >= 4) BS_LAUNCH(MV, 1);
    else if (mode == M_ENCODE) BS_LAUNCH(M_ENCODE, 1);
    else BS_LAUNCH(M_LINEAR, 1);
#undef BS_BY_MODE
#undef BS_LAUNCH
    return launch_rc();
}

// spec: the CDF specification (1: any bins, step ignored; 2 / 3: uniform-width bins, step required, K >= 256)
template <typename PT>
int dispatch_logistic(int K, int mode, int spec, const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale,
                      const int32_t* sym, int B, int D, int bits, int quantbits, uint32_t* out0, uint32_t* out1,
                      int64_t ld, int32_t* status, hipStream_t st) {
#define BS_CASE(NPL)                                                                                             \
    case 64 * NPL:                                                                                               \
        return launch_logistic<NPL, PT>(mode, spec, endpoints, e_stride, step, mu, scale, sym, B, D, bits, quantbits, out0, \
                                        out1, ld, status, st)
    if (spec != 1 && K < 256) return BS_EUNSUPPORTED;  // CDF specs 2 / 3 are defined for K >= 256 (groups of K/64 >= 4 bins)
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

// cdf_spec argument of the C ABI -> 1 / 2 / 3 / 4, or 0 when it contradicts bin_step
inline int checked_spec(int cdf_spec, const double* bin_step) {
    if (cdf_spec == 1) return 1;                                  // bin_step is ignored
    if (cdf_spec >= 2 && cdf_spec <= 4 && bin_step) return cdf_spec;
    return 0;
}

}  // namespace

extern "C" {

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

int bs_logistic_tables(const double* endpoints, int64_t e_stride, const double* bin_step, int cdf_spec, const void* mu,
                       const void* scale, int param_dtype, int B, int D, int K, int bits, int quantbits,
                       uint32_t* cdf_out, int64_t ld, int layout, int32_t* status, void* stream) {
    const int spec = checked_spec(cdf_spec, bin_step);
    if (!endpoints || !mu || !scale || !cdf_out || B < 0 || D < 0 || bits < 1 || bits > 31 || quantbits < 0 ||
        quantbits >= bits || e_stride < 0 || !spec)
        return BS_EINVAL;
    if (spec == 1) bin_step = nullptr;
    int mode;
    if (layout == BS_LAYOUT_LINEAR) {
        if (ld < K + 1) return BS_EINVAL;
        mode = (aligned16(cdf_out) && (ld % 4 == 0)) ? M_LINEAR_VEC : M_LINEAR;
    } else if (layout == BS_LAYOUT_WAVE) {
        if (ld < K + 64 || K < 256) return BS_EINVAL;
        mode = M_WAVE;
    } else if (layout == BS_LAYOUT_PIVOT) {      // 64 x (cumulative value, aux) per row: uniform bins (specs 2 / 3) only
        if (ld < 128 || ld % 2 || K < 256 || !bin_step || (reinterpret_cast<uintptr_t>(cdf_out) & 7u)) return BS_EINVAL;
        mode = M_PIVOT;
    } else {
        return BS_EINVAL;
    }
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_logistic<float>(K, mode, spec, endpoints, e_stride, bin_step, mu, scale, nullptr, B, D, bits, quantbits,
                                        cdf_out, nullptr, ld, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_logistic<double>(K, mode, spec, endpoints, e_stride, bin_step, mu, scale, nullptr, B, D, bits, quantbits,
                                         cdf_out, nullptr, ld, status, S(stream));
    return BS_EINVAL;
}

int bs_logistic_fc(const double* endpoints, int64_t e_stride, const double* bin_step, int cdf_spec, const void* mu,
                   const void* scale, int param_dtype, const int32_t* sym, int B, int D, int K, int bits, int quantbits,
                   uint32_t* f_out, uint32_t* c_out, int32_t* status, void* stream) {
    const int spec = checked_spec(cdf_spec, bin_step);
    if (!endpoints || !mu || !scale || !sym || !f_out || !c_out || !status || B < 0 || D < 0 || bits < 1 ||
        bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || !spec)
        return BS_EINVAL;
    if (spec == 1) bin_step = nullptr;
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_logistic<float>(K, M_ENCODE, spec, endpoints, e_stride, bin_step, mu, scale, sym, B, D, bits, quantbits,
                                        f_out, c_out, 0, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_logistic<double>(K, M_ENCODE, spec, endpoints, e_stride, bin_step, mu, scale, sym, B, D, bits, quantbits,
                                         f_out, c_out, 0, status, S(stream));
    return BS_EINVAL;
}

}  // extern "C"
