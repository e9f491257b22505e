// pop.hip -- the serial pop kernels of the reference stream format: k_rans_pop_wave (BS_LAYOUT_WAVE rows), k_rans_pop_pivot (BS_LAYOUT_PIVOT: rebuilds one group of bins), k_rans_pop / _generic (linear rows)
// (one of the translation units of libbitswap_hip.so; shared device helpers: bitswap_dev.h; entry points: include/bitswap_hip.h)
#include "bitswap_dev.h"

namespace {

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
// k_rans_pop_pivot: BS_LAYOUT_PIVOT rows (uniform-width bins, CDF spec 2 or 3), one wavefront per chain.
//
// The table kernel hands over 64 cumulative values per row (one per group of NPL bins) plus which bin took the remnant
// and how much.  Per symbol: ballot(pivot <= m) names the group L; lanes 0 .. NPL-1 rebuild the cdf of its NPL bins, each with
// exactly the operations logistic_row spends on that bin (own anchor exponential, own geometric factor, residual of the
// stored endpoint; spec 2: a correctly rounded reciprocal per bin, spec 3: the block's product tree built by a butterfly
// exchange over its lanes -- IEEE multiplication commutes, so every lane holds the node values the table kernel computed --
// ONE reciprocal of the root and one multiplication per level back down) -- the truncated
// differences of bins 1 .. NPL-1 are therefore the table's; the frequency of the group's FIRST bin, whose pmf is differenced
// against the last cdf of the group below, is what the integers leave: pivot[L+1] - pivot[L] - sum of the others (the remnant
// bump included), so the group below is never evaluated.  A scan on top of the pivot gives c_s and f_s.  About five times the
// instructions of k_rans_pop_wave per symbol (205 against 38), but 512 B of HBM traffic per row instead of 4352 B: at 400 chains the
// row-reading pop kernel ran at the HBM roof (3.57 GB per launch in 0.57 ms) and nothing overlapped with it
// (profiles/r03m_overlap2.txt); this one touches the L2-resident endpoint table and little else.
// Endpoints of the group are fetched AFTER the group is known (data dependent) and consumed after the two exponentials
// that do not need them; pivots and the 64 anchor endpoints of a row are prefetched PF rows ahead like the rows of
// k_rans_pop_wave; (mu, 1 / scale, bin width, their product) of a 64-symbol chunk wait in LDS, one broadcast read per symbol.
// ------------------------------------------------------------------------------------------
// lane i <- lane i-1 of the whole wavefront (DPP wave_shr:1; lane 0 has no source: its result is not used)
__device__ __forceinline__ double wave_shr1_f64(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)u, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// 64-bit lane exchange by DPP: two 32-bit moves (every lane of a row has a source, so no `old` value is needed -- with
// update_dpp(old = v, ...) hipcc copied v first: four instructions per exchange instead of two)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)u, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// the two half-waves of v, each spread over the whole wavefront (gfx950 v_permlane32_swap: no LDS round trip as __shfl's
// ds_bpermute has): lo = lane k & 31 of v, up = lane 32 + (k & 31) of v
__device__ __forceinline__ void split_halves_f64(double v, double& lo, double& up) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const auto a = __builtin_amdgcn_permlane32_swap((uint32_t)u, (uint32_t)u, false, false);
    const auto c = __builtin_amdgcn_permlane32_swap((uint32_t)(u >> 32), (uint32_t)(u >> 32), false, false);
    lo = __longlong_as_double((long long)(((uint64_t)c[0] << 32) | a[0]));
    up = __longlong_as_double((long long)(((uint64_t)c[1] << 32) | a[1]));
}
// 1 / x of every lane by CDF spec 3's product tree over aligned blocks of N <= 16 lanes (bitswap_dev.h::tree_inverse, one bin
// per lane): level k pairs the node of lanes [2^k j', 2^k (j'+1)) with its sibling -- quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror: any lane of the sibling node holds the sibling's value -- and multiplies; the root is inverted
// once, and on the way down a lane multiplies by the sibling values it met on the way up.
template <int N>
__device__ __forceinline__ double tree_inverse_lanes(double x) {
    static_assert(N == 4 || N == 8 || N == 16, "one block of 4, 8 or 16 lanes");
    const double w0 = dpp_f64<0xB1>(x);
    double v = x * w0;
    const double w1 = dpp_f64<0x4E>(v);
    v = v * w1;
    double w2 = 1.0, w3 = 1.0;
    if (N >= 8) { w2 = dpp_f64<0x141>(v); v = v * w2; }
    if (N >= 16) { w3 = dpp_f64<0x140>(v); v = v * w3; }
    double inv = recip_1_to_huge(v);
    if (N >= 16) inv = inv * w3;
    if (N >= 8) inv = inv * w2;
    inv = inv * w1;
    return inv * w0;
}

template <int NPL, typename PT, int PF, int SPEC>
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
    __shared__ double4 sh_prm[64];           // (mu, 1/scale, h, h/scale) of the chunk's 64 rows: one broadcast read per symbol
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
    // this lane's role in the rebuild.  Lanes 0 .. NPL-1: bin `bi` of the symbol's group L (the other lanes repeat them, unused).
    // With ONE_EXP the upper half-wave evaluates the geometric factors Q_b = exp(-b h/scale) in the same instructions in
    // which the lower half evaluates the anchors exp(-t_a); lane 32 + k serves lane k (NPL divides 32: same bin index).
    const bool is_rest = lane >= 1 && lane < NPL;      // bins 1 .. NPL-1: frequencies from the cdf; bin 0: from the pivots
    constexpr unsigned long long REST_MASK = ((NPL >= 64 ? 0ull : (1ull << NPL)) - 1ull) & ~1ull;
    const int bi = lane & (NPL - 1);
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
        // spec 3: which rows of the chunk take the batch inversion (logistic_row's test, one bit per row)
        const unsigned long long batch_rows = SPEC >= 3 ? __ballot((double)NPL * fabs(hr_l) < BS_SPEC3_FAST_HR) : 0ull;
        __syncthreads();                      // one wavefront per block: orders the LDS accesses, costs nothing
        sh_prm[lane] = make_double4(mu_l, rs_l, h_l, hr_l);
        __syncthreads();
        int o = 0;
        uint32_t mysym = 0;
        for (int g = 64 / PF - 1; g >= 0; --g) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int dk = d & 63;
                const uint32_t m = (uint32_t)h & mask;
                const int L = __popcll(__ballot(pv[u].x <= m)) - 1;          // group of the symbol: 0..63 (c_0 = 0 <= m)
                const int j = L * NPL + bi;                                  // this lane's bin
                // its upper endpoint: data dependent, requested first, used last
                const double e_j = erow[min(j, K - 2)];
                erow -= d > 0 ? e_stride : 0;
                const double4 prm4 = sh_prm[dk];                             // every lane the same address: a broadcast
                const double m_ = prm4.x, rs = prm4.y, hstep = prm4.z, hr = prm4.w;
                const double e_a = readlane_f64(anc[u], L);
                const uint32_t piv_L = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].x, L);
                const uint32_t piv_up = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].x, min(L + 1, 63));
                const uint32_t piv_N = L == 63 ? (1u << bits) : piv_up;      // cumulative value behind the group
                const uint32_t bumped = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].y, 0);
                const uint32_t rem = (uint32_t)__builtin_amdgcn_readlane((int)pv[u].y, 1);
                // refill: the row PF steps ahead
                pv[u] = *pp;
                anc[u] = *ap;
                if (dl > 0) { pp -= ld2; ap -= e_stride; }
                --dl;
                // logistic_row, one bin per lane
                const bool batch = SPEC >= 3 && ((batch_rows >> dk) & 1ull);   // wave-uniform
                const double hi = batch ? BS_SPEC3_ANCHOR_HI : 700.0;
                double A, Q;
                if (ONE_EXP) {
                    const double x = det_exp_hi(q_lane ? -((double)bi * hr) : -((e_a - m_) * rs), hi);
                    split_halves_f64(x, A, Q);                               // lane k < 32: its own A, the Q of lane 32 + k
                } else {
                    A = det_exp_hi(-((e_a - m_) * rs), hi);
                    Q = det_exp(-((double)bi * hr));
                }
                double r = e_j - fma((double)bi, hstep, e_a);
                double c;
                if (batch) {
                    if (j == K - 1) r = 0.0;                                 // the K-th, virtual endpoint: on the progression
                    const double Ars = A * rs;
                    const double x = fma(Q, fma(-Ars, r, A), 1.0);           // bi == 0: Q = 1, r = 0 -> 1 + A
                    c = tree_inverse_lanes<SpecBlock<SPEC, NPL>::N>(x);       // spec 4: blocks of 8 lanes ...
                    if constexpr (SPEC == 4) c = newton_correct(x, c);         // ... and the residual correction of logistic_row
                } else {
                    const double eps = r * rs;
                    const double uu = fma(-A, eps, A);
                    c = recip_1_to_huge(fma(Q, uu, 1.0));
                }
                if (j == K - 1) c = 1.0;                                     // the last bin has no upper endpoint
                // bins 1 .. NPL-1: pmf = cdf - cdf of the lane below (mnist_compress.py:184), frequency as ANS.__init__ has it
                const double below = wave_shr1_f64(c);
                uint32_t f = trunc_u32((c - below) * M) + 1u;
                if ((uint32_t)j == bumped) f += rem;
                if (!is_rest) f = 0u;
                uint32_t incl = f;                                           // inclusive scan over the NPL bins
                incl += dpp_or0<0x111, 0xf>(incl);
                incl += dpp_or0<0x112, 0xf>(incl);
                if (NPL > 4) incl += dpp_or0<0x114, 0xf>(incl);
                if (NPL > 8) incl += dpp_or0<0x118, 0xf>(incl);
                if (NPL > 16) incl += dpp_or0<0x142, 0xa>(incl);
                // bin 0: what the integers leave (the group's total is pivot[L+1] - pivot[L], remnant bump included)
                const uint32_t rest = (uint32_t)__builtin_amdgcn_readlane((int)incl, NPL - 1);
                const uint32_t f0 = piv_N - piv_L - rest;
                const uint32_t cst = (piv_L + f0) + (incl - f);              // c of this lane's bin (lanes 1 .. NPL-1)
                const int pos = __popcll(__ballot(cst <= m) & REST_MASK) + 1;   // 1..NPL (bin 0 starts at pivot[L] <= m)
                const uint32_t cs_r = (uint32_t)__builtin_amdgcn_readlane((int)cst, pos - 1);
                const uint32_t fs_r = (uint32_t)__builtin_amdgcn_readlane((int)f, pos - 1);
                const uint32_t cs = pos == 1 ? piv_L : cs_r;
                const uint32_t fs = pos == 1 ? f0 : fs_r;
                mysym = (lane == dk) ? (uint32_t)(L * NPL + pos - 1) : mysym;
#ifdef BS_POP_PAD     /* sensitivity experiment (round 6): BS_POP_PAD extra float64 VALU instructions per symbol, results unused */
                {
                    double pad_ = c;
#pragma unroll
                    for (int pi_ = 0; pi_ < BS_POP_PAD; ++pi_) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(pad_));
                }
#endif
                h = (uint64_t)fs * (h >> bits) + (uint64_t)(m - cs);
                uint32_t hhi = (uint32_t)(h >> 32);
#if !__has_feature(address_sanitizer)
                asm("" : "+s"(hhi));              // keep this a 32-bit scalar compare (hipcc otherwise builds a 64-bit VALU one)
#endif
                if (hhi == 0u) {  // h < 2^32, mnist_compress.py:65
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

}  // namespace

template <typename PT>
int dispatch_pop_pivot(int spec, uint64_t* head, uint32_t* stack, int32_t* len, int64_t cap, const uint32_t* piv, int64_t ld,
                       const double* endpoints, int64_t e_stride, const double* step, const void* mu, const void* scale, int B,
                       int D, int K, int bits, int quantbits, int32_t* sym_out, const double* centres, int64_t c_stride,
                       float* centre_out, int32_t* status, hipStream_t st) {
    dim3 grid(B), block(64);
    const PT* m = static_cast<const PT*>(mu);
    const PT* s = static_cast<const PT*>(scale);
#define BS_POPP(NPL, PF)                                                                                              \
    do {                                                                                                              \
        if (spec == 4)                                                                                                \
            hipLaunchKernelGGL((k_rans_pop_pivot<NPL, PT, PF, 4>), grid, block, (size_t)D * 4, st, head, stack, len, cap, piv, ld, \
                               endpoints, e_stride, step, m, s, D, bits, quantbits, sym_out, centres, c_stride, centre_out, status); \
        else if (spec == 3)                                                                                           \
            hipLaunchKernelGGL((k_rans_pop_pivot<NPL, PT, PF, 3>), grid, block, (size_t)D * 4, st, head, stack, len, cap, piv, ld, \
                               endpoints, e_stride, step, m, s, D, bits, quantbits, sym_out, centres, c_stride, centre_out, status); \
        else                                                                                                          \
            hipLaunchKernelGGL((k_rans_pop_pivot<NPL, PT, PF, 2>), grid, block, (size_t)D * 4, st, head, stack, len, cap, piv, ld, \
                               endpoints, e_stride, step, m, s, D, bits, quantbits, sym_out, centres, c_stride, centre_out, status); \
    } while (0)
    if (K == 256) BS_POPP(4, BS_POP_PF);
    else if (K == 512) BS_POPP(8, BS_POP_PF);
    else if (K == 1024) BS_POPP(16, BS_POP_PF);
    else if (K == 2048) BS_POPP(32, BS_POP_PF);
    else return BS_EUNSUPPORTED;
#undef BS_POPP
    return launch_rc();
}

extern "C" {

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
                      const double* endpoints, int64_t e_stride, const double* bin_step, int cdf_spec, const void* mu,
                      const void* scale, int param_dtype, int B, int D, int K, int bits, int quantbits, int32_t* sym_out,
                      const double* centres, int64_t c_stride, float* centre_out, int32_t* status, void* stream) {
    if (cdf_spec < 2 || cdf_spec > 4 || !head || !stack || !len || !pivots || !endpoints || !bin_step || !mu || !scale || !sym_out || !status || B < 0 ||
        D < 0 || cap < 0 || bits < 1 || bits > 31 || quantbits < 0 || quantbits >= bits || e_stride < 0 || c_stride < 0 ||
        (centres && !centre_out) || ld < 128 || ld % 2 || (reinterpret_cast<uintptr_t>(pivots) & 7u))
        return BS_EINVAL;
    // whole 64-symbol chunks; a chain's symbols (D x 4 B, dynamic) + sh_prm (2 KB, static) fit the 64 KB a launch gets
    // without hipFuncAttributeMaxDynamicSharedMemorySize
    static_assert(BS_POP_PIVOT_MAX_D % 64 == 0 && BS_POP_PIVOT_MAX_D * 4 + 64 * 32 <= 64 * 1024, "pivot pop LDS budget");
    if (D % 64 != 0 || D > BS_POP_PIVOT_MAX_D) return BS_EUNSUPPORTED;
    if (B == 0 || D == 0) return BS_OK;
    if (param_dtype == BS_PARAM_F32)
        return dispatch_pop_pivot<float>(cdf_spec, head, stack, len, cap, pivots, ld, endpoints, e_stride, bin_step, mu, scale, B, D, K,
                                         bits, quantbits, sym_out, centres, c_stride, centre_out, status, S(stream));
    if (param_dtype == BS_PARAM_F64)
        return dispatch_pop_pivot<double>(cdf_spec, head, stack, len, cap, pivots, ld, endpoints, e_stride, bin_step, mu, scale, B, D, K,
                                          bits, quantbits, sym_out, centres, c_stride, centre_out, status, S(stream));
    return BS_EINVAL;
}

}  // extern "C"
