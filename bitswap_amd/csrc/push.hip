// push.hip -- the serial push kernels of the reference stream format: k_rans_push / k_rans_push_table (64-lane systolic array)
// (one of the translation units of libbitswap_hip.so; shared device helpers: bitswap_dev.h; entry points: include/bitswap_hip.h)
#include "bitswap_dev.h"

namespace {

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
        auto systolic_step = [&](int t) {
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
        };
        if (cnt == 64) {                // whole chunks (every table of the codec): no loop counter, compare and branch per symbol
#pragma unroll
            for (int t = 0; t < 64; ++t) systolic_step(t);
        } else {
            for (int t = 0; t < cnt; ++t) systolic_step(t);
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

}  // namespace

extern "C" {

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

}  // extern "C"
