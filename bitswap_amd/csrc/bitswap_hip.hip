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
//
// Since round 4 the kernels live in one translation unit each -- tables.hip (k_logistic, k_table_rows), pop.hip, push.hip,
// layer64.hip -- around the shared device helpers of bitswap_dev.h; this file keeps the small entry points (version, error
// strings, centre gather, the sigmoid probe and the device self-test).
#include "bitswap_dev.h"

namespace {

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

}  // namespace

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

int bs_stream_create_cu_mask(int first_cu, int n_cus, void** stream_out) {
    // mask bit b = compute unit b / nxcd of XCD b % nxcd (the driver deals the bits of a queue's mask round the XCDs), so a run
    // of n_cus = 8 m consecutive bits is m compute units on each of the 8 XCDs
    if (!stream_out || first_cu < 0 || n_cus < 1 || first_cu + n_cus > 1024) return BS_EINVAL;
    uint32_t mask[32] = {};
    for (int b = first_cu; b < first_cu + n_cus; ++b) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)((first_cu + n_cus + 31) / 32), mask) != hipSuccess) {
        (void)hipGetLastError();
        return BS_ELAUNCH;
    }
    *stream_out = st;
    return BS_OK;
}

int bs_stream_destroy(void* stream) {
    if (!stream) return BS_EINVAL;
    return hipStreamDestroy(S(stream)) == hipSuccess ? BS_OK : BS_ELAUNCH;
}

// where the wavefronts of a launch on `stream` land: ids_out[i] = HW_ID of workgroup i's first wavefront (bits 8..11 CU,
// 12 SH, 13..15 SE) | XCC_ID << 16, n workgroups of one wavefront that spin ~spin_cycles so that the launch spreads
__global__ void k_where(uint32_t* ids, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));       // XCC_ID
    const uint64_t t0 = __builtin_readcyclecounter();
    while ((int64_t)(__builtin_readcyclecounter() - t0) < spin) {}
    if (threadIdx.x == 0) ids[blockIdx.x] = (hw & 0xffffu) | ((xcc & 0xfu) << 16);
}
int bs_debug_where(uint32_t* ids_out, int n, int spin_cycles, void* stream) {
    if (!ids_out || n < 1) return BS_EINVAL;
    hipLaunchKernelGGL(k_where, dim3((unsigned)n), dim3(64), 0, S(stream), ids_out, spin_cycles);
    return launch_rc();
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
