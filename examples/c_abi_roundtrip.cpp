// c_abi_roundtrip.cpp -- libbitswap_hip.so from plain C++/HIP, no Python, no torch: the boundary of this project is the
// C ABI of include/bitswap_hip.h.  Builds the integer tables of B chains x D latent dims from (mu, scale), pops D symbols
// per chain (bits back, ANS.decode, mnist_compress.py:58-68), pushes them again under the same model (ANS.encode, :49-56)
// and checks that every rANS state is exactly what it was -- first in the reference's single-state stream format
// (bs_logistic_tables + bs_rans_pop, bs_logistic_fc + bs_rans_push; whole rows and the pivot hand-off), then in the 64-state format (bs_layer_pop64 /
// bs_layer_push64).
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_roundtrip.cpp -L bitswap_amd/csrc -lbitswap_hip \
//         -Wl,-rpath,$PWD/bitswap_amd/csrc -o /tmp/c_abi_roundtrip && /tmp/c_abi_roundtrip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bitswap_hip.h"

#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } \
    } while (0)
#define BS_OKAY(x)                                                                       \
    do {                                                                                 \
        int r_ = (x);                                                                    \
        if (r_ != BS_OK) { std::printf("%s -> %s\n", #x, bs_strerror(r_)); std::exit(3); } \
    } while (0)

template <class T>
T* to_device(const std::vector<T>& h) {
    T* d = nullptr;
    HIP_OK(hipMalloc(&d, h.size() * sizeof(T)));
    HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <class T>
std::vector<T> to_host(const T* d, size_t n) {
    std::vector<T> h(n);
    HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
}

int main() {
    const int B = 6, D = 256, K = 1024, bits = 31, q = 10;
    const int64_t cap = 4096;
    std::printf("libbitswap_hip ABI %d, CDF spec up to %d\n", bs_abi_version(), bs_cdf_spec());

    // uniform-width bins per dim (what discretize() makes below the top layer) and their widths (CDF spec 2)
    std::vector<double> e((size_t)D * (K - 1)), step(D), centres((size_t)D * K);
    for (int d = 0; d < D; ++d) {
        const double lo = -6.0 - 0.01 * d, hi = 7.0 + 0.005 * d, h = (hi - lo) / K;
        for (int j = 1; j < K; ++j) e[(size_t)d * (K - 1) + j - 1] = lo + j * h;
        for (int j = 0; j < K; ++j) centres[(size_t)d * K + j] = lo + (j + 0.5) * h;
        step[d] = (e[(size_t)d * (K - 1) + K - 2] - e[(size_t)d * (K - 1)]) / (double)(K - 2);
    }
    std::vector<float> mu((size_t)B * D), sc((size_t)B * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : mu) v = 3.0f * (rnd() - 0.5f);
    for (auto& v : sc) v = 0.1f + 0.9f * rnd();

    double *d_e = to_device(e), *d_step = to_device(step), *d_cen = to_device(centres);
    float *d_mu = to_device(mu), *d_sc = to_device(sc);
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    // ---- reference stream format: one state per chain --------------------------------------------------------
    std::vector<uint64_t> head(B);
    std::vector<uint32_t> stack((size_t)B * cap);
    std::vector<int32_t> len(B, 2000), status(B, 0);
    for (auto& w : stack) w = 65536u + (uint32_t)(rnd() * 4.0e9f);
    for (int b = 0; b < B; ++b) head[b] = ((uint64_t)(65536u + (uint32_t)(rnd() * 4.0e9f))) << 32;
    uint64_t* d_head = to_device(head);
    uint32_t* d_stack = to_device(stack);
    int32_t *d_len = to_device(len), *d_status = to_device(status);
    const int64_t ld = K + 64;  // BS_LAYOUT_WAVE rows
    uint32_t *d_cdf, *d_f, *d_c;
    int32_t* d_sym;
    float* d_z;
    HIP_OK(hipMalloc(&d_cdf, (size_t)B * D * ld * 4));
    HIP_OK(hipMalloc(&d_f, (size_t)B * D * 4));
    HIP_OK(hipMalloc(&d_c, (size_t)B * D * 4));
    HIP_OK(hipMalloc(&d_sym, (size_t)B * D * 4));
    HIP_OK(hipMalloc(&d_z, (size_t)B * D * 4));

    BS_OKAY(bs_logistic_tables(d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, BS_PARAM_F32, B, D, K, bits, q, d_cdf, ld, BS_LAYOUT_WAVE,
                               d_status, stream));
    BS_OKAY(bs_rans_pop(d_head, d_stack, d_len, cap, d_cdf, (int64_t)D * ld, ld, BS_LAYOUT_WAVE, B, D, K, bits, d_sym, d_cen, K,
                        d_z, d_status, stream));
    auto len_after_pop = to_host(d_len, B);
    BS_OKAY(bs_logistic_fc(d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, BS_PARAM_F32, d_sym, B, D, K, bits, q, d_f, d_c, d_status, stream));
    BS_OKAY(bs_rans_push(d_head, d_stack, d_len, cap, d_f, d_c, B, D, bits, d_status, stream));
    HIP_OK(hipStreamSynchronize(stream));
    int bad = 0;
    auto head2 = to_host(d_head, B);
    auto len2 = to_host(d_len, B);
    auto st2 = to_host(d_status, B);
    auto stack2 = to_host(d_stack, (size_t)B * cap);
    auto sym = to_host(d_sym, (size_t)B * D);
    for (int b = 0; b < B; ++b) {
        bad += head2[b] != head[b] || len2[b] != len[b] || st2[b] != BS_ST_OK || len_after_pop[b] >= len[b];
        for (int i = 0; i < len[b]; ++i) bad += stack2[(size_t)b * cap + i] != stack[(size_t)b * cap + i];
    }
    std::printf("reference format: popped %d symbols per chain (first: %d %d %d), %d words taken, state %s\n", D, sym[0], sym[1],
                sym[2], len[0] - len_after_pop[0], bad ? "NOT restored" : "restored exactly");

    // ---- the same pop through the big-batch hand-off: 64 cumulative values per row, bs_rans_pop_pivot rebuilds the
    //      symbol's group of bins (BS_LAYOUT_PIVOT) -- same symbols, and the same push restores the state
    {
        const int64_t ldp = 128;
        uint32_t* d_piv;
        int32_t* d_sym2;
        HIP_OK(hipMalloc(&d_piv, (size_t)B * D * ldp * 4));
        HIP_OK(hipMalloc(&d_sym2, (size_t)B * D * 4));
        BS_OKAY(bs_logistic_tables(d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, BS_PARAM_F32, B, D, K, bits, q, d_piv, ldp, BS_LAYOUT_PIVOT,
                                   d_status, stream));
        BS_OKAY(bs_rans_pop_pivot(d_head, d_stack, d_len, cap, d_piv, ldp, d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, BS_PARAM_F32, B, D, K,
                                  bits, q, d_sym2, d_cen, K, d_z, d_status, stream));
        BS_OKAY(bs_logistic_fc(d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, BS_PARAM_F32, d_sym2, B, D, K, bits, q, d_f, d_c, d_status, stream));
        BS_OKAY(bs_rans_push(d_head, d_stack, d_len, cap, d_f, d_c, B, D, bits, d_status, stream));
        HIP_OK(hipStreamSynchronize(stream));
        auto sym2 = to_host(d_sym2, (size_t)B * D);
        auto head3 = to_host(d_head, B);
        auto len3 = to_host(d_len, B);
        int badp = 0;
        for (size_t i = 0; i < sym2.size(); ++i) badp += sym2[i] != sym[i];
        for (int b = 0; b < B; ++b) badp += head3[b] != head[b] || len3[b] != len[b];
        std::printf("pivot hand-off: %s\n", badp ? "symbols or state DIFFER" : "same symbols as whole rows, state restored exactly");
        bad += badp;
    }

    // ---- 64-state format: table + rANS step in one launch --------------------------------------------------
    const int64_t cap64 = 256;
    std::vector<uint64_t> head64((size_t)B * 64);
    std::vector<uint32_t> stack64((size_t)B * 64 * cap64);
    std::vector<int32_t> len64((size_t)B * 64, 100);
    for (auto& w : stack64) w = 65536u + (uint32_t)(rnd() * 4.0e9f);
    for (auto& h : head64) h = ((uint64_t)(65536u + (uint32_t)(rnd() * 4.0e9f))) << 32;
    uint64_t* d_head64 = to_device(head64);
    uint32_t* d_stack64 = to_device(stack64);
    int32_t* d_len64 = to_device(len64);
    HIP_OK(hipMemset(d_status, 0, B * 4));
    BS_OKAY(bs_layer_pop64(d_head64, d_stack64, d_len64, cap64, d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, D, BS_PARAM_F32, B, D, K, bits, q,
                           d_sym, d_cen, K, d_z, d_status, stream));
    BS_OKAY(bs_layer_push64(d_head64, d_stack64, d_len64, cap64, d_e, K - 1, d_step, BS_CDF_SPEC, d_mu, d_sc, D, BS_PARAM_F32, d_sym, B, D, K,
                            bits, q, d_status, stream));
    HIP_OK(hipStreamSynchronize(stream));
    auto h64 = to_host(d_head64, (size_t)B * 64);
    auto l64 = to_host(d_len64, (size_t)B * 64);
    auto s64 = to_host(d_stack64, (size_t)B * 64 * cap64);
    auto st64 = to_host(d_status, B);
    int bad64 = 0;
    for (size_t i = 0; i < h64.size(); ++i) {
        bad64 += h64[i] != head64[i] || l64[i] != len64[i];
        for (int k = 0; k < len64[i]; ++k) bad64 += s64[i * cap64 + k] != stack64[i * cap64 + k];
    }
    for (int b = 0; b < B; ++b) bad64 += st64[b] != BS_ST_OK;
    std::printf("64-state format: %d states %s\n", B * 64, bad64 ? "NOT restored" : "restored exactly");
    std::printf(bad || bad64 ? "C_ABI_ROUNDTRIP_FAILED\n" : "C_ABI_ROUNDTRIP_OK\n");
    return bad || bad64 ? 1 : 0;
}
