// recip_check.hip -- is a cheaper Newton schedule still the correctly rounded reciprocal?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/recip_check tools/probes/recip_check.hip && /tmp/recip_check
// Compares, over 2^34 arguments x = 1 + e (the only arguments det_sigmoid feeds it: x in [1, 2^1011)),
//   A: v_rcp_f64 + 2 quadratic Newton steps + residual correction   (the shipped sequence)
//   B: v_rcp_f64 + 1 cubic step + residual correction               (one fma less)
// against IEEE division 1.0 / x, and reports the worst relative error of the hardware seed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ double recip_a(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    const double r = fma(-x, y, 1.0);
    return fma(r, y, y);
}

__device__ __forceinline__ double recip_b(double x) {
    double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    const double t = fma(e, e, e);
    y = fma(y, t, y);
    const double r = fma(-x, y, 1.0);
    return fma(r, y, y);
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ void k_check(uint64_t seed, int per_thread, unsigned long long* bad_a, unsigned long long* bad_b,
                        unsigned long long* seed_err_bits) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long na = 0, nb = 0;
    double worst = 0.0;
    for (int i = 0; i < per_thread; ++i) {
        const uint64_t r = mix(seed + tid * (uint64_t)per_thread + i);
        double x;
        const int kind = (int)(r & 7);
        uint64_t mant = (r >> 12) & 0xfffffffffffffull;
        if (kind == 0) mant |= 0xfffffff000000ull;            // long runs of ones: Markstein's hard cases
        if (kind == 1) mant &= 0x0000000000fffull;            // just above a power of two
        if (kind == 2) mant = 0xfffffffffffffull - (mant & 0xff);
        int ex;
        if (kind < 5) ex = 1023 + (int)((r >> 3) % 1010);     // [1, 2^1010)
        else ex = 1023 + (int)((r >> 3) % 60);                // the range sigmoid arguments near 0 produce
        x = __longlong_as_double((long long)(((uint64_t)ex << 52) | mant));
        // 1 + e style values: a few mantissa bits only
        if (kind == 7) x = 1.0 + __longlong_as_double((long long)(((uint64_t)(1023 - (int)((r >> 3) % 60)) << 52) | mant));
        const double q = 1.0 / x;
        if (recip_a(x) != q) ++na;
        if (recip_b(x) != q) ++nb;
        const double s = __builtin_amdgcn_rcp(x);
        const double rel = fabs(fma(-x, s, 1.0));
        worst = fmax(worst, rel);
    }
    if (na) atomicAdd(bad_a, na);
    if (nb) atomicAdd(bad_b, nb);
    atomicMax(seed_err_bits, (unsigned long long)__double_as_longlong(worst));
}

int main() {
    unsigned long long* d;
    if (hipMalloc(&d, 3 * sizeof(unsigned long long)) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMemset(d, 0, 3 * sizeof(unsigned long long));
    const int blocks = 256 * 64, threads = 256, per_thread = 4096;
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 12345ull, per_thread, d, d + 1, d + 2);
    unsigned long long h[3];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double worst;
    __builtin_memcpy(&worst, &h[2], 8);
    printf("arguments %.3g  mismatches vs 1.0/x: A(2 quadratic) %llu  B(1 cubic) %llu  worst seed rel err %.3g (2^%.1f)\n",
           (double)blocks * threads * per_thread, h[0], h[1], worst, log2(worst));
    return (h[0] || h[1]) ? 2 : 0;
}
