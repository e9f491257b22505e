// instr_rate.hip -- issue rate of the float64 / conversion instructions the logistic kernel is built from.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/instr_rate tools/probes/instr_rate.hip && /tmp/instr_rate
// One wave-instruction = 64 lanes.  Reported: cycles per wave-instruction per SIMD with 8 waves/SIMD resident
// (throughput, not latency), measured with s_memtime-free wall clock: ops / (time * SIMDs * clock).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITERS 4096

#define KERNEL(NAME, BODY)                                                                     \
    __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                    \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
               a6 = a0 + 6, a7 = a0 + 7;                                                       \
        const double k1 = seed * 0.5, k2 = seed * 0.25;                                        \
        for (int i = 0; i < ITERS; ++i) {                                                      \
            BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7)            \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;   \
    }

#define B_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
#define B_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(k1));
#define B_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(k1));
#define B_MIN(x) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x) : "v"(k1));
#define B_RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define B_RND(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define B_LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));
#define B_CVTI(x) { int t_; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t_) : "v"(x)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(t_)); }
#define B_CVTF(x) { float t_; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t_) : "v"(x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(t_)); }
#define B_RCPF(x) { float t_ = (float)threadIdx.x; asm volatile("v_rcp_f32 %0, %0" : "+v"(t_)); asm volatile("v_rcp_f32 %0, %0" : "+v"(t_)); x += t_; }
#define B_ADDU(x) { uint32_t lo_ = (uint32_t)__double_as_longlong(x); asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo_) : "v"(i)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo_) : "v"(i)); x = __longlong_as_double((long long)lo_ | 0x3ff0000000000000ll); }
#define B_FMAF(x) { float t_ = (float)x; asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t_) : "v"(t_)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t_) : "v"(t_)); x = t_; }

KERNEL(k_fma, B_FMA)
KERNEL(k_mul, B_MUL)
KERNEL(k_add, B_ADD)
KERNEL(k_min, B_MIN)
KERNEL(k_rcp, B_RCP)
KERNEL(k_rnd, B_RND)
KERNEL(k_ldexp, B_LDEXP)
KERNEL(k_cvti, B_CVTI)
KERNEL(k_cvtf, B_CVTF)

typedef void (*kern_t)(double*, double);

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;  // Hz
    printf("device %s: %d CUs, clock %.0f MHz\n", p.name, cus, clk / 1e6);
    const int blocks = cus * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
    double* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    struct { const char* name; kern_t k; int per_body; } tests[] = {
        {"v_fma_f64", k_fma, 1}, {"v_mul_f64", k_mul, 1}, {"v_add_f64", k_add, 1}, {"v_min_f64", k_min, 1},
        {"v_rcp_f64", k_rcp, 1}, {"v_rndne_f64", k_rnd, 1}, {"v_ldexp_f64", k_ldexp, 1},
        {"v_cvt_i32_f64+v_cvt_f64_i32", k_cvti, 2}, {"v_cvt_f32_f64+v_cvt_f64_f32", k_cvtf, 2},
    };
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double wave_instr = (double)blocks * 4 * ITERS * 8 * t.per_body;
        const double per_simd = wave_instr / (cus * 4.0);
        const double cyc = ms * 1e-3 * clk / per_simd;
        printf("%-32s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at nominal clock)  %.1f Ginstr/s chip\n", t.name, ms, cyc,
               wave_instr / (ms * 1e-3) / 1e9);
    }
    return 0;
}
