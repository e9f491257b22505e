// instr_latency.hip -- dependent-issue latency of the instructions the serial rANS kernels chain together.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/instr_latency tools/probes/instr_latency.hip && /tmp/instr_latency
// ONE wavefront on the whole GPU, each test a chain of N dependent copies of one instruction (or one
// VALU<->SALU round trip); reported in ns per link and in cycles at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 20000
#define REP8(X) X X X X X X X X

#define KERNEL(NAME, DECL, BODY, FIN)                                        \
    __global__ __launch_bounds__(64) void NAME(double* out, int seed) {      \
        DECL                                                                 \
        for (int i = 0; i < ITERS; ++i) { REP8(BODY) }                       \
        FIN                                                                  \
    }

KERNEL(l_fma64, double a = seed; double k = seed * 0.5;, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(k));, out[threadIdx.x] = a;)
KERNEL(l_add64, double a = seed; double k = seed * 0.5;, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(k));, out[threadIdx.x] = a;)
KERNEL(l_cvt, double a = seed; uint32_t t;, asm volatile("v_cvt_u32_f64 %0, %1\n v_cvt_f64_u32 %1, %0" : "=&v"(t), "+v"(a));, out[threadIdx.x] = a;)
KERNEL(l_addu32, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(k));, out[threadIdx.x] = a;)
KERNEL(l_mullo, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(k));, out[threadIdx.x] = a;)
KERNEL(l_sadd, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(k) : "scc");, out[threadIdx.x] = a;)
KERNEL(l_smul, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("s_mul_i32 %0, %0, %1" : "+s"(a) : "s"(k));, out[threadIdx.x] = a;)
KERNEL(l_scsel, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("s_cmp_ge_u32 %0, %1\n s_cselect_b32 %0, %0, %1" : "+s"(a) : "s"(k) : "scc");, out[threadIdx.x] = a;)
// VALU -> SGPR -> VALU round trip
KERNEL(l_rfl, uint32_t a = seed; uint32_t s;, asm volatile("v_readfirstlane_b32 %0, %1\n s_nop 0\n v_mov_b32 %1, %0" : "=&s"(s), "+v"(a));, out[threadIdx.x] = a;)
// VALU -> SGPR -> SALU -> VALU
KERNEL(l_rfl_salu, uint32_t a = seed; uint32_t s;, asm volatile("v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n v_mov_b32 %1, %0" : "=&s"(s), "+v"(a) : : "scc");, out[threadIdx.x] = a;)
// v_cmp -> vcc -> s_bcnt1 -> v_mov
KERNEL(l_ballot, uint32_t a = seed; uint32_t s;, asm volatile("v_cmp_ge_u32 vcc, %1, %1\n s_bcnt1_i32_b64 %0, vcc\n v_mov_b32 %1, %0" : "=&s"(s), "+v"(a) : : "vcc", "scc");, out[threadIdx.x] = a;)
// readlane with SGPR lane select
KERNEL(l_readlane, uint32_t a = seed; uint32_t s = 3;, asm volatile("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n v_mov_b32 %1, %0" : "+s"(s), "+v"(a) : : "scc");, out[threadIdx.x] = a;)

typedef void (*kern_t)(double*, int);

int main() {
    double* out;
    if (hipMalloc(&out, 64 * sizeof(double)) != hipSuccess) { printf("no device\n"); return 1; }
    struct { const char* name; kern_t k; int links; } tests[] = {
        {"v_fma_f64 (dependent)", l_fma64, 1}, {"v_add_f64", l_add64, 1}, {"v_cvt_u32_f64 + v_cvt_f64_u32", l_cvt, 2},
        {"v_add_u32", l_addu32, 1}, {"v_mul_lo_u32", l_mullo, 1}, {"s_add_u32", l_sadd, 1}, {"s_mul_i32", l_smul, 1},
        {"s_cmp + s_cselect", l_scsel, 2}, {"v_readfirstlane + s_nop + v_mov", l_rfl, 2},
        {"v_readfirstlane + s_add + v_mov", l_rfl_salu, 3}, {"v_cmp + s_bcnt1 + v_mov", l_ballot, 3},
        {"v_readlane(sgpr) + s_and + v_mov", l_readlane, 3},
    };
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, out, 1);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, out, 1);
        (void)hipEventRecord(b);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        const double groups = (double)ITERS * 8;
        const double ns = ms * 1e6 / groups;
        printf("%-36s %7.2f ns per group of %d  = %6.1f cycles @2.4GHz  (%5.1f per instruction)\n", t.name, ns, t.links,
               ns * 2.4, ns * 2.4 / t.links);
    }
    return 0;
}
