// lone_wave.hip -- what a single wavefront gets out of a CU: the shader clock it runs at (s_memtime against the 100 MHz
// s_memrealtime), dependent against independent issue, and whether independent instructions fill the slots of a dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lone_wave tools/probes/lone_wave.hip && /tmp/lone_wave [workgroups]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define ITERS 20000
#define REP8(X) X X X X X X X X

#define KERNEL(NAME, DECL, BODY, FIN)                                                    \
    __global__ __launch_bounds__(64) void NAME(uint64_t* out, int seed) {                \
        DECL                                                                             \
        const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();           \
        for (int i = 0; i < ITERS; ++i) { REP8(BODY) }                                   \
        const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();           \
        FIN                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; } \
    }

// 1 dependent SALU chain
KERNEL(k_s1, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(k) : "scc");, out[2 + threadIdx.x] = a;)
// 2 independent SALU chains, interleaved
KERNEL(k_s2, uint32_t a = seed; uint32_t b = seed + 2; uint32_t k = seed + 1;,
       asm volatile("s_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %2" : "+s"(a), "+s"(b) : "s"(k) : "scc");, out[2 + threadIdx.x] = a + b;)
// 4 independent SALU chains
KERNEL(k_s4, uint32_t a = seed; uint32_t b = seed + 2; uint32_t c = seed + 3; uint32_t d = seed + 4; uint32_t k = seed + 1;,
       asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : "s"(k) : "scc");,
       out[2 + threadIdx.x] = a + b + c + d;)
// 1 dependent VALU chain / 2 / 4 independent
KERNEL(k_v1, uint32_t a = seed; uint32_t k = seed + 1;, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(k));, out[2 + threadIdx.x] = a;)
KERNEL(k_v2, uint32_t a = seed; uint32_t b = seed + 2; uint32_t k = seed + 1;,
       asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(k));, out[2 + threadIdx.x] = a + b;)
KERNEL(k_v4, uint32_t a = seed; uint32_t b = seed + 2; uint32_t c = seed + 3; uint32_t d = seed + 4; uint32_t k = seed + 1;,
       asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));,
       out[2 + threadIdx.x] = a + b + c + d;)
// dependent SALU chain with one independent VALU between links: does the VALU ride for free?
KERNEL(k_s1v1, uint32_t a = seed; uint32_t b = seed + 2; uint32_t k = seed + 1;,
       asm volatile("s_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+s"(a), "+v"(b) : "s"(k) : "scc");, out[2 + threadIdx.x] = a + b;)
// the pop kernel's inner dependency: v_cmp -> s_bcnt1 -> v_readlane -> s_sub -> (back into a VALU compare operand)
KERNEL(k_search, uint32_t x = threadIdx.x * 3 + seed; uint32_t m = seed + 90; uint32_t c;,
       asm volatile("v_cmp_le_u32 vcc, %1, %2\n s_bcnt1_i32_b64 %0, vcc\n s_sub_u32 %0, %0, 1\n v_readlane_b32 %0, %1, %0\n s_add_u32 %2, %0, 60"
                    : "=&s"(c), "+v"(x), "+s"(m) : : "vcc", "scc");, out[2 + threadIdx.x] = m;)


// ---- hops between the vector and the scalar unit (each group is a closed dependent chain)
// SALU -> VALU operand -> readfirstlane -> SALU
KERNEL(k_h_sv, uint32_t a = seed; uint32_t v;, asm volatile("v_mov_b32 %1, %0\n v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1" : "+s"(a), "=&v"(v) : : "scc");, out[2 + threadIdx.x] = a;)
// v_cmp -> vcc -> s_bcnt1 -> (SGPR operand of the next v_cmp)
KERNEL(k_h_vcc, uint32_t x = threadIdx.x * 3 + seed; uint32_t m = seed + 90;,
       asm volatile("v_cmp_le_u32 vcc, %0, %1\n s_bcnt1_i32_b64 %1, vcc" : "+v"(x), "+s"(m) : : "vcc", "scc");, out[2 + threadIdx.x] = m;)
// the same through an SGPR pair instead of vcc
KERNEL(k_h_sgpr, uint32_t x = threadIdx.x * 3 + seed; uint32_t m = seed + 90; uint64_t bm;,
       asm volatile("v_cmp_le_u32 %2, %0, %1\n s_bcnt1_i32_b64 %1, %2" : "+v"(x), "+s"(m), "=&s"(bm) : : "scc");, out[2 + threadIdx.x] = m;)
// readlane with an SGPR lane select written by the SALU, result back into the SALU
KERNEL(k_h_rl, uint32_t x = threadIdx.x * 3 + seed; uint32_t l = 5; uint32_t r;,
       asm volatile("v_readlane_b32 %2, %0, %1\n s_and_b32 %1, %2, 63" : "+v"(x), "+s"(l), "=&s"(r) : : "scc");, out[2 + threadIdx.x] = l;)
// readlane through m0
KERNEL(k_h_rl_m0, uint32_t x = threadIdx.x * 3 + seed; uint32_t l = 5; uint32_t r;,
       asm volatile("s_mov_b32 m0, %1\n v_readlane_b32 %2, %0, m0\n s_and_b32 %1, %2, 63" : "+v"(x), "+s"(l), "=&s"(r) : : "scc", "m0");, out[2 + threadIdx.x] = l;)
// scalar-indexed register read: gpr index mode (gfx9 has no v_movrels)
KERNEL(k_h_idx, uint32_t x0 = seed; uint32_t x1 = seed + 1; uint32_t l = 1; uint32_t v;,
       asm volatile("s_set_gpr_idx_on %2, 1\n v_mov_b32 %3, %0\n s_set_gpr_idx_off\n v_readfirstlane_b32 %2, %3\n s_and_b32 %2, %2, 1" : "+v"(x0), "+v"(x1), "+s"(l), "=&v"(v) : : "scc", "m0");,
       out[2 + threadIdx.x] = l;)
// 64-bit head update as the compiler writes it, on the SALU (uniform values) and on the VALU (values pinned to vector registers)
KERNEL(k_h_head, uint64_t h = 0x123456789abcull + seed; uint32_t f = 77 + seed; uint32_t d = 5;,
       h = (uint64_t)f * (h >> 16) + d; asm volatile("" : "+s"(h));, out[2 + threadIdx.x] = h;)
KERNEL(k_h_headv, uint64_t h = 0x123456789abcull + seed; uint32_t f = 77 + seed; uint32_t d = 5; asm volatile("" : "+v"(f), "+v"(d));,
       h = (uint64_t)f * (h >> 16) + d; asm volatile("" : "+v"(h));, out[2 + threadIdx.x] = h;)

typedef void (*kern_t)(uint64_t*, int);

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 1;
    uint64_t* out;
    if (hipMalloc(&out, 80 * sizeof(uint64_t)) != hipSuccess) { printf("no device\n"); return 1; }
    struct { const char* name; kern_t k; int instr; } tests[] = {
        {"s_add dependent", k_s1, 1}, {"s_add 2 chains", k_s2, 2}, {"s_add 4 chains", k_s4, 4},
        {"v_add dependent", k_v1, 1}, {"v_add 2 chains", k_v2, 2}, {"v_add 4 chains", k_v4, 4},
        {"s_add dep + 1 indep v_add", k_s1v1, 2}, {"cmp/bcnt/sub/readlane/add (5 dep)", k_search, 5},
        {"v_mov(s) / readfirstlane / s_add", k_h_sv, 3}, {"v_cmp->vcc / s_bcnt1", k_h_vcc, 2}, {"v_cmp->sgpr pair / s_bcnt1", k_h_sgpr, 2},
        {"v_readlane(sgpr sel) / s_and", k_h_rl, 2}, {"s_mov m0 / v_readlane(m0) / s_and", k_h_rl_m0, 3},
        {"gpr_idx on / v_mov / off / rfl / s_and", k_h_idx, 5},
        {"head update on SALU (see ISA)", k_h_head, 1}, {"head update on VALU (see ISA)", k_h_headv, 1},
    };
    for (auto& t : tests) {
        uint64_t h[2];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(t.k, dim3(wgs), dim3(64), 0, 0, out, 1);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        const double groups = (double)ITERS * 8, ns = h[1] * 10.0 / groups;     // s_memrealtime: 100 MHz
        printf("%-36s %6.2f ns / group = %5.2f ns per instruction; s_memtime ticks per group %6.2f (tick rate %.0f MHz)\n", t.name, ns,
               ns / t.instr, h[0] / groups, h[0] / (h[1] * 10.0) * 1e3);
    }
    return 0;
}
