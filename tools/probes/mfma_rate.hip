// mfma_rate.hip -- what the bf16 matrix pipe of an MI355X SUSTAINS under its power budget, by operand content.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/probes/mfma_rate.hip && /tmp/mfma_rate
// Why (round 6): the bf16x3 product of the conv GEMM is priced against 2.5 PFLOP/s (6 limb products = 2.67x the
// fp32 MFMA rate).  MI355X_MICROARCH.md says the chip clocks to its power budget (a tuned bf16 kernel: 1,247 TFLOP/s on random
// data at ~1.9 GHz, 1,483 on zeros).  This probe is the kernel-free ceiling: NOTHING but v_mfma_f32_32x32x16_bf16 in the loop --
// 48 per iteration on 8 accumulator tiles, the multiply pattern of k_wino_gemm_bf16x3 (3 A limbs x 4 B tiles x 3 limbs, the six
// products of NPROD = 6), operands resident in registers -- with 1 or 2 wavefronts per SIMD, long launches back to back so the
// clock settles.  Operand sets: zeros; random bf16 (full-scale mantissas); "limbs" = the three bf16 limbs of random float32
// (what the product kernel multiplies: leading limb full-scale, the others 2^-8 / 2^-16 of it, same mantissa entropy).
// Also v_mfma_f32_32x32x2_f32 (the fp32 route's instruction) the same way.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
union Pack8 { u32x4 u; bf16x8 b; };

__device__ constexpr int ORDER6[6][2] = {{0, 2}, {1, 1}, {2, 0}, {0, 1}, {1, 0}, {0, 0}};

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_bf16(const u32x4* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2][3], b[4][3];
    const u32x4* p = ops + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) % 64 * 18 * 64;
#pragma unroll
    for (int i = 0; i < 6; ++i) { Pack8 q; q.u = p[i * 64 + lane]; a[i / 3][i % 3] = q.b; }
#pragma unroll
    for (int i = 0; i < 12; ++i) { Pack8 q; q.u = p[(6 + i) * 64 + lane]; b[i / 3][i % 3] = q.b; }
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][ORDER6[pr][0]], b[ni][ORDER6[pr][1]], acc[mi][ni], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) s += acc[mi][ni][v];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// the same multiplies with only NACC accumulator tiles in rotation: consecutive MFMAs on the same accumulator are NACC issues apart
// (the wave-specialised GEMM of round 6 first rotated 2: is a dependent v_mfma_f32_32x32x16_bf16 ready after 64 cycles?)
template <int NACC>
__global__ __launch_bounds__(256, 1) void k_bf16_rot(const u32x4* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2][3], b[4][3];
    const u32x4* p = ops + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) % 64 * 18 * 64;
#pragma unroll
    for (int i = 0; i < 6; ++i) { Pack8 q; q.u = p[i * 64 + lane]; a[i / 3][i % 3] = q.b; }
#pragma unroll
    for (int i = 0; i < 12; ++i) { Pack8 q; q.u = p[(6 + i) * 64 + lane]; b[i / 3][i % 3] = q.b; }
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[k][v] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m)
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 1][(m >> 1) % 3], b[(m >> 2) & 3][(m >> 4) % 3], acc[m % NACC], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[k][v];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_f32(const float* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    float a[2][8], b[4][8];
    const float* p = ops + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % 64) * 48 * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i / 8][i % 8] = p[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i / 8][i % 8] = p[(16 + i) * 64 + lane];
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][k], b[ni][k], acc[mi][ni], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) s += acc[mi][ni][v];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static float frand() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); }
static uint16_t bf16_trunc(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }

int main() {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = pr.multiProcessorCount;
    printf("device %s: %d CUs, nominal clock %.0f MHz\n", pr.name, cus, pr.clockRate / 1e3);
    // 64 operand sets x 18 fragments x 64 lanes x 8 bf16
    const size_t nfrag = 64 * 18 * 64;
    std::vector<uint16_t> h(nfrag * 8);
    u32x4* d_ops; float* d_f32; float* d_out;
    hipMalloc(&d_ops, nfrag * 16); hipMalloc(&d_f32, 64 * 48 * 64 * 4); hipMalloc(&d_out, (size_t)cus * 2 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"zeros", "random bf16", "limbs of random float32"};
    for (int wps = 1; wps <= 2; ++wps) {
        for (int kind = 0; kind < 3; ++kind) {
            srand(7);
            for (size_t f = 0; f < nfrag; ++f) {
                const int limb = (int)((f / 64) % 18) % 3;      // fragments are stored limb-minor: (tile, limb)
                for (int e = 0; e < 8; ++e) {
                    float x = frand();
                    uint16_t v = 0;
                    if (kind == 1) v = bf16_trunc(x);
                    if (kind == 2) {
                        uint16_t l0 = bf16_trunc(x); float r1 = x - bf16_f(l0);
                        uint16_t l1 = bf16_trunc(r1); float r2 = r1 - bf16_f(l1);
                        v = limb == 0 ? l0 : limb == 1 ? l1 : bf16_trunc(r2);
                    }
                    h[f * 8 + e] = v;
                }
            }
            hipMemcpy(d_ops, h.data(), nfrag * 16, hipMemcpyHostToDevice);
            const int iters = 400, blocks = cus * wps, warm = 60, reps = 60;
            auto launch = [&]() {
                if (wps == 1) hipLaunchKernelGGL(k_bf16<1>, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters);
                else hipLaunchKernelGGL(k_bf16<2>, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters);
            };
            for (int i = 0; i < warm; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double nm = (double)reps * blocks * 4 * iters * 48, fl = nm * 2.0 * 32 * 32 * 16;
            const double cyc = (double)ms * 1e-3 / ((double)reps * iters * 48 * wps);     // seconds per MFMA per SIMD
            printf("bf16 32x32x16, %d wave(s)/SIMD, %-24s: %7.1f us/launch, %7.1f TFLOP/s, %.1f ns per MFMA per SIMD = %.2f GHz-equivalent at 32 cycles\n",
                   wps, names[kind], ms * 1e3 / reps, fl / (ms * 1e-3) / 1e12, cyc * 1e9, 32.0 / (cyc * 1e9));
        }
        {
            std::vector<float> hf(64 * 48 * 64);
            srand(9);
            for (auto& x : hf) x = frand();
            hipMemcpy(d_f32, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
            const int iters = 100, blocks = cus * wps, warm = 40, reps = 40;
            auto launch = [&]() {
                if (wps == 1) hipLaunchKernelGGL(k_f32<1>, dim3(blocks), dim3(256), 0, 0, d_f32, d_out, iters);
                else hipLaunchKernelGGL(k_f32<2>, dim3(blocks), dim3(256), 0, 0, d_f32, d_out, iters);
            };
            for (int i = 0; i < warm; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double nm = (double)reps * blocks * 4 * iters * 64, fl = nm * 2.0 * 32 * 32 * 2;
            const double cyc = (double)ms * 1e-3 / ((double)reps * iters * 64 * wps);
            printf("f32  32x32x2 , %d wave(s)/SIMD, %-24s: %7.1f us/launch, %7.1f TFLOP/s, %.1f ns per MFMA per SIMD = %.2f GHz-equivalent at 64 cycles\n",
                   wps, "random float32", ms * 1e3 / reps, fl / (ms * 1e-3) / 1e12, cyc * 1e9, 64.0 / (cyc * 1e9));
        }
    }
    {   // accumulators in rotation, one wavefront per SIMD, limb operands (the last operand set uploaded above is "limbs")
        const int iters = 400, blocks = cus, warm = 40, reps = 40;
        for (int nacc = 1; nacc <= 8; nacc *= 2) {
            auto launch = [&]() {
                if (nacc == 1) hipLaunchKernelGGL(k_bf16_rot<1>, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters);
                else if (nacc == 2) hipLaunchKernelGGL(k_bf16_rot<2>, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters);
                else if (nacc == 4) hipLaunchKernelGGL(k_bf16_rot<4>, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters);
                else hipLaunchKernelGGL(k_bf16_rot<8>, dim3(blocks), dim3(256), 0, 0, d_ops, d_out, iters);
            };
            for (int i = 0; i < warm; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double nm = (double)reps * blocks * 4 * iters * 48, fl = nm * 2.0 * 32 * 32 * 16;
            const double sec = (double)ms * 1e-3 / ((double)reps * iters * 48);
            printf("bf16 32x32x16, 1 wave/SIMD, limbs, %d accumulator tile(s) in rotation: %7.1f TFLOP/s, %.1f ns per MFMA per SIMD\n",
                   nacc, fl / (ms * 1e-3) / 1e12, sec * 1e9);
        }
    }
    // what the numbers mean for the conv GEMM's big launch: T36 x [256 x 256] x [256 x 8000], 6 limb products
    printf("T36 x 256 x 256 x 8000: 37.75 GFLOP as float32, x6 limb products = 226.5 GFLOP of bf16 MFMA work\n");
    return 0;
}
