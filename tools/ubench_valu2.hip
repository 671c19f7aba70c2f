// tools/ubench_valu2.hip -- VALU throughput micro-benchmark for gfx950, second take (design input, not product).
// Fixes over ubench_valu.hip: (1) the device is warmed up for ~0.3 s before anything is timed (the first numbers of
// the old tool were taken while the clocks were still ramping); (2) every kernel reads s_memtime (shader clock
// counter) and s_memrealtime (constant 100 MHz) so the ACTUAL shader clock under this load is reported;
// (3) mixes of packed and plain ops as they occur in the FM+SVF kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

// KIND: 0 v_mul_f32, 1 v_fma_f32, 2 v_pk_mul_f32, 3 v_pk_fma_f32, 4 v_pk_add_f32, 5 alternating pk_mul / v_mul,
//       6 v_mul_f32 dependent chain, 7 v_pk_mul dependent chain, 8 v_cndmask-like integer (v_and_b32), 9 v_bitop3/v_lshl mix
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a, float b) {
    float x[16];
    float2_ y[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2_{x[i], x[i] + 1.0f}; }
    float2_ a2 = {a, a * 1.0001f}, b2 = {b, b * 1.0001f};
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
            if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
            if (KIND == 5) { if (i & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
            if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[0]) : "v"(a));
            if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[0]) : "v"(a2));
            if (KIND == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1 \n v_mul_f32 %2, %2, %1" : "+v"(x[i]), "+v"(y[i].x) : "v"(a));
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* clk, int wall_khz) {
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4, 8}) {
        int grid = 256 * wps;  // 256-thread workgroups: 4 waves land on the 4 SIMDs of a CU
        float ms = 0;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[2];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double n = (double)iters * 16 * (KIND == 9 ? 2 : 1);
        double real_s = (double)h[1] / (wall_khz * 1e3);
        double mhz_memtime = (double)h[0] / real_s / 1e6;
        printf("%-26s waves/SIMD=%d  %.3f ms  memtime-rate %.0f MHz  %.2f ns/instr/wave -> %.3f ns per SIMD-instruction\n", name,
               wps, ms, mhz_memtime, ms * 1e6 / n, ms * 1e6 / n / wps);
    }
}

int main() {
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 1024 * 16 * 64 * sizeof(float));
    hipMalloc(&clk, 16);
    int clk_khz = 0, wall_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("nominal clock %d kHz, wall clock %d kHz\n", clk_khz, wall_khz);
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, clk, 40000, 0.9999f, 0.5f);  // warm-up
    hipDeviceSynchronize();
    run<0>("v_mul_f32", out, clk, wall_khz);
    run<1>("v_fma_f32", out, clk, wall_khz);
    run<2>("v_pk_mul_f32", out, clk, wall_khz);
    run<3>("v_pk_fma_f32", out, clk, wall_khz);
    run<4>("v_pk_add_f32", out, clk, wall_khz);
    run<5>("pk_mul / v_mul alternating", out, clk, wall_khz);
    run<6>("v_mul_f32 dependent", out, clk, wall_khz);
    run<7>("v_pk_mul_f32 dependent", out, clk, wall_khz);
    run<8>("v_and_b32", out, clk, wall_khz);
    run<9>("v_mul_f32 x2 per asm", out, clk, wall_khz);
    return 0;
}
