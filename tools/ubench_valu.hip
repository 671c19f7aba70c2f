// tools/ubench_valu.hip -- VALU issue-rate micro-benchmark for gfx950 (design input, not part of the product).
// Question: with ONE wave per SIMD (BASELINE config 3 = 1024 waves on 1024 SIMDs) how many cycles does an
// independent f32 VALU instruction cost, and does a second / fourth wave per SIMD raise aggregate throughput?
// Each wave runs ITER iterations of 16 independent ops of one kind; we report cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[16];
    float2_ y[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2_{x[i], x[i] + 1.0f}; }
    float2_ a2 = {a, a * 1.0001f}, b2 = {b, b * 1.0001f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));   // dependent chain below
            if (KIND == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double*)&y[i]) : "v"(*(double*)&a2));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// dependent chain: one accumulator
__global__ __launch_bounds__(256) void kdep(float* out, int iters, float a) {
    float x = threadIdx.x * 0.001f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

int main() {
    const int iters = 20000;
    float* out;
    hipMalloc(&out, 1024 * 16 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32", "v_mul_f64", "v_mul_f32 dependent"};
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk_khz);
    for (int kind = 0; kind < 7; kind++) {
        for (int wps : {1, 2, 4, 8}) {
            int grid = 256 * wps;  // 256-thread workgroups: 4 waves land on the 4 SIMDs of a CU
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 0.9999f, 0.5f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters, 0.9999f, 0.5f); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); break;
                default: hipLaunchKernelGGL(kdep, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            double instr_per_wave = (double)iters * 16;
            double cyc = ms * 1e-3 * (double)clk_khz * 1e3 / instr_per_wave;  // cycles per instruction per wave (wall)
            printf("%-22s waves/SIMD=%d  %.3f ms  %.2f cyc/instr/wave  -> %.2f cyc per SIMD-instruction\n", names[kind], wps,
                   ms, cyc, cyc / wps);
        }
    }
    return 0;
}
