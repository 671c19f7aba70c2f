// tests/host/check_tanh_common_device.hip -- the DEVICE build of tanhf_common (the ladder filter's packed-path saturator,
// fd_math.hpp) against the device build of tanhf_musl -- itself identical to the oracle on all 2^32 inputs
// (check_tanh_device.hip, profiles/r02_tanh_device_exhaustive.txt) -- on EVERY f32 bit pattern: bit-equal wherever the
// guard does not trip, and the guard trips exactly for |x| > 7.5 and NaN.  Default arithmetic and, with
// -fgpu-flush-denormals-to-zero, the arithmetic of graphs with a Feedback node.  ~10 s on an MI355X box.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I fundsp_amd/csrc -o tests/host/_build/check_tanh_common_device tests/host/check_tanh_common_device.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fd_math.hpp"

__global__ void k(uint32_t base, unsigned long long* bad, uint32_t* first, uint32_t* lohi) {
    const uint32_t u = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = fd::u2f(u);
    uint32_t wm = 0;
    const uint32_t a = fd::f2u(fd::tanhf_common(x, wm));
    const bool trip = wm > fd::TANH_COMMON_MAX_BITS, should = (u & 0x7fffffffu) > fd::TANH_COMMON_MAX_BITS;
    if (trip != should || (!trip && a != fd::f2u(fd::tanhf_musl(x)))) {
        if (atomicAdd(bad, 1ull) == 0) *first = u;
        atomicMin(&lohi[0], u & 0x7fffffffu);  // the range of magnitudes that differ (diagnosis of experiments)
        atomicMax(&lohi[1], u & 0x7fffffffu);
    }
}

int main() {
    unsigned long long* d_bad;
    uint32_t* d_first;
    if (hipMalloc((void**)&d_bad, 8) != hipSuccess || hipMalloc((void**)&d_first, 4) != hipSuccess) return 2;
    hipMemset(d_bad, 0, 8);
    hipMemset(d_first, 0, 4);
    uint32_t* d_lohi;
    if (hipMalloc((void**)&d_lohi, 8) != hipSuccess) return 2;
    const uint32_t init[2] = {0xffffffffu, 0u};
    hipMemcpy(d_lohi, init, 8, hipMemcpyHostToDevice);
    for (uint32_t c = 0; c < 256; c++) hipLaunchKernelGGL(k, dim3((1u << 24) / 256), dim3(256), 0, 0, c << 24, d_bad, d_first, d_lohi);
    unsigned long long bad = 0;
    uint32_t first = 0;
    if (hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("tanhf_common vs tanhf_musl on the device, all 2^32 f32 bit patterns (equal where the guard holds, guard == |x| > 7.5 or NaN): bad %llu", bad);
    uint32_t lohi[2] = {0, 0};
    hipMemcpy(lohi, d_lohi, 8, hipMemcpyDeviceToHost);
    if (bad) printf(" (e.g. 0x%08x; magnitudes 0x%08x .. 0x%08x)", first, lohi[0], lohi[1]);
    printf("\n");
    return bad ? 1 : 0;
}
