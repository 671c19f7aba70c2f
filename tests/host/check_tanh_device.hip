// tests/host/check_tanh_device.hip -- the DEVICE build of fd_math.hpp's tanhf_musl on every one of the 2^32 f32 bit patterns
// against the oracle's branch-form musl tanhf (bit for bit; NaN results count as equal), and the device form of the
// tolerance-mode fast_tanh1 (v_exp_f32 / v_rcp_f32) against double tanh (max absolute / relative error).  One-off tool
// (about a minute on an MI355X box); the result is kept in profiles/.  Build: see tools/README.md.
#include <hip/hip_runtime.h>
#ifndef CHECK_FTZ
#define CHECK_FTZ 0
#endif
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <xmmintrin.h>
#include "fd_math.hpp"
extern "C" float o_math_tanhf(float);

__global__ void k_tanh(uint32_t base, uint32_t* exact, float* fast) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = fd::u2f(base + i);
    exact[i] = fd::f2u(fd::tanhf_musl(x));
    fast[i] = fd::fast_tanh1(x);
}

int main() {
    const uint32_t CH = 1u << 24;
    uint32_t* d_e;
    float* d_f;
    if (hipMalloc((void**)&d_e, CH * 4) != hipSuccess || hipMalloc((void**)&d_f, CH * 4) != hipSuccess) return 2;
    std::vector<uint32_t> e(CH);
    std::vector<float> f(CH);
    std::atomic<unsigned long long> bad{0};
    const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
    std::vector<double> t_abs(nt, 0.0), t_rel(nt, 0.0);
    for (uint32_t c = 0; c < 256; c++) {
        const uint32_t base = c << 24;
        hipLaunchKernelGGL(k_tanh, dim3(CH / 256), dim3(256), 0, 0, base, d_e, d_f);
        if (hipMemcpy(e.data(), d_e, CH * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        if (hipMemcpy(f.data(), d_f, CH * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++)
            th.emplace_back([&, k] {
#if CHECK_FTZ  // the arithmetic of graphs with a Feedback node: build with -fgpu-flush-denormals-to-zero -DCHECK_FTZ=1
                _mm_setcsr(0x9fc0);  // FTZ + DAZ, as the oracle renders such graphs
#endif
                for (uint32_t i = k; i < CH; i += nt) {
                    const float x = fd::u2f(base + i);
                    const uint32_t want = fd::f2u(o_math_tanhf(x));
                    const bool both_nan = (e[i] & 0x7fffffffu) > 0x7f800000u && (want & 0x7fffffffu) > 0x7f800000u;
                    if (e[i] != want && !both_nan) {
                        if (bad++ < 10) printf("x = %a (%08x): device %08x, oracle %08x\n", x, base + i, e[i], want);
                    }
                    if (x == x && std::fabs(x) <= 3.0e38f) {
                        const double w = std::tanh((double)x), d = std::fabs((double)f[i] - w);
                        if (d > t_abs[k]) t_abs[k] = d;
                        if (w != 0 && std::fabs(x) >= 1.2e-38f && d / std::fabs(w) > t_rel[k]) t_rel[k] = d / std::fabs(w);
                    }
                }
            });
        for (auto& t : th) t.join();
    }
    double ma = 0, mr = 0;
    for (unsigned k = 0; k < nt; k++) { ma = std::fmax(ma, t_abs[k]); mr = std::fmax(mr, t_rel[k]); }
    printf("tanhf_musl on the device vs the oracle%s, all 2^32 f32 bit patterns: bad %llu\n", CHECK_FTZ ? " (both with denormals flushed)" : "", (unsigned long long)bad);
    printf("fast_tanh1 on the device vs double tanh, all finite normal f32: max abs error %.3g, max relative error %.3g\n", ma, mr);
    return bad ? 1 : 0;
}
