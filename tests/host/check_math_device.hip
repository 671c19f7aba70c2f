// tests/host/check_math_device.hip -- the DEVICE builds of fd_math.hpp's restated libm / wide functions on every one of the
// 2^32 f32 bit patterns against the oracle's restatements (bit for bit; NaN results count as equal): sinf, cosf, tanf,
// expf, expm1f, tanhf, atanf (musl) and wide's f32x8 sin / atan, plus the packed wide_sin2 against the scalar form on
// the device itself.  One-off tool (two to three minutes on an MI355X box); the result is kept in profiles/.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "fd_math.hpp"
extern "C" {
float o_math_sinf(float); float o_math_cosf(float); float o_math_tanf(float); float o_math_tanhf(float); float o_math_expf(float);
float o_math_expm1f(float); float o_math_wide_sinf(float); float o_math_atanf(float); float o_math_wide_atanf(float);
}
template <int F> __device__ float dev_fn(float x) {
    using namespace fd;
    if (F == 0) return sinf_musl(x);
    if (F == 1) return cosf_musl(x);
    if (F == 2) return tanf_musl(x);
    if (F == 3) return expf_musl(x);
    if (F == 4) return expm1f_musl(x);
    if (F == 5) return tanhf_musl(x);
    if (F == 6) return atanf_musl(x);
    if (F == 7) return wide_atanf(x);
    return wide_sinf(x);
}
template <int F> __global__ void k_fn(uint32_t base, uint32_t* out, unsigned long long* packed_bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = fd::u2f(base + i);
    const float y = dev_fn<F>(x);
    out[i] = fd::f2u(y);
    if (F == 8) {  // the packed form the headline kernel runs, both lanes, against the scalar form
        float tmax = 0.0f;
        const fd::v2f p = fd::wide_sin2(fd::v2f{x, -x}, tmax);
        const float ym = fd::wide_sinf(-x);
        const bool nan0 = y != y && p.x != p.x, nan1 = ym != ym && p.y != p.y;
        // wide_sin2's own domain: quadrant index below 8192 (it raises tmax beyond, and the caller re-renders the block with the
        // scalar form), and not -0.0 (diverted by the caller)
        const bool in_domain = tmax < 8192.0f && (base + i) != 0u && (base + i) != 0x80000000u;
        if (in_domain && ((fd::f2u(p.x) != fd::f2u(y) && !nan0) || (fd::f2u(p.y) != fd::f2u(ym) && !nan1))) atomicAdd(packed_bad, 1ull);
    }
}
typedef float (*host_fn)(float);
template <int F> unsigned long long sweep(const char* name, host_fn want_fn, uint32_t* d_out, unsigned long long* d_bad) {
    const uint32_t CH = 1u << 24;
    std::vector<uint32_t> e(CH);
    std::atomic<unsigned long long> bad{0};
    const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
    hipMemset(d_bad, 0, 8);
    for (uint32_t c = 0; c < 256; c++) {
        const uint32_t base = c << 24;
        hipLaunchKernelGGL(k_fn<F>, dim3(CH / 256), dim3(256), 0, 0, base, d_out, d_bad);
        if (hipMemcpy(e.data(), d_out, CH * 4, hipMemcpyDeviceToHost) != hipSuccess) return ~0ull;
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++)
            th.emplace_back([&, k] {
                for (uint32_t i = k; i < CH; i += nt) {
                    const float x = fd::u2f(base + i);
                    const uint32_t want = fd::f2u(want_fn(x));
                    const bool both_nan = (e[i] & 0x7fffffffu) > 0x7f800000u && (want & 0x7fffffffu) > 0x7f800000u;
                    if (e[i] != want && !both_nan)
                        if (bad++ < 5) printf("  %s: x = %a (%08x): device %08x, oracle %08x\n", name, x, base + i, e[i], want);
                }
            });
        for (auto& t : th) t.join();
    }
    unsigned long long pb = 0;
    hipMemcpy(&pb, d_bad, 8, hipMemcpyDeviceToHost);
    printf("%-12s device vs oracle, all 2^32 f32 bit patterns: bad %llu", name, (unsigned long long)bad);
    if (F == 8) printf("; packed wide_sin2 (x, -x) vs scalar on the device, inside its stated domain: bad %llu", pb);
    printf("\n");
    fflush(stdout);
    return bad + pb;
}
int main() {
    uint32_t* d_out;
    unsigned long long* d_bad;
    if (hipMalloc((void**)&d_out, (1u << 24) * 4) != hipSuccess || hipMalloc((void**)&d_bad, 8) != hipSuccess) return 2;
    unsigned long long bad = 0;
    bad += sweep<8>("wide_sinf", o_math_wide_sinf, d_out, d_bad);
    bad += sweep<0>("sinf_musl", o_math_sinf, d_out, d_bad);
    bad += sweep<1>("cosf_musl", o_math_cosf, d_out, d_bad);
    bad += sweep<2>("tanf_musl", o_math_tanf, d_out, d_bad);
    bad += sweep<3>("expf_musl", o_math_expf, d_out, d_bad);
    bad += sweep<4>("expm1f_musl", o_math_expm1f, d_out, d_bad);
    bad += sweep<5>("tanhf_musl", o_math_tanhf, d_out, d_bad);
    bad += sweep<6>("atanf_musl", o_math_atanf, d_out, d_bad);
    bad += sweep<7>("wide_atanf", o_math_wide_atanf, d_out, d_bad);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
