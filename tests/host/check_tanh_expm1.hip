// tests/host/check_tanh_expm1.hip -- host-side parity (hipcc, host only): the engine's select-form restatements of musl
// expm1f / tanhf (fd_math.hpp: every case evaluated, one selected) against the oracle's branch-form ones
// (oracle/o_math.h), bit for bit: every f32 whose low 9 mantissa bits are zero (all exponents, both signs, 2^23
// values), the neighbourhoods of every case boundary, and 40 M random bit patterns.  `--all`: all 2^32 bit patterns.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <atomic>
#include <thread>
#include <vector>
#define FD_HOST_ONLY 1
#include "fd_math.hpp"
extern "C" float o_math_tanhf(float);
extern "C" float o_math_expm1f(float);
static std::atomic<unsigned long long> bad{0}, seen{0};
static void check(uint32_t u) {
    using namespace fd;
    const float x = u2f(u);
    const uint32_t a = f2u(tanhf_musl(x)), b = f2u(o_math_tanhf(x));
    const uint32_t c = f2u(expm1f_musl(x)), d = f2u(o_math_expm1f(x));
    const bool nan_t = (a & 0x7fffffffu) > 0x7f800000u && (b & 0x7fffffffu) > 0x7f800000u;
    const bool nan_e = (c & 0x7fffffffu) > 0x7f800000u && (d & 0x7fffffffu) > 0x7f800000u;
    seen++;
    if ((a != b && !nan_t) || (c != d && !nan_e)) {
        if (bad < 10) printf("x = %a (%08x): tanh %08x vs %08x, expm1 %08x vs %08x\n", x, u, a, b, c, d);
        bad++;
    }
}
// tanhf_common (the ladder's packed path) == tanhf_musl wherever its guard does not trip; the guard trips exactly above 7.5 / on NaN
static void check_common(uint32_t u) {
    using namespace fd;
    const float x = u2f(u);
    uint32_t wm = 0;
    const uint32_t a = f2u(tanhf_common(x, wm));
    const bool trip = wm > TANH_COMMON_MAX_BITS;
    const bool should = (u & 0x7fffffffu) > TANH_COMMON_MAX_BITS;
    seen++;
    if (trip != should || (!trip && a != f2u(tanhf_musl(x)))) {
        if (bad < 10) printf("tanhf_common: x = %a (%08x): %08x vs %08x, guard %d (expected %d)\n", x, u, a, f2u(tanhf_musl(x)), (int)trip, (int)should);
        bad++;
    }
}
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--common")) {  // all bit patterns up to 7.5 in magnitude (both signs) + a band above the guard
        const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
        const uint64_t stride = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++)
            th.emplace_back([k, nt, stride] {
                for (uint64_t m = (uint64_t)k * stride; m <= (uint64_t)fd::TANH_COMMON_MAX_BITS + 70000; m += (uint64_t)nt * stride) {
                    check_common((uint32_t)m);
                    check_common((uint32_t)m | 0x80000000u);
                }
            });
        for (auto& t : th) t.join();
        for (uint32_t u : {0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00001u, 0x7f800001u, 0x41200001u, 0x7f7fffffu}) check_common(u);
        printf("tanhf_common: %llu values (stride %llu), bad %llu\n", (unsigned long long)seen, (unsigned long long)stride, (unsigned long long)bad);
        return bad ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--all")) {  // every one of the 2^32 bit patterns (one-off; ~1 min on 8 threads)
        const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++)
            th.emplace_back([k, nt] { for (uint64_t u = k; u < (1ull << 32); u += nt) check((uint32_t)u); });
        for (auto& t : th) t.join();
        printf("%llu values (all f32 bit patterns), bad %llu\n", (unsigned long long)seen, (unsigned long long)bad);
        return bad ? 1 : 0;
    }
    for (uint64_t u = 0; u < (1ull << 32); u += 512) check((uint32_t)u);
    const uint32_t edges[] = {0x41200000u, 0x3f0c9f54u, 0x3e82c578u, 0x00800000u, 0x4195b844u, 0x33000000u, 0x3eb17218u,
                              0x3F851592u, 0x42b17180u, 0x7f800000u, 0x3e800000u /* 0.25 */, 0x3f800000u, 0x40000000u};
    for (uint32_t e : edges)
        for (int d = -4096; d <= 4096; d++) { check(e + (uint32_t)d); check((e + (uint32_t)d) | 0x80000000u); }
    uint64_t st = 99;
    for (long i = 0; i < 40000000; i++) {
        st = st * 6364136223846793005ULL + 1442695040888963407ULL;
        uint32_t u = (uint32_t)(st >> 32);
        if (i & 1) u = (u & 0x807fffffu) | ((0x70u + (u >> 28)) << 23);  // half of them with exponents around 1
        check(u);
    }
    printf("%llu values, bad %llu\n", (unsigned long long)seen, (unsigned long long)bad);
    return bad ? 1 : 0;
}
