// tests/host/check_fast_sin.hip -- host-side bound (hipcc, host only): the tolerance-mode sine of fd_math.hpp
// (fast_sin1 / fast_sin2, FDSP_MATH_FAST) against the engine's restatement of wide::f32x8::sin and against double sin,
// over the arguments Sine::process produces: x = fl(phase * TAU), |phase| up to a few thousand turns.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#define FD_HOST_ONLY 1
#include "fd_math.hpp"
int main() {
    using namespace fd;
    uint64_t st = 4242;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(st >> 32); };
    double worst_w = 0, worst_d = 0, worst_w_big = 0;
    unsigned long long twin = 0;
    for (long i = 0; i < 40000000; i++) {
        const float span = (i % 4 == 0) ? 3000.0f : (i % 4 == 1 ? 64.0f : 3.5f);   // turns
        const float phase = ((int32_t)rnd()) * (1.0f / 2147483648.0f) * span;
        const float x = phase * F32_TAU;
        const float f = fast_sin1(x), w = wide_sinf(x);
        const v2f f2 = fast_sin2(v2f{x, -x});
        if (f2u(f2.x) != f2u(f) || f2u(f2.y) != f2u(fast_sin1(-x))) twin++;
        const double dw = std::fabs((double)f - (double)w), dd = std::fabs((double)f - std::sin((double)x));
        if (span <= 64.0f) { if (dw > worst_w) worst_w = dw; if (dd > worst_d) worst_d = dd; }
        else if (dw > worst_w_big) worst_w_big = dw;
    }
    printf("max |fast - wide| %.3g (|phase| <= 64 turns), %.3g (<= 3000 turns); max |fast - sin| %.3g; packed != scalar: %llu\n",
           worst_w, worst_w_big, worst_d, twin);
    const bool ok = worst_w < 1.5e-7 && worst_d < 1.5e-7 && worst_w_big < 4e-7 && twin == 0;  // (sin(-0.0) comes out +0.0 here: a tolerance mode, |diff| = 0)
    printf(ok ? "ok\n" : "FAILED\n");
    return ok ? 0 : 1;
}
