// tests/host/check_fast_sin.hip -- host-side bound (hipcc, host only): the tolerance-mode sine of fd_math.hpp
// (fast_sin1 / fast_sin2, FDSP_MATH_FAST; fast_tanh1 at the end) against the engine's restatement of wide::f32x8::sin and against double sin,
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
    // fast_tanh1 (Moog in tolerance mode).  The host build evaluates exp2f and a division where the device has v_exp_f32 and
    // v_rcp_f32 (1 ulp each): this bounds the FORMULA (branch point, polynomial, cancellation); the device form is bounded
    // through the ladder in tests/test_gpu_math_fast.py.
    double t_abs = 0, t_rel = 0;
    for (long i = 0; i < 20000000; i++) {
        const float span = (i % 3 == 0) ? 12.0f : (i % 3 == 1 ? 1.0f : 0.01f);
        const float x = ((int32_t)rnd()) * (1.0f / 2147483648.0f) * span;
        const double want = std::tanh((double)x), d = std::fabs((double)fast_tanh1(x) - want);
        if (d > t_abs) t_abs = d;
        if (want != 0 && d / std::fabs(want) > t_rel) t_rel = d / std::fabs(want);
    }
    const bool t_ok = t_abs < 2.5e-7 && t_rel < 8e-7 && fast_tanh1(INFINITY) == 1.0f && fast_tanh1(-INFINITY) == -1.0f &&
                      fast_tanh1(NAN) != fast_tanh1(NAN) && fast_tanh1(0.0f) == 0.0f && fast_tanh1(100.0f) == 1.0f;
    printf("fast_tanh1: max abs error %.3g, max relative error %.3g\n", t_abs, t_rel);
    printf(ok && t_ok ? "ok\n" : "FAILED\n");
    return ok && t_ok ? 0 : 1;
}
