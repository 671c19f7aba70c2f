// tests/host/check_wide_sin2.hip -- host-side check of the engine's packed / signed sine evaluation (fd_math.hpp) against
// its own general scalar restatement of wide::f32x8::sin (wide_sinf, which tests/test_gpu_parity.py pins to the oracle
// on the device).  Compiled with hipcc for the HOST only; no GPU needed.  Also covers the scalar twin wide_sin1.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#define FD_HOST_ONLY 1
#include "fd_math.hpp"
int main() {
    using namespace fd;
    unsigned long long bad = 0, n = 0;
    uint64_t st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(st >> 32); };
    for (long i = 0; i < 30000000; i++) {
        float a, b;
        uint32_t r = rnd();
        int mode = i % 5;
        if (mode == 0) { a = ((int32_t)rnd()) * (1.0f / 2147483648.0f) * 12000.0f; b = ((int32_t)rnd()) * (1.0f / 2147483648.0f) * 7.0f; }
        else if (mode == 1) { uint32_t u = rnd(); memcpy(&a, &u, 4); u = rnd(); memcpy(&b, &u, 4); if (!(fabsf(a) < 12000.f)) a = 0.3f; if (!(fabsf(b) < 12000.f)) b = 0.0f; }
        else if (mode == 2) { // near quadrant ties: (k + 0.5) * pi/2
            int k = (int)(r % 16000) - 8000; a = (float)((k + 0.5) * 1.5707963267948966); a = nextafterf(a, (r & 1) ? 1e9f : -1e9f); b = (float)(k * 1.5707963267948966); if (b == 0.0f) b = 0.0f; }
        else if (mode == 3) { a = ((int32_t)rnd()) * (1.0f / 2147483648.0f) * 1e-3f; b = -a; if (a == 0.0f) { a = 1e-3f; b = -1e-3f; } }
        else { a = (float)(int)(r % 20000 - 10000) * 0.785398163f; b = ((int32_t)rnd()) * (1.0f / 2147483648.0f) * 100.f; }
        float tm = 0;
        v2f o = wide_sin2(v2f{a, b}, tm);
        if (!(tm < 8192.0f)) continue;
        float ea = wide_sinf(a), eb = wide_sinf(b);
        float tm1 = 0;
        float s1 = wide_sin1(a, tm1);
        n++;
        if (f2u(o.x) != f2u(ea) || f2u(o.y) != f2u(eb) || f2u(s1) != f2u(ea)) { if (bad < 10) printf("mismatch a=%a b=%a got %a %a exp %a %a\n", a, b, o.x, o.y, ea, eb); bad++; }
    }
    printf("checked %llu pairs, bad %llu\n", n, bad);
    return bad != 0;
}
