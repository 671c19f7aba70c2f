// tests/host/check_trig_big.hip -- host-side check (hipcc, host only, no GPU): the engine's sinf/cosf/tanf
// (fd_math.hpp, musl restatement incl. the Payne-Hanek branch __rem_pio2_large) against the oracle's restatement
// (oracle/o_math.h) bit for bit over the whole f32 range: every exponent, both signs, inf / NaN.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <cfloat>
#define FD_HOST_ONLY 1
#include "fd_math.hpp"
#include "o_math.h"
static bool same(float a, float b) { return (a != a && b != b) || fd::f2u(a) == fd::f2u(b); }
int main() {
    using namespace fd;
    unsigned long long bad = 0, n = 0, big = 0;
    uint64_t st = 99;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(st >> 32); };
    for (long i = 0; i < 20000000; i++) {
        uint32_t u = rnd();
        if (i % 4 == 0) u = (u & 0x80ffffffu) | ((0x4du + (rnd() % 0x33u)) << 24);  // concentrate on 2^27 .. 2^127 and inf/NaN
        float x;
        memcpy(&x, &u, 4);
        if ((u & 0x7fffffffu) >= 0x4dc90fdbu) big++;
        n++;
        if (!same(sinf_musl(x), o_sinf(x)) || !same(cosf_musl(x), o_cosf(x)) || !same(tanf_musl(x), o_tanf(x))) {
            if (bad < 10) printf("mismatch x=%a: sin %a/%a cos %a/%a tan %a/%a\n", x, sinf_musl(x), o_sinf(x), cosf_musl(x), o_cosf(x), tanf_musl(x), o_tanf(x));
            bad++;
        }
    }
    printf("checked %llu arguments (%llu past the medium range), bad %llu\n", n, big, bad);
    return bad != 0;
}
