// tests/host/check_svf_paths.hip -- host-side check (hipcc, host only): the packed-path forms of the SVF recurrence
// (SvfCore::tick_fused: 2*v - ic as one FMA; FixedSvfLp: lowpass output = v2) with their tile guards and rollback,
// against the reference-order scalar form SvfCore::tick and against the oracle's svf tick -- on bursts of huge inputs
// (2*v overflows), infinities, NaNs, denormals, zeros.  Mirrors what pipe_stage / render_body do per tile.
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#define FD_HOST_ONLY 1
#include "fd_nodes.hpp"
extern "C" {
#include "fundsp_oracle.h"
}
using namespace fd;
static bool same(float a, float b) { return (a != a && b != b) || f2u(a) == f2u(b); }
template <class NODE>
static void render_tiled(NODE g, const std::vector<float>& x, std::vector<float>& y, int tile) {
    for (size_t t0 = 0; t0 < x.size(); t0 += tile) {
        g.begin_block(tile);
        const NODE snap = g;
        for (int i = 0; i < tile; i += 2) {
            v2f in{x[t0 + i], x[t0 + i + 1]}, o;
            g.template step2<PH_SIMD>(&in, &o);
            y[t0 + i] = o.x;
            y[t0 + i + 1] = o.y;
        }
        if (g.tripped()) {
            g = snap;
            for (int i = 0; i < tile; i++) g.template step<PH_SIMD>(&x[t0 + i], &y[t0 + i]);
        }
    }
}
int main() {
    uint64_t st = 7;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(st >> 32); };
    auto uni = [&]() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); };
    unsigned long long bad = 0, cases = 0;
    for (int trial = 0; trial < 4000; trial++) {
        const int mode = trial % 3 == 0 ? (int)(rnd() % 9) : SVF_LOWPASS;
        const float fc = 20.0f * powf(1000.0f, uni()), q = 0.3f + 5.0f * uni(), gain = 0.25f + 3.0f * uni();
        const int T = 256;
        std::vector<float> x(T), want(T), a(T), b(T);
        for (auto& v : x) v = 2.0f * uni() - 1.0f;
        const int kind = trial % 8, at = 8 + (int)(rnd() % 200), len = 1 + (int)(rnd() % 40);
        for (int i = at; i < at + len && i < T; i++) {
            if (kind == 1) x[i] = 3.0e38f;
            if (kind == 2) x[i] = -3.3e38f;
            if (kind == 3) x[i] = (i & 1) ? 3.0e38f : -3.0e38f;
            if (kind == 4 && i == at) x[i] = INFINITY;
            if (kind == 5 && i == at) x[i] = NAN;
            if (kind == 6) x[i] = 1.0e-41f;
            if (kind == 7) x[i] = -0.0f;
        }
        FixedSvf g;
        Ctx ctx{};
        g.init();
        g.bind(ctx);
        g.mode = (float)mode; g.cutoff = fc; g.q = q; g.gain = gain;
        g.update(48000.0);
        // oracle
        onode* n = o_fixed_svf(mode, fc, q, gain);
        o_set_sample_rate(n, 48000.0);
        for (int i = 0; i < T; i++) o_tick(n, &x[i], &want[i]);
        o_free(n);
        render_tiled(g, x, a, trial % 2 ? 64 : 32);
        for (int i = 0; i < T; i++) if (!same(a[i], want[i])) { if (bad < 5) printf("generic: trial %d frame %d got %a want %a\n", trial, i, a[i], want[i]); bad++; break; }
        if (svf_is_plain_lowpass(g)) {
            FixedSvfLp l;
            memcpy((void*)&l, (const void*)&g, sizeof g);
            render_tiled(l, x, b, trial % 2 ? 64 : 32);
            for (int i = 0; i < T; i++) if (!same(b[i], want[i])) { if (bad < 5) printf("lowpass: trial %d frame %d got %a want %a\n", trial, i, b[i], want[i]); bad++; break; }
            cases++;
        }
    }
    printf("lowpass-specialised cases %llu, bad %llu\n", cases, bad);
    return bad != 0;
}
