// tests/host/check_envelope_spec.hip -- host-side fuzz (hipcc, host only): the process paths of AdsrLive and Envelope with
// their once-per-block preparation of the next segment (fd_nodes.hpp speculate / commit) against the oracle's envelope.rs
// restatement, which runs next_segment where the reference does.  Random ADSR times, sample rates from 2 kHz (many
// segment ends per block) to 192 kHz, seeds, block partitions with remainders, and gates with edges anywhere -- on the
// boundary sample, inside the attack, twice in one block, NaN / negative / zero levels.
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
static uint64_t st = 99;
static uint32_t rnd() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(st >> 32); }
static float uni() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); }

// one AudioNode::process call sequence: blocks of 64 with a shorter last one, as pipe_stage / render_body walk them
template <class NODE>
static void render(NODE& g, const std::vector<float>& x, std::vector<float>& y, int nout) {
    const size_t T = x.size();
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)(T - t0 < 64 ? T - t0 : 64);
        g.begin_block(size);
        for (int i = 0; i < size; i++) {
            float o[4];
            g.template step<PH_SIMD>(&x[t0 + i], o);
            for (int c = 0; c < nout; c++) y[(t0 + i) * nout + c] = o[c];
        }
    }
}
static void env_exp(float t, float* out, void* ctx) {
    const float* p = (const float*)ctx;
    out[0] = p[0] * expf_musl(-t * p[1]);
}
int main() {
    unsigned long long bad = 0, samples = 0;
    const double rates[] = {2000.0, 5512.5, 8000.0, 22050.0, 44100.0, 48000.0, 96000.0, 192000.0};
    for (int trial = 0; trial < 6000; trial++) {
        const double sr = rates[rnd() % 8];
        const size_t T = 64 * (2 + rnd() % 12) + (rnd() % 3 == 0 ? rnd() % 64 : 0);
        const uint64_t seed = ((uint64_t)rnd() << 32) | rnd();
        std::vector<float> x(T), want(T), got(T);
        // gate: a few random edges; sometimes exotic levels
        float level = (rnd() % 4 == 0) ? 1.0f : 0.0f;
        size_t next_edge = rnd() % 40;
        for (size_t i = 0; i < T; i++) {
            if (i == next_edge) {
                level = level > 0.0f ? ((rnd() % 8 == 0) ? -1.0f : 0.0f) : (0.25f + uni());
                next_edge = i + 1 + rnd() % (T / 2 + 1);
            }
            x[i] = level;
        }
        if (trial % 17 == 0) x[rnd() % T] = NAN;
        if (trial % 2 == 0) {
            const float a = 0.0005f + 0.05f * uni(), d = 0.001f + 0.2f * uni(), s = uni(), r = 0.001f + 0.3f * uni();
            AdsrLive g;
            Ctx ctx{};
            g.init();
            g.bind(ctx);
            g.attack = a; g.decay = d; g.sustain = s; g.release = r;
            g.update(sr);
            g.ping(false, seed);
            onode* n = o_adsr_live(a, d, s, r);
            o_set_sample_rate(n, sr);
            o_set_seed(n, seed);
            // two process calls in a row (state carried across launches)
            render(g, x, got, 1);
            o_render_blocks(n, T, 64, x.data(), want.data());
            for (size_t i = 0; i < T; i++) if (!same(got[i], want[i])) { if (bad++ < 5) printf("adsr trial %d sr %g frame %zu: %a vs %a\n", trial, sr, i, got[i], want[i]); break; }
            render(g, x, got, 1);
            o_render_blocks(n, T, 64, x.data(), want.data());
            for (size_t i = 0; i < T; i++) if (!same(got[i], want[i])) { if (bad++ < 5) printf("adsr (2nd call) trial %d sr %g frame %zu: %a vs %a\n", trial, sr, i, got[i], want[i]); break; }
            o_free(n);
        } else {
            float p[2] = {0.2f + uni(), 0.5f + 40.0f * uni()};
            Envelope<EnvExp> g;
            Ctx ctx{};
            g.init();
            g.bind(ctx);
            g.fn.a = p[0]; g.fn.k = p[1];
            g.ping(false, seed);
            g.update(sr);
            onode* n = o_envelope(0.002f, 1, env_exp, p);
            o_set_seed(n, seed);
            o_set_sample_rate(n, sr);
            std::vector<float> none(T, 0.0f);
            render(g, none, got, 1);
            o_render_blocks(n, T, 64, nullptr, want.data());
            for (size_t i = 0; i < T; i++) if (!same(got[i], want[i])) { if (bad++ < 5) printf("lfo trial %d sr %g frame %zu: %a vs %a\n", trial, sr, i, got[i], want[i]); break; }
            o_free(n);
        }
        samples += T;
    }
    printf("%llu samples, bad %llu\n", samples, bad);
    return bad ? 1 : 0;
}
