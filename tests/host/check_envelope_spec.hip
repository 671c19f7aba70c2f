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
// ... and the way the engine's kernels walk a block: packed pairs (step2) tile by tile over the full 8-sample items, every
// tile followed by the rollback test -- tripped(), or a forced one standing for another node of the segment --, end_simd,
// then the remainder samples.  `tile` = 8 .. 64 frames (pipe_stage's SUB); `force` = per-mille of tiles re-rendered anyway.
static unsigned long long n_tiles = 0, n_planned = 0, n_tripped = 0, n_forced = 0;
template <class NODE>
static void render_packed(NODE& g, const std::vector<float>& x, std::vector<float>& y, int tile, unsigned force) {
    const size_t T = x.size();
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)(T - t0 < 64 ? T - t0 : 64), full = size & ~7;
        g.begin_block(size);
        for (int lo = 0; lo < full; lo += tile) {
            const int hi = lo + tile < full ? lo + tile : full;
            const NODE snap = g;
            for (int i = lo; i < hi; i += 2) {
                v2f in = v2f{x[t0 + i], x[t0 + i + 1]}, out;
                g.template step2<PH_SIMD>(&in, &out);
                y[t0 + i] = out.x;
                y[t0 + i + 1] = out.y;
            }
            n_tiles++;
            n_planned += g.fast;
            const bool trip = g.tripped(), forced = !trip && rnd() % 1000 < force;
            n_tripped += trip;
            n_forced += forced;
            if (trip || forced) {
                g = snap;
                for (int i = lo; i < hi; i++) g.template step<PH_SIMD>(&x[t0 + i], &y[t0 + i]);
            }
        }
        g.end_simd();
        for (int i = full; i < size; i++) g.template step<PH_REM>(&x[t0 + i], &y[t0 + i]);
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
            // the packed walk (planned blocks): same two calls on a fresh node, random tile size, sometimes forced rollbacks
            {
                AdsrLive h;
                h.init();
                h.bind(ctx);
                h.attack = a; h.decay = d; h.sustain = s; h.release = r;
                h.update(sr);
                h.ping(false, seed);
                onode* m = o_adsr_live(a, d, s, r);
                o_set_sample_rate(m, sr);
                o_set_seed(m, seed);
                const int tile = 8 << (rnd() % 4);
                const unsigned force = (rnd() % 3 == 0) ? 200u : 0u;
                for (int call = 0; call < 2; call++) {
                    render_packed(h, x, got, tile, force);
                    o_render_blocks(m, T, 64, x.data(), want.data());
                    for (size_t i = 0; i < T; i++) if (!same(got[i], want[i])) { if (bad++ < 5) printf("adsr packed (call %d, tile %d, force %u) trial %d sr %g frame %zu: %a vs %a\n", call, tile, force, trial, sr, i, got[i], want[i]); break; }
                }
                // the registers the walk leaves behind are state: compare them with a node walked sample by sample
                const float regs_h[] = {h.t, h.t0, h.t1, h.v0, h.v1, h.value, h.value_d, h.attacked, h.attack_start, h.release_start};
                const float regs_g[] = {g.t, g.t0, g.t1, g.v0, g.v1, g.value, g.value_d, g.attacked, g.attack_start, g.release_start};
                for (int k = 0; k < 10; k++) if (!same(regs_h[k], regs_g[k])) { if (bad++ < 5) printf("adsr packed trial %d sr %g: register %d %a vs %a\n", trial, sr, k, regs_h[k], regs_g[k]); break; }
                if (h.t_hash != g.t_hash) { if (bad++ < 5) printf("adsr packed trial %d: t_hash differs\n", trial); }
                o_free(m);
            }
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
    printf("packed walk of adsr_live: %llu tiles, %llu of them in a planned block, %llu re-rendered for a trigger / an unplannable block, %llu by force\n",
           n_tiles, n_planned, n_tripped, n_forced);
    if (n_planned < n_tiles / 2 || n_tripped == 0 || n_forced == 0) { printf("the packed walk was not exercised\n"); bad++; }
    printf("%llu samples, bad %llu\n", samples, bad);
    return bad ? 1 : 0;
}
