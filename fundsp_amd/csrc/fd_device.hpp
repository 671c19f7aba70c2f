// fd_device.hpp -- device side of the MI355X voice-bank engine: bank kernels generic over a voice-graph type G.
// Included by the ahead-of-time kinds (fd_engine.hpp) and, verbatim, by run-time compiled graphs (fd_jit.cpp, hiprtc).
//
// Launch geometry: one wave64 workgroup per 64 voices (lane == voice).  At BASELINE config 3 (65 536 voices)
// that is 1024 single-wave workgroups = one wave per SIMD on a 256-CU MI355X; all of them are co-resident, so
// at any instant the chip writes one contiguous [frame][voice] row segment per wave (256 B per store
// instruction, 2 full 128-B lines) and rows advance in near lock-step -- HBM sees a streaming write.
// No MFMA anywhere: every per-lane recurrence is a scalar IIR / phase accumulator, not a contraction.
//
// Per-voice registers are loaded once per launch from the SoA `slots[slot][voice]` (coalesced, 256 B per wave
// per slot) and only STATE slots are written back at the end; output samples are the only per-frame HBM
// traffic of a generator graph (4 B per voice-sample).
//
// FDSP_LAYOUT_PLANAR ([voice][channel][frame], the reference BufferArray layout) is served by transposing
// 64 voices x 64 frames tiles through LDS: rows padded to 68 floats so that both the per-lane row access
// (ds_read/write_b128 of 4 consecutive frames) and the global<->LDS staging (16 lanes x float4 = one 256-B
// voice row) are bank-conflict free and every HBM access is a 256-B contiguous run.
#pragma once

#include "fd_nodes.hpp"

namespace fd {

constexpr int LAYOUT_VOICE_MINOR = 0, LAYOUT_PLANAR = 1;
constexpr int MODE_PROCESS = 0, MODE_TICK = 1;
constexpr int TILE_STRIDE = 68;  // floats per voice row in an LDS tile (64 + 4: keeps b128 alignment, breaks bank aliasing)

// ---- visitors --------------------------------------------------------------------------------------------
struct VLoad {
    const float* p;
    size_t stride;
    int slot;
    FD_D void f(float& x, FieldKind, const char*) { x = p[(size_t)slot++ * stride]; }
    FD_D void fi(float& x, FieldKind, const char*, int) { x = p[(size_t)slot++ * stride]; }
    FD_D void u32(uint32_t& x, FieldKind, const char*) { x = f2u(p[(size_t)slot++ * stride]); }
    FD_D void u64(uint64_t& x, FieldKind, const char*) {
        uint32_t lo = f2u(p[(size_t)slot * stride]);
        uint32_t hi = f2u(p[(size_t)(slot + 1) * stride]);
        slot += 2;
        x = ((uint64_t)hi << 32) | lo;
    }
    FD_D void enter(int) {}
    FD_D void leave() {}
};

template <bool ALL>
struct VStore {
    float* p;
    size_t stride;
    int slot;
    FD_D void f(float& x, FieldKind k, const char*) {
        if (ALL || k == STATE) p[(size_t)slot * stride] = x;
        slot++;
    }
    FD_D void fi(float& x, FieldKind k, const char*, int) {
        if (ALL || k == STATE) p[(size_t)slot * stride] = x;
        slot++;
    }
    FD_D void u32(uint32_t& x, FieldKind k, const char*) {
        if (ALL || k == STATE) p[(size_t)slot * stride] = u2f(x);
        slot++;
    }
    FD_D void u64(uint64_t& x, FieldKind k, const char*) {
        if (ALL || k == STATE) {
            p[(size_t)slot * stride] = u2f((uint32_t)x);
            p[(size_t)(slot + 1) * stride] = u2f((uint32_t)(x >> 32));
        }
        slot += 2;
    }
    FD_D void enter(int) {}
    FD_D void leave() {}
};

// ---- lifecycle kernels -------------------------------------------------------------------------------------
// op: 0 = construct (defaults, DEFAULT_SR, constructor ping), 1 = update (set_sample_rate / parameter change),
//     2 = reset, 3 = set_seed (seeds != null) or re-apply the construction hash (seeds == null)
template <class G>
FD_D void lifecycle_body(float* slots, size_t stride, size_t first, size_t count, int op, double sr,
                         const uint64_t* seeds, const void* aux, float* ring, uint32_t ring_cap) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    size_t v = first + i;
    G g;
    Ctx ctx{static_cast<const Aux*>(aux), ring + v, ring_cap, stride, 0};
    g.bind(ctx);
    {
        VLoad ld{slots + v, stride, 0};
        g.visit(ld);
    }
    if (op == 0) {
        g.init();
        g.update(sr);
        uint64_t h = g.ping(true, G::ID);  // Pipe::new etc: ping(true, AttoHash::new(Self::ID)) then ping(false, h)
        g.ping(false, h);
    } else if (op == 1) {
        g.update(sr);
    } else if (op == 2) {
        g.reset();
    } else {
        if (seeds) {
            g.ping(false, seeds[i]);  // AudioNode::set_seed audionode.rs:366-368
        } else {
            uint64_t h = g.ping(true, G::ID);
            g.ping(false, h);
        }
    }
    VStore<true> st{slots + v, stride, 0};
    g.visit(st);
}

template <class G>
__global__ __launch_bounds__(64) void k_lifecycle(float* slots, size_t stride, size_t first, size_t count, int op,
                                                  double sr, const uint64_t* seeds, const void* aux, float* ring,
                                                  uint32_t ring_cap) {
    lifecycle_body<G>(slots, stride, first, count, op, sr, seeds, aux, ring, ring_cap);
}

// LDS hand-off inside ONE wave (each wave owns its tiles): order the wave's own DS operations and stop the
// compiler from moving LDS accesses across the hand-off.  No s_barrier: waves of a workgroup stay decoupled.
FD_D void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the hot kernel ----------------------------------------------------------------------------------------
// WPB = waves per workgroup.  Four-wave workgroups are used whenever LDS allows: the dispatcher places the four
// waves of one workgroup on the four SIMDs of a CU, so a 65 536-voice bank (256 workgroups) lands exactly one wave
// per SIMD.  Single-wave workgroups do NOT spread evenly (measured with tools/census.hip: 1024 x 64-thread
// workgroups leave ~10 % of the SIMDs idle and double up as many), which stretches a VALU-bound launch.
template <class G, int MODE, int LAYOUT, int WPB>
FD_D void render_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                      float* __restrict__ out, size_t T, size_t fstride, const void* aux, float* ring,
                      uint32_t ring_cap) {
    constexpr int NI = G::IN, NO = G::OUT;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;  // wave in block
    // Voices per wave.  Small banks (fewer than one full wave per SIMD) are spread over more, partially filled waves so
    // that every SIMD of the chip has a wave to run: in the voice-minor layout the otherwise unused `fstride` argument
    // carries the number of voices per wave (16 / 32; 0 or 64 = full waves).
    const int vpw = (LAYOUT == LAYOUT_VOICE_MINOR && fstride > 0 && fstride < 64) ? (int)fstride : 64;
    const size_t v0 = ((size_t)blockIdx.x * WPB + wib) * vpw;
    const size_t v = v0 + lane;
    const bool active = v < V && lane < vpw;
    if (v0 >= stride) return;  // whole wave beyond the padded bank (last workgroup of a ragged bank)

    G g;
    // Padding lanes (V <= v < stride) own a private, zero-initialised column of the padded slot / ring arrays, so the
    // planar path may run them harmlessly; lanes past the padding (partial waves only) alias lane 0 and only read.
    const size_t vc = v < stride ? v : v0;
    Ctx ctx{static_cast<const Aux*>(aux), ring + vc, ring_cap, stride, 0};
    g.bind(ctx);
    {
        VLoad ld{slots + vc, stride, 0};  // stride is padded to a multiple of 64: always in bounds
        g.visit(ld);
    }

    if (LAYOUT == LAYOUT_VOICE_MINOR) {
        if (!active) return;
        const float* inv = in + v;
        float* outv = out + v;
        for (size_t t0 = 0; t0 < T; t0 += 64) {
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
            float fi[NI > 0 ? NI : 1], fo[NO];
            g.begin_block(size);
            const G snap = g;  // block-start registers, for the rollback below
#pragma unroll 4
            for (int i = 0; i < full; i += 2) {  // two frames per iteration (full is a multiple of 8)
                const size_t t = t0 + i;
                v2f pi[NI > 0 ? NI : 1], po[NO];
#pragma unroll
                for (int c = 0; c < NI; c++)
                    pi[c] = v2f{inv[((size_t)c * T + t) * V], inv[((size_t)c * T + t + 1) * V]};
                g.template step2<PH_SIMD>(pi, po);
#pragma unroll
                for (int c = 0; c < NO; c++) {
                    outv[((size_t)c * T + t) * V] = po[c].x;
                    outv[((size_t)c * T + t + 1) * V] = po[c].y;
                }
            }
            if (__builtin_expect(g.tripped(), 0)) {  // a packed-path shortcut left its exact domain: redo the block
                g = snap;
                for (int i = 0; i < full; i++) {
                    const size_t t = t0 + i;
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t) * V];
                    g.template step<PH_SIMD>(fi, fo);
#pragma unroll
                    for (int c = 0; c < NO; c++) outv[((size_t)c * T + t) * V] = fo[c];
                }
            }
            if (MODE == MODE_PROCESS) g.end_simd();
            for (int i = full; i < size; i++) {
                const size_t t = t0 + i;
#pragma unroll
                for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t) * V];
                g.template step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(fi, fo);
#pragma unroll
                for (int c = 0; c < NO; c++) outv[((size_t)c * T + t) * V] = fo[c];
            }
        }
    } else {
        // one private tile set per wave: no cross-wave sharing, so only wave-level ordering is needed
        __shared__ __attribute__((aligned(16))) float tin_all[WPB * (NI > 0 ? NI : 1) * 64 * TILE_STRIDE];
        __shared__ __attribute__((aligned(16))) float tout_all[WPB * NO * 64 * TILE_STRIDE];
        float* tin = tin_all + wib * (NI > 0 ? NI : 1) * 64 * TILE_STRIDE;
        float* tout = tout_all + wib * NO * 64 * TILE_STRIDE;
        const int sub = lane >> 4;         // which of 4 voice rows this lane stages per pass
        const int fr = (lane & 15) << 2;   // first of its 4 frames
        const bool aligned = ((fstride & 3) == 0) && ((((uintptr_t)in) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
        for (size_t t0 = 0; t0 < T; t0 += 64) {
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
            const bool vec = aligned && (t0 + 64 <= fstride);
            // stage inputs: global [voice][ch][frame] -> LDS [ch][voice][frame]
            if (NI > 0) {
#pragma unroll
                for (int c = 0; c < NI; c++) {
                    for (int r = 0; r < 16; r++) {
                        const int vr = r * 4 + sub;
                        const size_t gv = v0 + vr;
                        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (gv < V) {
                            const float* src = in + (gv * NI + c) * fstride + t0 + fr;
                            if (vec) {
                                x = *reinterpret_cast<const float4*>(src);
                            } else {
                                if (fr + 0 < size) x.x = src[0];
                                if (fr + 1 < size) x.y = src[1];
                                if (fr + 2 < size) x.z = src[2];
                                if (fr + 3 < size) x.w = src[3];
                            }
                        }
                        *reinterpret_cast<float4*>(&tin[(c * 64 + vr) * TILE_STRIDE + fr]) = x;
                    }
                }
                wave_sync();
            }
            // compute: each lane walks its own LDS row, 4 frames per ds_read_b128 / ds_write_b128.
            // pass 0 = packed two-frame path; pass 1 (rare) = rollback + scalar path if a packed shortcut tripped.
            g.begin_block(size);
            const G snap = g;
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1) {
                    if (__builtin_expect(!g.tripped(), 1)) break;
                    g = snap;
                }
                for (int i4 = 0; i4 < size; i4 += 4) {
                    float xi[NI > 0 ? NI : 1][4];
                    float xo[NO][4];
#pragma unroll
                    for (int c = 0; c < NI; c++) {
                        float4 q = *reinterpret_cast<const float4*>(&tin[(c * 64 + lane) * TILE_STRIDE + i4]);
                        xi[c][0] = q.x; xi[c][1] = q.y; xi[c][2] = q.z; xi[c][3] = q.w;
                    }
                    if (i4 < full) {
                        if (pass == 0) {
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                v2f pi[NI > 0 ? NI : 1], po[NO];
#pragma unroll
                                for (int c = 0; c < NI; c++) pi[c] = v2f{xi[c][j], xi[c][j + 1]};
                                g.template step2<PH_SIMD>(pi, po);
#pragma unroll
                                for (int c = 0; c < NO; c++) {
                                    xo[c][j] = po[c].x;
                                    xo[c][j + 1] = po[c].y;
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                float fi[NI > 0 ? NI : 1], fo[NO];
#pragma unroll
                                for (int c = 0; c < NI; c++) fi[c] = xi[c][j];
                                g.template step<PH_SIMD>(fi, fo);
#pragma unroll
                                for (int c = 0; c < NO; c++) xo[c][j] = fo[c];
                            }
                        }
                        if (MODE == MODE_PROCESS && i4 + 4 == full && (pass == 1 || !g.tripped())) g.end_simd();
                    } else {
                        if (pass == 0 && g.tripped()) break;  // the remainder is rendered by the redo pass
                        if (MODE == MODE_PROCESS && i4 == full && full == 0) g.end_simd();
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            float fi[NI > 0 ? NI : 1], fo[NO];
#pragma unroll
                            for (int c = 0; c < NI; c++) fi[c] = xi[c][j];
#pragma unroll
                            for (int c = 0; c < NO; c++) fo[c] = 0.0f;
                            if (i4 + j < size) g.template step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(fi, fo);
#pragma unroll
                            for (int c = 0; c < NO; c++) xo[c][j] = fo[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NO; c++)
                        *reinterpret_cast<float4*>(&tout[(c * 64 + lane) * TILE_STRIDE + i4]) =
                            make_float4(xo[c][0], xo[c][1], xo[c][2], xo[c][3]);
                }
            }
            wave_sync();
            // stage outputs: LDS [ch][voice][frame] -> global [voice][ch][frame]
#pragma unroll
            for (int c = 0; c < NO; c++) {
                for (int r = 0; r < 16; r++) {
                    const int vr = r * 4 + sub;
                    const size_t gv = v0 + vr;
                    if (gv < V) {
                        float4 x = *reinterpret_cast<const float4*>(&tout[(c * 64 + vr) * TILE_STRIDE + fr]);
                        float* dst = out + (gv * NO + c) * fstride + t0 + fr;
                        if (vec && fr + 4 <= ((size + 3) & ~3)) {
                            *reinterpret_cast<float4*>(dst) = x;
                        } else {
                            if (fr + 0 < size) dst[0] = x.x;
                            if (fr + 1 < size) dst[1] = x.y;
                            if (fr + 2 < size) dst[2] = x.z;
                            if (fr + 3 < size) dst[3] = x.w;
                        }
                    }
                }
            }
            wave_sync();
        }
        if (!active) return;
    }

    VStore<false> st{slots + v, stride, 0};
    g.visit(st);
}

template <class G, int MODE, int LAYOUT, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_render(float* __restrict__ slots, size_t stride, size_t V,
                                                     const float* __restrict__ in, float* __restrict__ out,
                                                     size_t T, size_t fstride, const void* aux, float* ring,
                                                     uint32_t ring_cap) {
    render_body<G, MODE, LAYOUT, WPB>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
}

// ---- two-wave pipeline split of a Pipe chain --------------------------------------------------------------------
// At one voice-wave per SIMD (65 536 voices on 1024 SIMDs) a lone wave issues one instruction per ~4.7 cycles while
// the VALU could take one every ~3.3 (profiles/r01_voice_sweep_*).  For graphs that are a chain A >> B >> C ... the
// chain is cut once: the PREFIX stages of 64 voices run in one wave, the SUFFIX stages of the same voices in a second
// wave one 64-sample block behind, the cut's channels handed over through a double-buffered LDS tile.  Both waves
// keep their own part of the voice state in registers; the per-sample arithmetic of every node is untouched, so the
// output is bit-identical to the single-wave kernel -- only the issue slots of the SIMD are now fed by two waves.
template <class T> struct Cost { static constexpr int v = 12; };  // rough VALU instructions per sample (cut placement only)
template <int N> struct Cost<Constant<N>> { static constexpr int v = 0; };
template <> struct Cost<Pass> { static constexpr int v = 0; };
template <> struct Cost<Sine> { static constexpr int v = 28; };
template <> struct Cost<Noise> { static constexpr int v = 10; };
template <> struct Cost<FixedSvf> { static constexpr int v = 12; };
template <int N> struct Cost<Moog<N>> { static constexpr int v = 130; };
template <int S> struct Cost<WaveSynth<S>> { static constexpr int v = 100; };
template <> struct Cost<AdsrLive> { static constexpr int v = 80; };
template <> struct Cost<Shaper> { static constexpr int v = 60; };
template <> struct Cost<Panner> { static constexpr int v = 2; };
template <class X, class Y> struct Cost<Pipe<X, Y>> { static constexpr int v = Cost<X>::v + Cost<Y>::v; };
template <class X, class Y> struct Cost<Stack<X, Y>> { static constexpr int v = Cost<X>::v + Cost<Y>::v; };
template <class O, class X, class Y> struct Cost<Binop<O, X, Y>> { static constexpr int v = Cost<X>::v + Cost<Y>::v + 1; };
template <class X, class U> struct Cost<Unop<X, U>> { static constexpr int v = Cost<X>::v + 1; };

template <class G> struct Chain { static constexpr int N = 1; };  // cut points of a (nested) Pipe chain = N - 1
template <class X, class Y> struct Chain<Pipe<X, Y>> { static constexpr int N = Chain<X>::N + Chain<Y>::N; };
// (A cut INSIDE Sine -- phase recurrence | polynomial -- balances config 3 at 35 | 36 instructions but measured slower
// (6.41 ms vs 5.93 ms): with both waves busy all the time the packed-f32 VALU pipe, not issue, is the limit.  The
// Split<1, Sine> specialisation below is kept for experiments: enable it with Chain<Sine>::N = 2.)

struct VGate {  // forwards to a slot visitor only while enabled; always advances the slot counter
    template <class V> struct W {
        V* v;
        bool on;
        FD_D void f(float& x, FieldKind k, const char* n) { if (on) v->f(x, k, n); else v->slot++; }
        FD_D void fi(float& x, FieldKind k, const char* n, int i) { if (on) v->fi(x, k, n, i); else v->slot++; }
        FD_D void u32(uint32_t& x, FieldKind k, const char* n) { if (on) v->u32(x, k, n); else v->slot++; }
        FD_D void u64(uint64_t& x, FieldKind k, const char* n) { if (on) v->u64(x, k, n); else v->slot += 2; }
        FD_D void enter(int) {}
        FD_D void leave() {}
    };
};

template <int K, class G> struct Split;  // K = number of chain stages in the prefix (1 .. N-1)

// Sine cut in the middle: the prefix owns the serial phase recurrence (all of Sine's state) and hands over the angle
// of the block path (or, on the tick path, the finished sample); the suffix owns the pure polynomial and its guard.
template <>
struct Split<1, Sine> {
    static constexpr int mid() { return 1; }
    static constexpr int pre_cost() { return 4; }
    template <int PH> static FD_D void pre_step2(Sine& g, const v2f* in, v2f* out) {
        if constexpr (PH == PH_SIMD) {
            v2f d = in[0] * g.sample_duration;
            float t0 = g.phase;
            g.phase += d.x;
            float t1 = g.phase;
            g.phase += d.y;
            out[0] = v2f{t0, t1} * F32_TAU;
        } else {
            g.template step2<PH>(in, out);
        }
    }
    template <int PH> static FD_D void pre_step(Sine& g, const float* in, float* out) {
        if constexpr (PH == PH_SIMD) {
            float tmp = g.phase;
            g.phase += in[0] * g.sample_duration;
            out[0] = tmp * F32_TAU;
        } else {
            g.template step<PH>(in, out);
        }
    }
    template <int PH> static FD_D void suf_step2(Sine& g, const v2f* in, v2f* out) {
        if constexpr (PH == PH_SIMD) out[0] = wide_sin2(in[0], g.tmax); else out[0] = in[0];
    }
    template <int PH> static FD_D void suf_step(Sine&, const float* in, float* out) {
        if constexpr (PH == PH_SIMD) out[0] = wide_sinf(in[0]); else out[0] = in[0];
    }
    static FD_D void pre_begin(Sine&, int) {}
    static FD_D void suf_begin(Sine& g, int n) { g.begin_block(n); }
    static FD_D void pre_end(Sine& g) { g.end_simd(); }
    static FD_D void suf_end(Sine&) {}
    static FD_D bool pre_tripped(const Sine&) { return false; }
    static FD_D bool suf_tripped(const Sine& g) { return g.tripped(); }
    template <bool PRE, class W> static FD_D void visit_part(Sine& g, W& w) { w.on = PRE; g.visit(w); }
};

template <int K, class X, class Y>
struct Split<K, Pipe<X, Y>> {
    using G = Pipe<X, Y>;
    static constexpr int NX = Chain<X>::N;
    static constexpr int WHERE = K == NX ? 0 : (K < NX ? -1 : 1);  // cut between x and y / inside x / inside y
    static constexpr int KX = K < NX ? K : 1, KY = K > NX ? K - NX : 1;
    static constexpr int mid() {
        if constexpr (WHERE == 0) return X::OUT; else if constexpr (WHERE < 0) return Split<KX, X>::mid(); else return Split<KY, Y>::mid();
    }
    static constexpr int pre_cost() {
        if constexpr (WHERE == 0) return Cost<X>::v;
        else if constexpr (WHERE < 0) return Split<KX, X>::pre_cost();
        else return Cost<X>::v + Split<KY, Y>::pre_cost();
    }
    template <int PH> static FD_D void pre_step2(G& g, const v2f* in, v2f* out) {
        if constexpr (WHERE == 0) g.x.template step2<PH>(in, out);
        else if constexpr (WHERE < 0) Split<KX, X>::template pre_step2<PH>(g.x, in, out);
        else { v2f t[X::OUT > 0 ? X::OUT : 1]; g.x.template step2<PH>(in, t); Split<KY, Y>::template pre_step2<PH>(g.y, t, out); }
    }
    template <int PH> static FD_D void pre_step(G& g, const float* in, float* out) {
        if constexpr (WHERE == 0) g.x.template step<PH>(in, out);
        else if constexpr (WHERE < 0) Split<KX, X>::template pre_step<PH>(g.x, in, out);
        else { float t[X::OUT > 0 ? X::OUT : 1]; g.x.template step<PH>(in, t); Split<KY, Y>::template pre_step<PH>(g.y, t, out); }
    }
    template <int PH> static FD_D void suf_step2(G& g, const v2f* in, v2f* out) {
        if constexpr (WHERE == 0) g.y.template step2<PH>(in, out);
        else if constexpr (WHERE < 0) { v2f t[X::OUT]; Split<KX, X>::template suf_step2<PH>(g.x, in, t); g.y.template step2<PH>(t, out); }
        else Split<KY, Y>::template suf_step2<PH>(g.y, in, out);
    }
    template <int PH> static FD_D void suf_step(G& g, const float* in, float* out) {
        if constexpr (WHERE == 0) g.y.template step<PH>(in, out);
        else if constexpr (WHERE < 0) { float t[X::OUT]; Split<KX, X>::template suf_step<PH>(g.x, in, t); g.y.template step<PH>(t, out); }
        else Split<KY, Y>::template suf_step<PH>(g.y, in, out);
    }
    static FD_D void pre_begin(G& g, int n) {
        if constexpr (WHERE == 0) g.x.begin_block(n);
        else if constexpr (WHERE < 0) Split<KX, X>::pre_begin(g.x, n);
        else { g.x.begin_block(n); Split<KY, Y>::pre_begin(g.y, n); }
    }
    static FD_D void suf_begin(G& g, int n) {
        if constexpr (WHERE == 0) g.y.begin_block(n);
        else if constexpr (WHERE < 0) { Split<KX, X>::suf_begin(g.x, n); g.y.begin_block(n); }
        else Split<KY, Y>::suf_begin(g.y, n);
    }
    static FD_D void pre_end(G& g) {
        if constexpr (WHERE == 0) g.x.end_simd();
        else if constexpr (WHERE < 0) Split<KX, X>::pre_end(g.x);
        else { g.x.end_simd(); Split<KY, Y>::pre_end(g.y); }
    }
    static FD_D void suf_end(G& g) {
        if constexpr (WHERE == 0) g.y.end_simd();
        else if constexpr (WHERE < 0) { Split<KX, X>::suf_end(g.x); g.y.end_simd(); }
        else Split<KY, Y>::suf_end(g.y);
    }
    static FD_D bool pre_tripped(const G& g) {
        if constexpr (WHERE == 0) return g.x.tripped();
        else if constexpr (WHERE < 0) return Split<KX, X>::pre_tripped(g.x);
        else return g.x.tripped() || Split<KY, Y>::pre_tripped(g.y);
    }
    static FD_D bool suf_tripped(const G& g) {
        if constexpr (WHERE == 0) return g.y.tripped();
        else if constexpr (WHERE < 0) return Split<KX, X>::suf_tripped(g.x) || g.y.tripped();
        else return Split<KY, Y>::suf_tripped(g.y);
    }
    // visit only the prefix (PRE) or only the suffix part of the slots; slot numbering stays that of G::visit
    template <bool PRE, class W> static FD_D void visit_part(G& g, W& w) {
        if constexpr (WHERE == 0) { w.on = PRE; g.x.visit(w); w.on = !PRE; g.y.visit(w); }
        else if constexpr (WHERE < 0) { Split<KX, X>::template visit_part<PRE>(g.x, w); w.on = !PRE; g.y.visit(w); }
        else { w.on = PRE; g.x.visit(w); Split<KY, Y>::template visit_part<PRE>(g.y, w); }
    }
};

template <class G> struct BestCut { static constexpr int K = 0; static constexpr bool ok = false; };
template <class X, class Y>
struct BestCut<Pipe<X, Y>> {
    using G = Pipe<X, Y>;
    static constexpr int N = Chain<G>::N, total = Cost<G>::v;
    template <int K> static constexpr int worst() { int p = Split<K, G>::pre_cost(); int s = total - p; return p > s ? p : s; }
    template <int K> static constexpr int best_from() {
        if constexpr (K >= N) return 0;
        else { int rest = best_from<K + 1>(); if (rest == 0) return K; return worst<K>() <= pick_worst(rest) ? K : rest; }
    }
    static constexpr int pick_worst(int k) { return pick<1>(k); }
    template <int K> static constexpr int pick(int k) { if constexpr (K >= N) return 1 << 30; else return k == K ? worst<K>() : pick<K + 1>(k); }
    static constexpr int K = best_from<1>();
    // worth it only if both halves carry real work and the graph has no delay rings / inputs on the suffix side
    static constexpr bool ok = G::RINGS == 0 && Split<K, G>::pre_cost() >= 8 && total - Split<K, G>::pre_cost() >= 8 &&
                               Split<K, G>::mid() == 1;  // one hand-over channel: 128 KiB of LDS for the double-buffered tiles of 4 groups
};

template <class G, int K, int MODE>
FD_D void render_split_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                            float* __restrict__ out, size_t T, const void* aux, float* ring, uint32_t ring_cap) {
    using S = Split<K, G>;
    constexpr int NI = G::IN, NO = G::OUT, NM = S::mid();
    // 8 waves: waves 0-3 run the prefix of voice groups 0-3, waves 4-7 the suffix of the same groups.  The hardware
    // places wave w and w+4 of a workgroup on the same SIMD, so every SIMD hosts one prefix and one suffix wave.
    __shared__ v2f hand[4][2][NM][32][64];  // [group][buffer][channel][frame pair][lane]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = w & 3;
    const bool suffix = w >= 4;
    const size_t v0 = ((size_t)blockIdx.x * 4 + grp) * 64;
    const size_t v = v0 + lane;
    const bool live = v0 < stride;             // whole group beyond the bank: still takes part in the barriers
    const bool active = v < V;
    G g;
    Ctx ctx{static_cast<const Aux*>(aux), ring + (live ? v : 0), ring_cap, stride, 0};
    g.bind(ctx);
    if (live) {
        VLoad ld{slots + v, stride, 0};
        VGate::W<VLoad> gate{&ld, true};
        if (suffix) S::template visit_part<false>(g, gate); else S::template visit_part<true>(g, gate);
    }
    const float* inv = in + v;
    float* outv = out + v;
    const size_t nblocks = (T + 63) / 64;
    for (size_t it = 0; it <= nblocks; it++) {
        const size_t b = suffix ? it - 1 : it;             // the block this wave works on in this round
        if (live && active && (suffix ? it >= 1 : it < nblocks)) {
            const size_t t0 = b * 64;
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
            v2f (*hb)[32][64] = hand[grp][b & 1];
            if (!suffix) {
                S::pre_begin(g, size);
                const G snap = g;
#pragma unroll 4
                for (int i = 0; i < full; i += 2) {
                    const size_t t = t0 + i;
                    v2f pi[NI > 0 ? NI : 1], pm[NM];
#pragma unroll
                    for (int c = 0; c < NI; c++) pi[c] = v2f{inv[((size_t)c * T + t) * V], inv[((size_t)c * T + t + 1) * V]};
                    S::template pre_step2<PH_SIMD>(g, pi, pm);
#pragma unroll
                    for (int c = 0; c < NM; c++) hb[c][i >> 1][lane] = pm[c];
                }
                if (__builtin_expect(S::pre_tripped(g), 0)) {
                    g = snap;
                    for (int i = 0; i < full; i++) {
                        float fi[NI > 0 ? NI : 1], fm[NM];
#pragma unroll
                        for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t0 + i) * V];
                        S::template pre_step<PH_SIMD>(g, fi, fm);
#pragma unroll
                        for (int c = 0; c < NM; c++) reinterpret_cast<float*>(&hb[c][i >> 1][lane])[i & 1] = fm[c];
                    }
                }
                if (MODE == MODE_PROCESS) S::pre_end(g);
                for (int i = full; i < size; i++) {
                    float fi[NI > 0 ? NI : 1], fm[NM];
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t0 + i) * V];
                    S::template pre_step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(g, fi, fm);
#pragma unroll
                    for (int c = 0; c < NM; c++) reinterpret_cast<float*>(&hb[c][i >> 1][lane])[i & 1] = fm[c];
                }
            } else {
                S::suf_begin(g, size);
                const G snap = g;
#pragma unroll 4
                for (int i = 0; i < full; i += 2) {
                    const size_t t = t0 + i;
                    v2f pm[NM], po[NO];
#pragma unroll
                    for (int c = 0; c < NM; c++) pm[c] = hb[c][i >> 1][lane];
                    S::template suf_step2<PH_SIMD>(g, pm, po);
#pragma unroll
                    for (int c = 0; c < NO; c++) {
                        outv[((size_t)c * T + t) * V] = po[c].x;
                        outv[((size_t)c * T + t + 1) * V] = po[c].y;
                    }
                }
                if (__builtin_expect(S::suf_tripped(g), 0)) {
                    g = snap;
                    for (int i = 0; i < full; i++) {
                        float fm[NM], fo[NO];
#pragma unroll
                        for (int c = 0; c < NM; c++) fm[c] = reinterpret_cast<const float*>(&hb[c][i >> 1][lane])[i & 1];
                        S::template suf_step<PH_SIMD>(g, fm, fo);
#pragma unroll
                        for (int c = 0; c < NO; c++) outv[((size_t)c * T + t0 + i) * V] = fo[c];
                    }
                }
                if (MODE == MODE_PROCESS) S::suf_end(g);
                for (int i = full; i < size; i++) {
                    float fm[NM], fo[NO];
#pragma unroll
                    for (int c = 0; c < NM; c++) fm[c] = reinterpret_cast<const float*>(&hb[c][i >> 1][lane])[i & 1];
                    S::template suf_step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(g, fm, fo);
#pragma unroll
                    for (int c = 0; c < NO; c++) outv[((size_t)c * T + t0 + i) * V] = fo[c];
                }
            }
        }
        __syncthreads();  // hand-over point: prefix block `it` is complete, suffix has drained block `it - 1`
    }
    if (live && active) {
        VStore<false> st{slots + v, stride, 0};
        VGate::W<VStore<false>> gate{&st, true};
        if (suffix) S::template visit_part<false>(g, gate); else S::template visit_part<true>(g, gate);
    }
}

template <class G, int K, int MODE>
__global__ __launch_bounds__(512) void k_render_split(float* __restrict__ slots, size_t stride, size_t V,
                                                      const float* __restrict__ in, float* __restrict__ out, size_t T,
                                                      const void* aux, float* ring, uint32_t ring_cap) {
    render_split_body<G, K, MODE>(slots, stride, V, in, out, T, aux, ring, ring_cap);
}

// Launch policy for the voice-minor layout: voices per wave such that the grid has at least one wave per SIMD.
inline int voices_per_wave(size_t V, int simds) {
    int vpw = 64;
    while (vpw > 16 && (V + vpw - 1) / vpw < (size_t)simds) vpw >>= 1;
    return vpw;
}

// LDS the planar path needs per wave, and the workgroup width chosen from it (shared by the AOT and JIT launchers)
template <class G, int LAYOUT>
struct RenderGeom {
    static constexpr size_t lds_per_wave = LAYOUT == LAYOUT_PLANAR ? (size_t)(G::IN + G::OUT) * 64 * TILE_STRIDE * 4 : 0;
    static constexpr int WPB = (lds_per_wave * 4 <= 160 * 1024 - 1024) ? 4 : 1;
};

// Device-side slot introspection for run-time compiled graphs: one thread walks visit() and writes
// "<path>:<name> <kind>\n" per slot word into `out` (the ahead-of-time kinds do this on the host, VDescribe).
struct VDescribeDev {
    char* out;
    int pos, cap;
    int path[32];
    int depth;
    FD_D void put(char c) { if (pos < cap - 1) out[pos++] = c; }
    FD_D void puts_(const char* s) { while (*s) put(*s++); }
    FD_D void putn(int n) {
        char buf[12];
        int k = 0;
        if (n == 0) buf[k++] = '0';
        while (n > 0) { buf[k++] = (char)('0' + n % 10); n /= 10; }
        while (k > 0) put(buf[--k]);
    }
    FD_D void prefix() {
        for (int i = 0; i < depth; i++) { if (i) put('.'); putn(path[i]); }
        put(':');
    }
    FD_D void line(const char* name, int index, const char* suffix, int kind) {
        prefix();
        puts_(name);
        if (index >= 0) { put('['); putn(index); put(']'); }
        puts_(suffix);
        put(' ');
        putn(kind);
        put('\n');
    }
    FD_D void f(float&, FieldKind k, const char* name) { line(name, -1, "", (int)k); }
    FD_D void fi(float&, FieldKind k, const char* name, int index) { line(name, index, "", (int)k); }
    FD_D void u32(uint32_t&, FieldKind k, const char* name) { line(name, -1, "", (int)k); }
    FD_D void u64(uint64_t&, FieldKind k, const char* name) { line(name, -1, ".lo", (int)k); line(name, -1, ".hi", (int)k); }
    FD_D void enter(int i) { if (depth < 32) path[depth] = i; depth++; }
    FD_D void leave() { depth--; }
};
template <class G>
FD_D void describe_body(char* out, int cap, int* meta) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    G g;
    VDescribeDev d{out, 0, cap, {0}, 0};
    g.visit(d);
    out[d.pos] = 0;
    meta[0] = G::IN;
    meta[1] = G::OUT;
    meta[2] = G::RINGS;
    meta[3] = d.pos;
    meta[4] = RenderGeom<G, LAYOUT_PLANAR>::WPB;
}

}  // namespace fd
