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
// What a node's constructor does about hashing.  Combinators ping themselves with AttoHash::new(Self::ID)
// (audionode.rs:871-876, 1242-1247, 1389-1394); the engine does the same for a bank's top-level node.
// Oversampler::new instead pings the ENCLOSED node with AttoHash::new(Self::ID) (oversample.rs:93-95).
template <class G> struct CtorPing {
    static FD_D void run(G& g) { uint64_t h = g.ping(true, G::ID); g.ping(false, h); }
};
template <class X> struct CtorPing<Oversampler<X>> {
    static FD_D void run(Oversampler<X>& g) { uint64_t h = g.x.ping(true, Oversampler<X>::ID); g.x.ping(false, h); }
};

template <class X> struct CtorPing<Resample<X>> {  // Resample::new resample.rs:230-232: same pattern
    static FD_D void run(Resample<X>& g) { uint64_t h = g.x.ping(true, Resample<X>::ID); g.x.ping(false, h); }
};

template <> struct CtorPing<PulseWave> {  // pulse() = An(PulseWave::new()): only the inner Pipe's constructor pinged
    static FD_D void run(PulseWave& g) {
        uint64_t h = g.pulse.ping(true, PulseWave::Inner::ID);
        g.pulse.ping(false, h);
    }
};

template <class G>
FD_D void lifecycle_body(float* slots, size_t stride, size_t first, size_t count, int op, double sr,
                         const uint64_t* seeds, const void* aux, float* ring, uint32_t ring_cap) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    size_t v = first + i;
    G g;
    Ctx ctx{static_cast<const Aux*>(aux), ring + v, ring_cap, stride, 0};
    // the "capacity too small" word sits right behind the bank's ring memory (fd_capi.hip allocates it)
    // (word 1 of that line holds the number of REAL voices: padding lanes keep default parameters and must not complain)
    if (G::RINGS > 0 && ring) {
        uint32_t* line = reinterpret_cast<uint32_t*>(ring + (size_t)G::RINGS * ring_cap * stride);
        if (v < line[1]) ctx.ring_need = line;
    }
    g.bind(ctx);
    {
        VLoad ld{slots + v, stride, 0};
        g.visit(ld);
    }
    if (op == 0) {
        g.init();
        g.update(sr);
        CtorPing<G>::run(g);
    } else if (op == 1) {
        g.update(sr);
    } else if (op == 2) {
        g.reset();
    } else {
        if (seeds) {
            g.ping(false, seeds[i]);  // AudioNode::set_seed audionode.rs:366-368
        } else {
            CtorPing<G>::run(g);
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

// ---- wide sums: branch-major blocks ---------------------------------------------------------------------------
// Reduce<N, X, OP> / MultiBus<N, X> over N >= 8 branches of one type (sumi / busi of oscillators -- additive synthesis; the reference's own `sine`
// bench is sumi::<U100>(sine_hz(100 (i + 1))), benches/benchmark.rs:4-10 --, of filters on a shared or on their own inputs).  Frame-major, the way
// render_body_frames walks every other graph, such a node keeps the state of all N branches in registers: 100 sines are 700 words per lane, the compiler
// parks them in scratch memory and the loop pays for it per frame (measured: 1.29 s per rendered second, whatever the bank size).  The reference's
// process() walks it the other way round (audionode.rs:2406-2462, 2123-2134): one 64-frame block of branch 0, then of branch 1 folded into it, ...
// The branches do not see each other, so branch-major and frame-major give every branch the same samples and every output frame the same left fold.
// Here: per 64-frame block and branch, load the branch's slots, render the block into registers (packed path, 32 frame pairs, rollback to the scalar
// path if a guard trips), fold into the accumulator, store the branch's state.  The accumulator lives in registers for whole process blocks and in LDS
// (one column per lane, [channel][frame][lane]: conflict-free, any frame index) for the ragged last block, the tick executor and the rollback.
//
// The sum need not be the whole graph: it may sit at the HEAD of a chain of Pipe / Unop nodes -- sumi(..) * 0.01 >> lowpass_hz(..) >> pan(..), an
// additive voice with its gain, filter and panner (WideSplit below).  The rest of the graph, its TAIL, is the same type with MultiPass in the sum's place:
// it walks the accumulated block frame-major, its state in registers for the whole launch (in the wave that finishes the fold), and produces the output.
// The sum's slots (and delay rings) come first in the graph's visit order, the tail's follow.
struct VCountWords {
    int n = 0;
    FD_D void f(float&, FieldKind, const char*) { n++; }
    FD_D void fi(float&, FieldKind, const char*, int) { n++; }
    FD_D void u32(uint32_t&, FieldKind, const char*) { n++; }
    FD_D void u64(uint64_t&, FieldKind, const char*) { n += 2; }
    FD_D void enter(int) {}
    FD_D void leave() {}
};
template <class G> struct WideSum { static constexpr bool value = false; };
template <int N_, class X, class OP_> struct WideSum<Reduce<N_, X, OP_>> {
    static constexpr bool value = N_ >= 8 && X::IN <= 4 && X::OUT >= 1 && X::OUT <= 2;
    static constexpr int N = N_;
    static constexpr bool BUS = false;
    using Branch = X;
    using OP = OP_;
};
template <int N_, class X> struct WideSum<MultiBus<N_, X>> {
    static constexpr bool value = N_ >= 8 && X::IN <= 4 && X::OUT >= 1 && X::OUT <= 2;
    static constexpr int N = N_;
    static constexpr bool BUS = true;   // tick folds from a zero frame: (0 + x0) + x1 .. (audionode.rs:2117-2121)
    using Branch = X;
    using OP = OpAdd;
};
// G = a wide sum (Head) followed by a Tail: Head at the left end of a spine of Pipe / Unop nodes
template <class G> struct WideSplit {
    static constexpr bool ok = WideSum<G>::value;
    using Head = G;
    using Tail = MultiPass<(G::OUT > 0 ? G::OUT : 1)>;
};
template <class X, class Y> struct WideSplit<Pipe<X, Y>> {
    static constexpr bool ok = WideSplit<X>::ok && Y::OUT >= 1 && Y::OUT <= 2;
    using Head = typename WideSplit<X>::Head;
    using Tail = Pipe<typename WideSplit<X>::Tail, Y>;
};
template <class X, class U> struct WideSplit<Unop<X, U>> {
    static constexpr bool ok = WideSplit<X>::ok;
    using Head = typename WideSplit<X>::Head;
    using Tail = Unop<typename WideSplit<X>::Tail, U>;
};
template <class G> struct WideGeom {  // channels a block tile holds: the sum's, or the tail's if it has more (mono sum >> pan)
    using H = typename WideSplit<G>::Head;
    static constexpr bool ok = WideSplit<G>::ok;
    static constexpr bool BARE = SameType<G, H>::v;
    static constexpr int NO = H::OUT, TO = G::OUT, C = NO > TO ? NO : TO;
    static constexpr int X = C - NO;   // channels the tail adds to the sum's: they live in a tile of their own next to the travelling ones (the chain)
};

// input sample of graph channel `ch` at frame t for voice v (the branches of a MultiBus share the graph's inputs, those of a Reduce have their own:
// branch i reads channels i X::IN ..).  Every branch re-reads the block's input rows; after the first branch they come from L2.
template <int LAYOUT>
FD_D float wide_in(const float* __restrict__ in, size_t ch, size_t t, size_t T, size_t V, size_t v, size_t gin, size_t fstride) {
    return LAYOUT == LAYOUT_VOICE_MINOR ? in[(ch * T + t) * V + v] : in[(v * gin + ch) * fstride + t];
}

// One 64-frame block of the branches [b0, b1) of the sum folded into the block's accumulator -- registers `acc` for a whole process block (FAST), else
// the LDS column `accl` ([(channel * 64 + frame) * 64]) in place.  `first`: branch 0 starts the fold.  `iv`: the voice whose input rows this lane reads,
// `live`: whether it may (lanes past the end of the bank read nothing).
template <class G, int MODE, int LAYOUT, bool FAST, class ACC>
FD_D void wide_fold(int b0, int b1, int K, float* __restrict__ slots, size_t stride, size_t V, size_t v, bool active, size_t iv, bool live,
                    const float* __restrict__ in, size_t T, size_t fstride, size_t t0, int size, const void* aux, float* ring, uint32_t ring_cap,
                    ACC& acc, float* accl) {
    using H = typename WideSplit<G>::Head;
    using WS = WideSum<H>;
    using X = typename WS::Branch;
    using OP = typename WS::OP;
    constexpr int NO = X::OUT, NI = X::IN;
    const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
#pragma unroll 1
    for (int i = b0; i < b1; i++) {
        X x;
        {
            Ctx ctx{static_cast<const Aux*>(aux), ring + v, ring_cap, stride, i * X::RINGS};
            x.bind(ctx);
            VLoad ld{slots + v, stride, i * K};
            x.visit(ld);
        }
        x.begin_block(size);
        if constexpr (FAST) {
            const X snap = x;
            v2f tmp[NO][32];
#pragma unroll
            for (int q = 0; q < 32; q++) {
                v2f po[NO], pi[NI > 0 ? NI : 1];
#pragma unroll
                for (int c = 0; c < NI; c++) {
                    const size_t ch = WS::BUS ? c : i * NI + c;
                    pi[c] = live ? v2f{wide_in<LAYOUT>(in, ch, t0 + 2 * q, T, V, iv, H::IN, fstride), wide_in<LAYOUT>(in, ch, t0 + 2 * q + 1, T, V, iv, H::IN, fstride)}
                                 : v2f{0.0f, 0.0f};
                }
                x.template step2<PH_SIMD>(pi, po);
#pragma unroll
                for (int c = 0; c < NO; c++) tmp[c][q] = po[c];
            }
            if (__builtin_expect(x.tripped(), 0)) {  // a packed-path shortcut left its exact domain: this branch's block again, scalar (the column is free: the accumulator is in registers)
                x = snap;
#pragma unroll 1
                for (int f = 0; f < 64; f++) {
                    float fo[NO], fi[NI > 0 ? NI : 1];
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = live ? wide_in<LAYOUT>(in, WS::BUS ? c : i * NI + c, t0 + f, T, V, iv, H::IN, fstride) : 0.0f;
                    x.template step<PH_SIMD>(fi, fo);
#pragma unroll
                    for (int c = 0; c < NO; c++) accl[(c * 64 + f) * 64] = fo[c];
                }
#pragma unroll
                for (int c = 0; c < NO; c++)
#pragma unroll
                    for (int q = 0; q < 32; q++) tmp[c][q] = v2f{accl[(c * 64 + 2 * q) * 64], accl[(c * 64 + 2 * q + 1) * 64]};
            }
            x.end_simd();
#pragma unroll
            for (int c = 0; c < NO; c++)
#pragma unroll
                for (int q = 0; q < 32; q++) acc[c][q] = i == 0 ? tmp[c][q] : OP::f(acc[c][q], tmp[c][q]);
        } else {
#pragma unroll 1
            for (int f = 0; f < size; f++) {
                float fo[NO], fi[NI > 0 ? NI : 1];
#pragma unroll
                for (int c = 0; c < NI; c++) fi[c] = live ? wide_in<LAYOUT>(in, WS::BUS ? c : i * NI + c, t0 + f, T, V, iv, H::IN, fstride) : 0.0f;
                if (f < full) {
                    x.template step<PH_SIMD>(fi, fo);
                } else {
                    if (MODE == MODE_PROCESS && f == full) x.end_simd();
                    x.template step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(fi, fo);
                }
#pragma unroll
                for (int c = 0; c < NO; c++) {
                    float* a = &accl[(c * 64 + f) * 64];
                    *a = i == 0 ? ((WS::BUS && MODE == MODE_TICK) ? 0.0f + fo[c] : fo[c]) : OP::f(*a, fo[c]);
                }
            }
            if (MODE == MODE_PROCESS && full == size) x.end_simd();
        }
        if (active) {
            VStore<false> st{slots + v, stride, i * K};
            x.visit(st);
        }
    }
}

// The finished fold of one block -> the graph's output: through the tail (if the sum is not the whole graph), then to HBM.  Voice-minor: straight from
// registers / the lane's column; planar: through the tile `tile0` ([channel][frame][voice]; channels past the sum's in `tilex`), transposed.  `acc` (FAST) or the lane's column of `tile0` hold the sum.
template <class G, int MODE, int LAYOUT, bool FAST, class ACC, class TAIL>
FD_D void wide_finish(TAIL& tail, ACC& acc, float* tile0, float* tilex, float* __restrict__ out, size_t T, size_t V, size_t v0, size_t v, bool active,
                      int lane, size_t fstride, size_t t0, int size) {
    using GEO = WideGeom<G>;
    constexpr int NO = GEO::NO, TO = GEO::TO;
    // channel c of the block: the sum's NO channels in the tile that travelled with the block, the channels a wider tail adds (mono sum >> pan) in `tilex`
    auto chp = [&](int c) { return c < NO ? tile0 + c * 4096 : tilex + (c - NO) * 4096; };
    auto colp = [&](int c) { return chp(c) + lane; };   // ... this lane's column of it: colp(c)[frame * 64]
    const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
    if constexpr (FAST) {
        // a finished frame pair of channel c leaves at once -- to HBM (voice-minor) or into the lane's column (planar: the tile is free, the sum is in
        // registers) --, so the block's output never sits in registers next to the accumulator (a chain of eight waves has 256 VGPRs per wave)
        auto emit = [&](int c, int q, v2f p) {
            if (LAYOUT == LAYOUT_VOICE_MINOR) {
                if (active) {
                    out[((size_t)c * T + t0 + 2 * q) * V + v] = p.x;
                    out[((size_t)c * T + t0 + 2 * q + 1) * V + v] = p.y;
                }
            } else {
                colp(c)[(2 * q) * 64] = p.x;
                colp(c)[(2 * q + 1) * 64] = p.y;
            }
        };
        if constexpr (GEO::BARE) {
#pragma unroll
            for (int c = 0; c < TO; c++)
#pragma unroll
                for (int q = 0; q < 32; q++) emit(c, q, acc[c][q]);
        } else {
            tail.begin_block(64);
            const TAIL snap = tail;
#pragma unroll
            for (int q = 0; q < 32; q++) {
                v2f pi[NO], po[TO];
#pragma unroll
                for (int c = 0; c < NO; c++) pi[c] = acc[c][q];
                tail.template step2<PH_SIMD>(pi, po);
#pragma unroll
                for (int c = 0; c < TO; c++) emit(c, q, po[c]);
            }
            if (__builtin_expect(tail.tripped(), 0)) {  // the tail's packed path left its exact domain: its block again, scalar, through the columns -- and out again, over what the packed pass wrote
                tail = snap;
#pragma unroll
                for (int c = 0; c < NO; c++)
#pragma unroll
                    for (int q = 0; q < 32; q++) {
                        colp(c)[(2 * q) * 64] = acc[c][q].x;
                        colp(c)[(2 * q + 1) * 64] = acc[c][q].y;
                    }
#pragma unroll 1
                for (int f = 0; f < 64; f++) {
                    float fi[NO], fo[TO];
#pragma unroll
                    for (int c = 0; c < NO; c++) fi[c] = colp(c)[(f) * 64];
                    tail.template step<PH_SIMD>(fi, fo);
#pragma unroll
                    for (int c = 0; c < TO; c++) colp(c)[(f) * 64] = fo[c];   // (frame f's inputs are consumed)
                }
                if (LAYOUT == LAYOUT_VOICE_MINOR && active) {
#pragma unroll 1
                    for (int f = 0; f < 64; f++)
#pragma unroll
                        for (int c = 0; c < TO; c++) out[((size_t)c * T + t0 + f) * V + v] = colp(c)[(f) * 64];
                }
            }
            tail.end_simd();
        }
    } else {
        if constexpr (!GEO::BARE) {
            tail.begin_block(size);
#pragma unroll 1
            for (int f = 0; f < size; f++) {
                float fi[NO], fo[TO];
#pragma unroll
                for (int c = 0; c < NO; c++) fi[c] = colp(c)[(f) * 64];
                if (f < full) {
                    tail.template step<PH_SIMD>(fi, fo);
                } else {
                    if (MODE == MODE_PROCESS && f == full) tail.end_simd();
                    tail.template step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(fi, fo);
                }
#pragma unroll
                for (int c = 0; c < TO; c++) colp(c)[(f) * 64] = fo[c];
            }
            if (MODE == MODE_PROCESS && full == size) tail.end_simd();
        }
        if (LAYOUT == LAYOUT_VOICE_MINOR && active) {
            for (int f = 0; f < size; f++)
#pragma unroll
                for (int c = 0; c < TO; c++) out[((size_t)c * T + t0 + f) * V + v] = colp(c)[(f) * 64];
        }
    }
    if (LAYOUT == LAYOUT_PLANAR) {  // tile [channel][frame][lane = voice] -> global [voice][channel][frame]: a lane writes 4 consecutive frames of one voice
        wave_sync();
        const int sub = lane >> 4, fr = (lane & 15) << 2;
        const bool vec = ((fstride & 3) == 0) && ((((uintptr_t)out) & 15) == 0) && (t0 + 64 <= fstride);
#pragma unroll
        for (int c = 0; c < TO; c++)
            for (int rr = 0; rr < 16; rr++) {
                const int vr = rr * 4 + sub;
                const size_t gv = v0 + vr;
                if (gv < V) {
                    float4 q4 = make_float4(chp(c)[(fr) * 64 + vr], chp(c)[(fr + 1) * 64 + vr], chp(c)[(fr + 2) * 64 + vr],
                                            chp(c)[(fr + 3) * 64 + vr]);
                    float* dst = out + (gv * TO + c) * fstride + t0 + fr;
                    if (vec && fr + 4 <= ((size + 3) & ~3)) {
                        *reinterpret_cast<float4*>(dst) = q4;
                    } else {
                        if (fr + 0 < size) dst[0] = q4.x;
                        if (fr + 1 < size) dst[1] = q4.y;
                        if (fr + 2 < size) dst[2] = q4.z;
                        if (fr + 3 < size) dst[3] = q4.w;
                    }
                }
            }
        wave_sync();
    }
}

template <class G> FD_D int wide_branch_words() {  // slot words of one branch (a constant after inlining)
    typename WideSum<typename WideSplit<G>::Head>::Branch probe;
    VCountWords c;
    probe.visit(c);
    return c.n;
}

// one wave per voice group: one-block launches and "pipe_split" 0 (the chain below renders everything else)
template <class G, int MODE, int LAYOUT, int WPB>
FD_D void render_body_wide(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, size_t T,
                           size_t fstride, const void* aux, float* ring, uint32_t ring_cap) {
    using GEO = WideGeom<G>;
    using H = typename WideSplit<G>::Head;
    using TAIL = typename WideSplit<G>::Tail;
    constexpr int N = WideSum<H>::N, NO = GEO::NO, C = GEO::C;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vpw = (LAYOUT == LAYOUT_VOICE_MINOR && fstride > 0 && fstride < 64) ? (int)fstride : 64;   // partially filled waves of small banks, as render_body_frames
    const size_t v0 = ((size_t)blockIdx.x * WPB + wib) * vpw;
    const size_t v = v0 + lane;
    const bool active = v < V && lane < vpw;
    if (v0 >= stride) return;
    // voice-minor: a lane past the end of the bank / of a partially filled wave has nothing to do (every LDS column is private to its lane, no
    // wave-level hand-over in that layout).  Planar: padding lanes (V <= v < stride) own private zeroed slot / ring columns and take part.
    if (LAYOUT == LAYOUT_VOICE_MINOR && !active) return;
    const size_t vin = v < V ? v : v0;  // planar: a padding lane reads a real voice's input rows (its samples are never stored)
    const int K = wide_branch_words<G>();
    __shared__ float acc_all[WPB * C * 64 * 64];
    float* tile0 = acc_all + (size_t)wib * C * 64 * 64;
    float* accl = tile0 + lane;
    TAIL tail;
    if constexpr (!GEO::BARE) {
        Ctx ctx{static_cast<const Aux*>(aux), ring + v, ring_cap, stride, N * WideSum<H>::Branch::RINGS};
        tail.bind(ctx);
        VLoad ld{slots + v, stride, N * K};
        tail.visit(ld);
    }
    for (size_t t0 = 0; t0 < T; t0 += 64) {
        const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
        if (MODE == MODE_PROCESS && size == 64) {
            v2f acc[NO][32];
            wide_fold<G, MODE, LAYOUT, true>(0, N, K, slots, stride, V, v, active, vin, true, in, T, fstride, t0, size, aux, ring, ring_cap, acc, accl);
            wide_finish<G, MODE, LAYOUT, true>(tail, acc, tile0, tile0 + NO * 4096, out, T, V, v0, v, active, lane, fstride, t0, size);
        } else {
            int none = 0;
            wide_fold<G, MODE, LAYOUT, false>(0, N, K, slots, stride, V, v, active, vin, true, in, T, fstride, t0, size, aux, ring, ring_cap, none, accl);
            wide_finish<G, MODE, LAYOUT, false>(tail, none, tile0, tile0 + NO * 4096, out, T, V, v0, v, active, lane, fstride, t0, size);
        }
    }
    if constexpr (!GEO::BARE) {
        if (active) {
            VStore<false> st{slots + v, stride, N * K};
            tail.visit(st);
        }
    }
}

// ... and the same sum with a voice group spread over W waves.  One wave per voice group takes as long for 64 instances as for 65 536; but the left fold
// fixes the ORDER of the additions, not who evaluates the branches: a voice group becomes a workgroup of W waves, wave w owns the branches
// [w N / W, (w + 1) N / W) and the blocks travel down the chain -- in round r wave w works on block r - w: it takes the block's accumulator tile from LDS
// as wave w - 1 left it, folds its own branches into it in order, and leaves it for wave w + 1; the last wave runs the tail and writes the output.  W
// blocks are in flight, block k lives in tile k mod W for its whole trip, so one workgroup barrier per round hands every tile on; the fill and drain of
// the chain cost W - 1 rounds per launch.  Same branch arithmetic, same fold, same slots traffic as render_body_wide -- bit-identical to it
// (tests/test_gpu_wide_sum.py renders every case through both).  W = 8 for mono sums of generators (8 tiles of 16 KB = 128 KB of LDS, one workgroup per
// CU, two waves per SIMD; a stereo tail adds ONE 16 KB tile for its second channel, the finishing wave's), 4 for stereo sums and for branches with inputs.
template <class G> struct WideChain {
    static constexpr bool on = WideSplit<G>::ok;
    static constexpr int W = !on ? 1 : (WideGeom<G>::NO == 1 && G::IN == 0 ? 8 : 4);  // (branches with inputs keep 64 more samples in flight per block: 4 waves leave each the whole register file)
};

template <class G, int MODE, int LAYOUT>
FD_D void render_body_wide_chain(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out,
                                 size_t T, size_t fstride, const void* aux, float* ring, uint32_t ring_cap) {
    if constexpr (WideSplit<G>::ok) {
        using GEO = WideGeom<G>;
        using H = typename WideSplit<G>::Head;
        using TAIL = typename WideSplit<G>::Tail;
        constexpr int N = WideSum<H>::N, NO = GEO::NO, C = GEO::C, W = WideChain<G>::W;
        const int lane = threadIdx.x & 63;
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // place in the chain
        const size_t v0 = (size_t)blockIdx.x * 64;
        const size_t v = v0 + lane;  // < stride: the slot / ring arrays are padded to whole voice groups (padding lanes own private zeroed columns)
        const bool active = v < V;
        const int K = wide_branch_words<G>();
        // The last wave also walks the tail: it gets fewer branches, by the tail's weight in branch units -- estimated from the state words of the
        // two (a gain, an SVF and a panner behind sines: 16 words against 7, measured as two branches' worth).  Any partition into consecutive runs
        // gives the same fold.
        int E = 0;
        if constexpr (!GEO::BARE) {
            TAIL probe;
            VCountWords c;
            probe.visit(c);
            E = (c.n + K / 2) / (K > 0 ? K : 1);
            E = E < 1 ? 1 : (E > N / W ? N / W : E);
        }
        const int b0r = w * (N + E) / W, b1r = (w + 1) * (N + E) / W;
        const int b0 = b0r < N ? b0r : N, b1 = b1r < N ? b1r : N;
        __shared__ float tiles[(W * NO + GEO::X) * 64 * 64];  // [tile][channel of the sum][frame][lane], then the channels only the tail has (the last wave's)
        float* tilex = tiles + W * NO * 64 * 64;
        TAIL tail;
        if constexpr (!GEO::BARE) {
            if (w == W - 1) {
                Ctx ctx{static_cast<const Aux*>(aux), ring + v, ring_cap, stride, N * WideSum<H>::Branch::RINGS};
                tail.bind(ctx);
                VLoad ld{slots + v, stride, N * K};
                tail.visit(ld);
            }
        }
        const size_t nblocks = (T + 63) / 64;
        for (size_t r = 0; r < nblocks + W - 1; r++) {
            __syncthreads();  // every tile moves one wave down the chain
            if (r < (size_t)w || r - w >= nblocks) continue;
            const size_t k = r - w, t0 = k * 64;
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            float* tile0 = tiles + (k % W) * (NO * 64 * 64);
            float* accl = tile0 + lane;  // this lane's column: accl[(c * 64 + frame) * 64]
            if (MODE == MODE_PROCESS && size == 64) {
                v2f acc[NO][32];
                if (w > 0) {
#pragma unroll
                    for (int c = 0; c < NO; c++)
#pragma unroll
                        for (int q = 0; q < 32; q++) acc[c][q] = v2f{accl[(c * 64 + 2 * q) * 64], accl[(c * 64 + 2 * q + 1) * 64]};
                }
                wide_fold<G, MODE, LAYOUT, true>(b0, b1, K, slots, stride, V, v, active, v, active, in, T, fstride, t0, size, aux, ring, ring_cap, acc, accl);
                if (w == W - 1) {
                    wide_finish<G, MODE, LAYOUT, true>(tail, acc, tile0, tilex, out, T, V, v0, v, active, lane, fstride, t0, size);
                } else {
#pragma unroll
                    for (int c = 0; c < NO; c++)
#pragma unroll
                        for (int q = 0; q < 32; q++) {
                            accl[(c * 64 + 2 * q) * 64] = acc[c][q].x;
                            accl[(c * 64 + 2 * q + 1) * 64] = acc[c][q].y;
                        }
                }
            } else {  // the ragged last block, the tick executor: the fold in place, in the tile
                int none = 0;
                wide_fold<G, MODE, LAYOUT, false>(b0, b1, K, slots, stride, V, v, active, v, active, in, T, fstride, t0, size, aux, ring, ring_cap, none, accl);
                if (w == W - 1) wide_finish<G, MODE, LAYOUT, false>(tail, none, tile0, tilex, out, T, V, v0, v, active, lane, fstride, t0, size);
            }
        }
        if constexpr (!GEO::BARE) {
            if (w == W - 1 && active) {
                VStore<false> st{slots + v, stride, N * K};
                tail.visit(st);
            }
        }
    }
}

// ---- the hot kernel ----------------------------------------------------------------------------------------
// WPB = waves per workgroup.  Four-wave workgroups are used whenever LDS allows: the dispatcher places the four
// waves of one workgroup on the four SIMDs of a CU, so a 65 536-voice bank (256 workgroups) lands exactly one wave
// per SIMD.  Single-wave workgroups do NOT spread evenly (measured with tools/census.hip: 1024 x 64-thread
// workgroups leave ~10 % of the SIMDs idle and double up as many), which stretches a VALU-bound launch.
template <class G, int MODE, int LAYOUT, int WPB>
FD_D void render_body_frames(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                             float* __restrict__ out, size_t T, size_t fstride, const void* aux, float* ring,
                             uint32_t ring_cap) {
    constexpr int NI = G::IN, NO = G::OUT;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave in block (wave-uniform -> SGPR)
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
        // wave-uniform base + lane offset: the per-frame address arithmetic runs on the scalar unit
        // (global_load/store saddr form), not in the VALU-bound instruction stream
        const float* inw = in + v0;
        float* outw = out + v0;
#define inv(i) inw[(i) + lane]
#define outv(i) outw[(i) + lane]
        // Input streams are software-pipelined: the 8 frames after the ones being computed are already in flight
        // (a lone wave per SIMD cannot hide a global-load round trip per frame any other way).
        float nx[NI > 0 ? NI : 1][8];
        auto fetch = [&](size_t tn) {  // branch-free: frames past the end re-read the last frame (never used)
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const size_t t = tn + k < T ? tn + k : T - 1;
                    nx[c][k] = inv(((size_t)c * T + t) * V);
                }
        };
        fetch(0);
        for (size_t t0 = 0; t0 < T; t0 += 64) {
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
            float fi[NI > 0 ? NI : 1], fo[NO];
            g.begin_block(size);
            const G snap = g;  // block-start registers, for the rollback below
            for (int i = 0; i < full; i += 8) {  // one SIMD item of the reference = 8 frames = 4 packed pairs
                float cu[NI > 0 ? NI : 1][8];
#pragma unroll
                for (int c = 0; c < NI; c++)
#pragma unroll
                    for (int k = 0; k < 8; k++) cu[c][k] = nx[c][k];
                fetch(t0 + i + 8);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const size_t t = t0 + i + k;
                    v2f pi[NI > 0 ? NI : 1], po[NO];
#pragma unroll
                    for (int c = 0; c < NI; c++) pi[c] = v2f{cu[c][k], cu[c][k + 1]};
                    g.template step2<PH_SIMD>(pi, po);
#pragma unroll
                    for (int c = 0; c < NO; c++) {
                        outv(((size_t)c * T + t) * V) = po[c].x;
                        outv(((size_t)c * T + t + 1) * V) = po[c].y;
                    }
                }
            }
            if (__builtin_expect(g.tripped(), 0)) {  // a packed-path shortcut left its exact domain: redo the block
                g = snap;
                for (int i = 0; i < full; i++) {
                    const size_t t = t0 + i;
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = inv(((size_t)c * T + t) * V);
                    g.template step<PH_SIMD>(fi, fo);
#pragma unroll
                    for (int c = 0; c < NO; c++) outv(((size_t)c * T + t) * V) = fo[c];
                }
            }
            if (MODE == MODE_PROCESS) g.end_simd();
            // Scalar paths (the remainder of a process block, every sample in tick mode) read their inputs at the
            // point of use: they are the slow paths, and this keeps the loop as simple as it can be.
            for (int i = full; i < size; i++) {
                const size_t t = t0 + i;
#pragma unroll
                for (int c = 0; c < NI; c++) fi[c] = inv(((size_t)c * T + t) * V);
                g.template step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(fi, fo);
#pragma unroll
                for (int c = 0; c < NO; c++) outv(((size_t)c * T + t) * V) = fo[c];
            }
            if (size > full) fetch(t0 + 64);  // the scalar frames were not consumed from nx: refill it for the next block
        }
#undef inv
#undef outv
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
FD_D void render_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                      float* __restrict__ out, size_t T, size_t fstride, const void* aux, float* ring,
                      uint32_t ring_cap) {
    if constexpr (WideSplit<G>::ok)  // a wide sum at the head of the graph: branch-major blocks (above)
        render_body_wide<G, MODE, LAYOUT, WPB>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
    else
        render_body_frames<G, MODE, LAYOUT, WPB>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
}

template <class G, int MODE, int LAYOUT, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_render(float* __restrict__ slots, size_t stride, size_t V,
                                                     const float* __restrict__ in, float* __restrict__ out,
                                                     size_t T, size_t fstride, const void* aux, float* ring,
                                                     uint32_t ring_cap) {
    render_body<G, MODE, LAYOUT, WPB>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
}

// ---- on-device voice scheduler: Sequencer semantics, one event per voice ---------------------------
// Reference: src/sequencer.rs, ReplayMode::None, no loop point: push :355-398, ready_to_active :584-605, process
// :838-951, tick :769-836, fade_in / fade_out :122-216, smooth5 / sine_ease math.rs:418-420,453-458.
// Voice v plays from ev[0][v] = start_time to ev[1][v] = end_time (seconds on the sequencer clock, f64 like the
// reference) with fade-in / fade-out times ev[2][v], ev[3][v] and curve fade[v] (0 = Fade::Power, 1 = Fade::Smooth).
// The kernel writes every voice's own faded contribution to [channel][frame][voice] (0 outside the event); the sum
// over voices is fdsp_sum_voices.  Exactly as in the reference, a unit is processed in the part of each 64-frame
// sequencer block that its event overlaps -- a SHORTER process() block at its start and end -- and the fade factors
// are re-derived per block from the f64 clock and accumulated in f32 inside the block.
FD_HD float smooth5f(float x) { return ((x * 6.0f - 15.0f) * x + 10.0f) * x * x * x; }
FD_HD float sine_easef(float x) {  // Bhaskara's approximation, math.rs:453-458
    constexpr float PI_F = (float)3.14159265358979323846, HALF_PI_F = (float)(3.14159265358979323846 * 0.5);
    constexpr float D = (float)(5.0 * 3.14159265358979323846 * 3.14159265358979323846);
    x = x * HALF_PI_F;
    return 16.0f * x * (PI_F - x) / (D - 4.0f * x * (PI_F - x));
}
FD_HD float fade_at(int ease, float x) { return ease == 0 ? sine_easef(x) : smooth5f(x); }
FD_HD long long round_index(double x) {  // `round(x) as usize`: half away from zero, negative / NaN -> 0
    double r = __builtin_round(x);
    return r > 0.0 ? (r < 4.0e18 ? (long long)r : (long long)4.0e18) : 0;
}

// ---- fused mix-down ("mode B" of SURVEY.md 8(d): on-device reduction of the voices to [channels][frames]) ------------
// The LAST stage of a voice group does not store its samples to HBM; it parks them in a small LDS tile
// [mix channel][frame of the chunk][voice] and, every MC frames, reads the tile back TRANSPOSED -- lane = (channel, frame,
// quarter of the group's voices) -- adds the 16 voices of its quarter one after the other, combines the four quarters
// through DPP and writes ONE float per (channel, frame): the group's partial mix.  The order is fixed and does not depend on
// the launch geometry:
//     partial(group) = (S0 + S1) + (S2 + S3),   Sq = ((x[16q] + x[16q+1]) + x[16q+2]) + ... + x[16q+15]
// (voices past the end of the bank count as +0.0).  k_mix_tree then adds the groups' partials in an aligned binary tree
// (an odd node at the end of a level passes through).  fdsp_sum_voices / fdsp_mix_stereo of a voice-out render use the same
// order, so the fused mix equals them bit for bit (tests/test_gpu_mix.py).
// Reference shape: the Panner / Reduce arithmetic of src/pan.rs:50-76, src/audionode.rs:2406-2462 over a bank of voices.
constexpr int MIX_NONE = 0, MIX_SUM = 1, MIX_PAN = 2;  // sum every output channel over the voices | pan a mono graph to stereo, then sum
constexpr int MIX_ROW = 68;  // floats per (channel, frame) row of a mix tile: 64 voices + 4 (16-byte runs stay aligned, lane-per-row b128 reads spread over the banks)
template <int NM, int GPW, int SUB>
struct MixGeom {  // frames per chunk: what fits the 32 KiB the pipeline's own tiles leave of a CU's 160 KiB of LDS
    static constexpr int fit = (31 * 1024) / GPW / (NM * MIX_ROW * 4);
    static constexpr int MC0 = fit >= 64 ? 64 : fit >= 32 ? 32 : fit >= 16 ? 16 : fit >= 8 ? 8 : 0;
    static constexpr int MC = MC0 > SUB ? SUB : MC0;
    static constexpr int FLOATS = NM * (MC > 0 ? MC : 1) * MIX_ROW;
    static constexpr bool ok = MC >= 8;
};
struct MixLane {   // what the last stage's wave knows about its mix tile
    float* tile;   // LDS: [tile channels][MC][MIX_ROW]; MIX_PAN keeps ONE channel (the mono samples) and pans when it flushes
    int col;       // this lane's column: its voice's, or the padding column 64 for lanes past the end of the bank
    // MIX_PAN: the equal-power weights (left, right; pan.rs:13-17) of the 16 voices of the quarter this lane adds up when it
    // flushes -- quarter = lane & 3 in every pass, so they are loaded once per launch (voices past the end of the bank: 0, 0) --
    // in registers (OL == 3), or, where 32 more registers would spill (the 14-wave time-split workgroup), in LDS: wlds[voice] (OL == 4)
    v2f w[16];
    const v2f* wlds;
};
FD_D float mix_quad(float x, int ctrl) {
    return u2f((uint32_t)(ctrl == 0 ? __builtin_amdgcn_update_dpp((int)f2u(x), (int)f2u(x), 0xB1, 0xF, 0xF, false)     // quad_perm [1,0,3,2]
                                    : __builtin_amdgcn_update_dpp((int)f2u(x), (int)f2u(x), 0x4E, 0xF, 0xF, false)));  // quad_perm [2,3,0,1]
}
// the chunk's first `nf` frames -> dst[channel * T + frame]; every lane of the wave takes part
template <int NM, int MC, bool ROLL = false>
FD_D void mix_flush(const float* tile, float* dst, size_t T, int nf, int lane) {
    constexpr int E = NM * MC;  // (channel, frame) entries of the tile
    constexpr int UF = ROLL ? 1 : 4;  // unroll factor of the pass loop
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the tile was written by other lanes of this wave (LDS operations of a wave execute in order)
    // ROLL: the passes stay a loop.  Unrolled, the ILP scheduling strategies the time-split kernels are built with hoist every pass's tile
    // reads to the top (214 VGPRs in the one-group kernel: fine at 8 waves per CU and 0.24 ms faster than the loop; 137 SPILLS in the
    // two-group one, whose 14 waves leave 128 registers each: there the loop is the faster form, profiles/r04_mix_bench_d.txt)
#pragma unroll UF
    for (int p = 0; p < (E * 4 + 63) / 64; p++) {
        const int idx = p * 64 + lane, e = idx >> 2, q = idx & 3;
        const bool on = (E * 4) % 64 == 0 || e < E;
        const float4* row = reinterpret_cast<const float4*>(tile + (on ? e : 0) * MIX_ROW + q * 16);
        const float4 a = row[0], b = row[1], c = row[2], d = row[3];
        float s = a.x;
        s += a.y; s += a.z; s += a.w;
        s += b.x; s += b.y; s += b.z; s += b.w;
        s += c.x; s += c.y; s += c.z; s += c.w;
        s += d.x; s += d.y; s += d.z; s += d.w;
        const float t = s + mix_quad(s, 0);
        const float u = t + mix_quad(t, 1);
        const int ch = e / MC, f = e % MC;
        if (on && q == 0 && f < nf) dst[(size_t)ch * T + f] = u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // ... and the next chunk's samples must not overtake these reads
}

// MIX_PAN: the tile holds the mono samples of MC frames; lane (frame, quarter) multiplies its quarter's 16 samples by their voices'
// weights -- left and right as one <2 x float> product, rounded like Panner::tick's `weight * sample` -- and adds them up one after the
// other (the same order per channel as mix_flush), the quarters through DPP; dst[frame] = left, dst[T + frame] = right.
template <int MC, bool WREG, bool ROLL = false>
FD_D void mix_flush_pan(const float* tile, float* dst, size_t T, int nf, int lane, const v2f* w, const v2f* wlds) {
    constexpr int E = MC;
    constexpr int UF = ROLL ? 1 : 4;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll UF
    for (int p = 0; p < (E * 4 + 63) / 64; p++) {
        const int idx = p * 64 + lane, e = idx >> 2, q = idx & 3;
        const bool on = (E * 4) % 64 == 0 || e < E;
        const float4* row = reinterpret_cast<const float4*>(tile + (on ? e : 0) * MIX_ROW + q * 16);
        v2f s = v2f{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; k++) {  // four samples at a time: the running sums are the only long-lived registers
            const float4 xq = row[k];
            const float x[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v2f wj = WREG ? w[4 * k + j] : wlds[q * 16 + 4 * k + j];
                const v2f t = splat2(x[j]) * wj;
                s = (k == 0 && j == 0) ? t : s + t;
            }
        }
        const float tl = s.x + mix_quad(s.x, 0), tr = s.y + mix_quad(s.y, 0);
        const float ul = tl + mix_quad(tl, 1), ur = tr + mix_quad(tr, 1);
        if (on && q == 0 && e < nf) {
            dst[e] = ul;
            dst[T + e] = ur;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// MIXE (k_render_events_mix, fdsp_bank_process_events_mix): the Sequencer's OUTPUT -- the sum of its events (sequencer.rs:838-951) -- leaves the
// launch instead of every event's own samples: a wave parks its block in an LDS tile [channel][64 frames][64 voices + 4] and flushes it as
// the voice group's partial mix (mix_flush, the mix-down's fixed order); `out` is then the partial buffer [groups][channels][T].  Every lane
// of a live group takes part in the flush; a padded voice of the last group is an event that never plays.
template <class G, int MODE, bool MIXE = false>
FD_D void render_events_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                             float* __restrict__ out, size_t T, const double* __restrict__ ev,
                             const int* __restrict__ fade, double time0, double sample_rate, const void* aux, float* ring,
                             uint32_t ring_cap) {
    constexpr int NI = G::IN, NO = G::OUT;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t v0 = ((size_t)blockIdx.x * 4 + wib) * 64;
    const size_t v = v0 + lane;
    if (v0 >= stride) return;
    const bool voice = v < V;
    if (!MIXE && !voice) return;
    __shared__ __attribute__((aligned(16))) float etile[MIXE ? 4 : 1][MIXE ? NO * 64 * MIX_ROW : 4];
    float* tile = &etile[MIXE ? wib : 0][0];
    float* part = MIXE ? out + (v0 / 64) * (size_t)NO * T : nullptr;  // this group's partial mix [channel][T]
    G g;
    Ctx ctx{static_cast<const Aux*>(aux), ring + v, ring_cap, stride, 0};
    g.bind(ctx);
    {
        VLoad ld{slots + v, stride, 0};
        g.visit(ld);
    }
    const double inf_ = __builtin_huge_val();
    const double e_start = voice ? ev[v] : inf_, e_end = voice ? ev[stride + v] : -inf_, e_fin = voice ? ev[2 * stride + v] : 0.0,
                 e_fout = voice ? ev[3 * stride + v] : 0.0;
    const int ease = (fade && voice) ? fade[v] : 1;
    const double sd = 1.0 / sample_rate;  // Sequencer::set_sample_rate :752-753
    double time = time0;
    const float* inv = in + v;
    float* outv = out + v;
    auto put = [&](int c, size_t t, float x) {  // frame t of the launch, channel c
        if constexpr (MIXE) tile[(c * 64 + (int)(t & 63)) * MIX_ROW + lane] = x;
        else outv[((size_t)c * T + t) * V] = x;
    };
    if (MODE == MODE_PROCESS) {
        for (size_t t0 = 0; t0 < T; t0 += 64) {
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            const double end_time = time + sd * (double)size;
            const double threshold = end_time - sd * 0.5;                       // ready_to_active :586
            bool act = e_start < threshold && !(e_end <= time + 0.5 * sd);      // :588, :861
            long long start_index = e_start <= time ? 0 : round_index((e_start - time) * sample_rate);
            long long end_index = size;
            if (!(e_end >= end_time)) {
                long long r = round_index((e_end - time) * sample_rate);
                end_index = r < size ? r : size;
            }
            act = act && end_index > start_index;
            const int n = act ? (int)(end_index - start_index) : 0;
            const int full = n & ~7;
            // fade_in :122-167
            bool fin_on = false;
            long long fin_end_i = 0;
            float fin_cur = 0.0f, fin_d = 0.0f;
            {
                const double fade_end = e_start + e_fin;
                if (act && e_fin > 0.0 && fade_end > time) {
                    fin_on = true;
                    fin_end_i = fade_end >= end_time ? end_index : round_index((fade_end - time) / sd);
                    fin_cur = (float)(((time + (double)start_index * sd) - e_start) / (fade_end - e_start));
                    fin_d = (float)(sd / e_fin);
                }
            }
            // fade_out :169-216
            bool fout_on = false;
            long long fout_i = 0;
            float fout_cur = 0.0f, fout_d = 0.0f;
            {
                const double fade_start = e_end - e_fout;
                if (act && e_fout > 0.0 && fade_start < end_time) {
                    fout_on = true;
                    fout_i = fade_start <= time ? 0 : round_index((fade_start - time) / sd);
                    fout_cur = (float)(((time + (double)fout_i * sd) - fade_start) / (e_end - fade_start));
                    fout_d = (float)(sd / e_fout);
                }
            }
            // Steady state -- every voice of the wave inside its event for the whole block, no fade running: the block is a
            // plain process(size) of the unit, so it takes the packed two-frame path of render_body (same arithmetic).
            const bool steady = act && start_index == 0 && end_index == size && !fin_on && !fout_on;
            if (__builtin_amdgcn_ballot_w64(steady) == __builtin_amdgcn_ballot_w64(true)) {
                g.begin_block(size);
                const G snap = g;
                for (int i = 0; i < full; i += 2) {
                    const size_t t = t0 + i;
                    v2f pi[NI > 0 ? NI : 1], po[NO];
#pragma unroll
                    for (int c = 0; c < NI; c++) pi[c] = v2f{inv[((size_t)c * T + t) * V], inv[((size_t)c * T + t + 1) * V]};
                    g.template step2<PH_SIMD>(pi, po);
#pragma unroll
                    for (int c = 0; c < NO; c++) {
                        put(c, t, po[c].x);
                        put(c, t + 1, po[c].y);
                    }
                }
                if (__builtin_expect(g.tripped(), 0)) {  // a packed-path shortcut left its exact domain: redo the block
                    g = snap;
                    for (int i = 0; i < full; i++) {
                        const size_t t = t0 + i;
                        float fi[NI > 0 ? NI : 1], fo[NO];
#pragma unroll
                        for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t) * V];
                        g.template step<PH_SIMD>(fi, fo);
#pragma unroll
                        for (int c = 0; c < NO; c++) put(c, t, fo[c]);
                    }
                }
                g.end_simd();
                for (int i = full; i < size; i++) {
                    const size_t t = t0 + i;
                    float fi[NI > 0 ? NI : 1], fo[NO];
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t) * V];
                    g.template step<PH_REM>(fi, fo);
#pragma unroll
                    for (int c = 0; c < NO; c++) put(c, t, fo[c]);
                }
                if constexpr (MIXE) mix_flush<NO, 64>(tile, part + t0, T, size, lane);
                time = end_time;
                continue;
            }
            if (act) {
                g.begin_block(n);
                if (full == 0) g.end_simd();
            }
            for (int i = 0; i < size; i++) {
                const size_t t = t0 + i;
                float fo[NO];
#pragma unroll
                for (int c = 0; c < NO; c++) fo[c] = 0.0f;
                if (act && i >= start_index && i < end_index) {
                    const int k = i - (int)start_index;  // index in the event's own sub-block buffer
                    float fi[NI > 0 ? NI : 1];
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t) * V];
                    if (k < full) g.template step<PH_SIMD>(fi, fo); else g.template step<PH_REM>(fi, fo);
                    if (k + 1 == full) g.end_simd();
                    if (fin_on && k < fin_end_i) {
                        const float e = fade_at(ease, fin_cur);
#pragma unroll
                        for (int c = 0; c < NO; c++) fo[c] *= e;
                        fin_cur += fin_d;
                    }
                    if (fout_on && k >= fout_i && k < end_index) {
                        const float e = fade_at(ease, 1.0f - fout_cur);
#pragma unroll
                        for (int c = 0; c < NO; c++) fo[c] *= e;
                        fout_cur += fout_d;
                    }
                }
#pragma unroll
                for (int c = 0; c < NO; c++) put(c, t, fo[c]);
            }
            if constexpr (MIXE) mix_flush<NO, 64>(tile, part + t0, T, size, lane);
            time = end_time;
        }
    } else {
        for (size_t t = 0; t < T; t++) {  // Sequencer::tick :769-836
            const double end_time = time + sd;
            const double threshold = end_time - sd * 0.5;
            const bool act = e_start < threshold && !(e_end <= time + 0.5 * sd);
            float fo[NO];
#pragma unroll
            for (int c = 0; c < NO; c++) fo[c] = 0.0f;
            if (act) {
                float fi[NI > 0 ? NI : 1];
#pragma unroll
                for (int c = 0; c < NI; c++) fi[c] = inv[((size_t)c * T + t) * V];
                g.template step<PH_TICK>(fi, fo);
                if (e_fin > 0.0) {
                    const float f = (float)((time - e_start) / ((e_start + e_fin) - e_start));
                    if (f < 1.0f) {
                        const float e = fade_at(ease, f);
#pragma unroll
                        for (int c = 0; c < NO; c++) fo[c] *= e;
                    }
                }
                if (e_fout > 0.0) {
                    const float f = (float)((time - (e_end - e_fout)) / (e_end - (e_end - e_fout)));
                    if (f > 0.0f) {
                        const float e = fade_at(ease, 1.0f - f);
#pragma unroll
                        for (int c = 0; c < NO; c++) fo[c] *= e;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NO; c++) put(c, t, fo[c]);
            if constexpr (MIXE) {
                if ((t & 63) == 63 || t + 1 == T) mix_flush<NO, 64>(tile, part + (t & ~(size_t)63), T, (int)(t & 63) + 1, lane);
            }
            time = end_time;
        }
    }
    if (voice) {
        VStore<false> st{slots + v, stride, 0};
        g.visit(st);
    }
}

template <class G, int MODE>
__global__ __launch_bounds__(256) void k_render_events_mix(float* __restrict__ slots, size_t stride, size_t V,
                                                          const float* __restrict__ in, float* __restrict__ part, size_t T,
                                                          const double* __restrict__ ev, const int* __restrict__ fade,
                                                          double time0, double sample_rate, const void* aux, float* ring,
                                                          uint32_t ring_cap) {
    static_assert(G::OUT * 64 * MIX_ROW * 4 * 4 <= 160 * 1024, "the block tiles of four waves must fit the CU's LDS");
    render_events_body<G, MODE, true>(slots, stride, V, in, part, T, ev, fade, time0, sample_rate, aux, ring, ring_cap);
}
template <class G, int MODE>
__global__ __launch_bounds__(256) void k_render_events(float* __restrict__ slots, size_t stride, size_t V,
                                                      const float* __restrict__ in, float* __restrict__ out, size_t T,
                                                      const double* __restrict__ ev, const int* __restrict__ fade,
                                                      double time0, double sample_rate, const void* aux, float* ring,
                                                      uint32_t ring_cap) {
    render_events_body<G, MODE>(slots, stride, V, in, out, T, ev, fade, time0, sample_rate, aux, ring, ring_cap);
}

#ifdef FD_PIPE_WPE        // A/B switch: tell the compiler how many waves per SIMD the pipeline kernel runs with
#define FD_PIPE_ATTR __attribute__((amdgpu_waves_per_eu(FD_PIPE_WPE, FD_PIPE_WPE)))
#else
#define FD_PIPE_ATTR
#endif
#ifndef FD_LP_ENABLE
#define FD_LP_ENABLE 1  // A/B switch (tools/build_variants.sh): 0 = always the generic SVF arithmetic
#endif
// Streams pass the caches by: every input line is read once and every output line written once, while a wavetable voice's table
// lines are re-read for 7-14 frames each (two groups of 64 voices x two tables = the 256 lines of a CU's L1).  Non-temporal feed loads
// and sample stores: config 4 9.14 -> 8.89 ms (either one alone 9.07-9.10; profiles/r04_ab_j_stream_policy.txt), the headline -- no
// tables -- within the noise (profiles/r03_ab20_21_small.txt).  A/B: 0 / 0 = the plain policy; 19 = sc0 sc1 nt stores.
#ifndef FD_PIPE_STORE_AUX
#define FD_PIPE_STORE_AUX 2     // cache-policy bits of the pipeline kernel's output stores (0 = none, 2 = nt, 19 = sc0 sc1 nt)
#endif
#ifndef FD_PIPE_PREFETCH
#define FD_PIPE_PREFETCH 1      // A/B switch: 0 = hand-over pairs read where they are used
#endif
// ---- multi-wave pipeline split of a Pipe chain ------------------------------------------------------------------
// At one voice-wave per SIMD (65 536 voices on 1024 SIMDs) a lone wave issues one instruction per ~4.7 cycles while
// the VALU could take one every ~2.5-3.3 (profiles/r01_ubench_valu.txt, r01_voice_sweep_*).  For graphs that are a
// chain A >> B >> C ... the chain is cut into S = 2 or 3 STAGES: each stage of the same 64 voices runs in its own
// wave, stage s one hand-over tile behind stage s-1, the cut's channel handed over through double-buffered LDS tiles.
// Every wave keeps its own part of the voice state in registers; the per-sample arithmetic of every node is
// untouched, so the output is bit-identical to the single-wave kernel -- only the issue slots of the SIMD are now fed
// by S waves whose dependent-instruction latencies overlap.
template <class T> struct Cost { static constexpr int v = 12; };  // rough VALU instructions per sample (cut placement only)
template <int N> struct Cost<Constant<N>> { static constexpr int v = 0; };
template <> struct Cost<Pass> { static constexpr int v = 0; };
template <> struct Cost<Sine> { static constexpr int v = 20; };
template <> struct Cost<SineFast> { static constexpr int v = 12; };
template <> struct Cost<FixedSvfLp> { static constexpr int v = 11; };
template <> struct Cost<Noise> { static constexpr int v = 10; };
template <> struct Cost<FixedSvf> { static constexpr int v = 16; };
template <int N> struct Cost<Moog<N>> { static constexpr int v = 130; };
template <int N> struct Cost<MoogFast<N>> { static constexpr int v = 130; };  // = Moog: FastOf keeps the stage plan (fd_jit.hip relies on it)
template <int S, int N> struct Cost<WaveSynth<S, N>> { static constexpr int v = 100; };
template <int S> struct Cost<PhaseSynth<S>> { static constexpr int v = 100; };
template <> struct Cost<PulseWave> { static constexpr int v = 210; };
template <> struct Cost<AdsrLive> { static constexpr int v = 80; };
template <> struct Cost<Shaper> { static constexpr int v = 60; };
template <> struct Cost<Panner> { static constexpr int v = 2; };
template <class X, class Y> struct Cost<Pipe<X, Y>> { static constexpr int v = Cost<X>::v + Cost<Y>::v; };
template <class X, class Y> struct Cost<Stack<X, Y>> { static constexpr int v = Cost<X>::v + Cost<Y>::v; };
template <class O, class X, class Y> struct Cost<Binop<O, X, Y>> { static constexpr int v = Cost<X>::v + Cost<Y>::v + 1; };
template <class X, class U> struct Cost<Unop<X, U>> { static constexpr int v = Cost<X>::v + 1; };

// Launch length (frames) from which a voice-minor launch takes the stage pipeline instead of the single-wave kernel.  Measured per
// kernel family at T = 16 ... 512 (tools/small_t_kernels.py, profiles/r04_small_t_kernels.txt): with a chain worth cutting (config 3: 59
// instructions per sample, config 4: 313) the pipeline wins from ONE 64-frame block on -- config 3, 65 536 voices: 13.5 vs 15.8 us at
// T = 64, 19.5 vs 24.8 at 128, 24.7 vs 32.4 at 192; config 4: 30.6 vs 33.8, 47.7 vs 58.4, 63.6 vs 82.6 -- while a light graph (config 2's
// noise >> biquad: 22) only gets its hand-over rounds paid back from four blocks on (7.1 vs 8.2 us at 64, 15.7 vs 15.2 at 256), and below a
// block the single wave is as fast or faster everywhere.  A/B: -DFD_PIPE_MIN_T=n overrides both.
template <class G> struct PipeMinT {
#ifdef FD_PIPE_MIN_T
    static constexpr int v = FD_PIPE_MIN_T;
#else
    static constexpr int v = Cost<G>::v >= 40 ? 64 : 256;
#endif
};

// Chain<G, HEAD>: the stages a graph can be cut into.  A Pipe chains its two sides.  A Binop whose left operand is a
// GENERATOR chain (no inputs) and which sits at the head of the graph -- so that its right operand reads the graph's own
// inputs -- counts the left operand's stages plus one TAIL stage (right operand + the operator):
//   ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * adsr_live(..) >> pan(p)  =  [saw stack] [moog] [* adsr] [pan]
template <class G, bool HEAD = true> struct Chain { static constexpr int N = 1; };
template <class X, class Y, bool HEAD> struct Chain<Pipe<X, Y>, HEAD> {
    static constexpr int N = Chain<X, HEAD>::N + Chain<Y, false>::N;
};
template <class O, class X, class Y> struct Chain<Binop<O, X, Y>, true> {
    static constexpr bool SPLIT = X::IN == 0 && X::OUT == Y::OUT;
    static constexpr int N = SPLIT ? Chain<X, true>::N + 1 : 1;
};

// A DF1 biquad is a chain of TWO stages cut at the seam of its own expression (fd_nodes.hpp BiquadT: feed-forward half | recurrence):
// behind a generator that can be split over time (Noise: a counter) the chain has three stages and small banks take the time-split
// kernel -- BASELINE config 2: the serial wave carries the recurrence alone.
template <uint64_t NODE_ID, bool HEAD> struct Chain<BiquadT<NODE_ID>, HEAD> { static constexpr int N = 2; };

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

// Weight<T>: the same estimate where a tolerance-mode twin is much cheaper than its exact type.  Cost places the cuts
// and must agree between G and FastOf<G> (one stage plan for both); Weight only decides which ROLES share a SIMD.
template <class T> struct Weight { static constexpr int v = Cost<T>::v; };
template <int N> struct Weight<MoogFast<N>> { static constexpr int v = 30; };
template <class X, class Y> struct Weight<Pipe<X, Y>> { static constexpr int v = Weight<X>::v + Weight<Y>::v; };
template <class X, class Y> struct Weight<Stack<X, Y>> { static constexpr int v = Weight<X>::v + Weight<Y>::v; };
template <class O, class X, class Y> struct Weight<Binop<O, X, Y>> { static constexpr int v = Weight<X>::v + Weight<Y>::v + 1; };
template <class X, class U> struct Weight<Unop<X, U>> { static constexpr int v = Weight<X>::v + 1; };

// HasSkip<T>: the node defines skip / skip2 (state advance without output) all the way down -- what a time-split stage
// needs of the stages it runs in several waves.  Leaves are detected; Pipe and Unop forward to their operands (their
// own skip2 members exist unconditionally and would only fail when instantiated); everything else: no.
template <class...> struct VoidT { using type = void; };
template <class T, class = void> struct LeafHasSkip { static constexpr bool v = false; };
template <class T> struct LeafHasSkip<T, typename VoidT<decltype(&T::template skip2<PH_SIMD>)>::type> { static constexpr bool v = true; };
template <class T> struct HasSkip { static constexpr bool v = LeafHasSkip<T>::v; };
template <class X, class Y> struct HasSkip<Pipe<X, Y>> { static constexpr bool v = HasSkip<X>::v && HasSkip<Y>::v; };
template <class X, class U> struct HasSkip<Unop<X, U>> { static constexpr bool v = HasSkip<X>::v; };
template <class O, class X, class Y> struct HasSkip<Binop<O, X, Y>> { static constexpr bool v = false; };
template <class X, class Y> struct HasSkip<Stack<X, Y>> { static constexpr bool v = HasSkip<X>::v && HasSkip<Y>::v; };

// ConstTail<X>: the trailing output channels of a Stack that are plain Constant nodes -- `(x | dc(a) | dc(b))` ends in two.
// Where a stage cut falls right behind such a Stack, those channels do not travel through the LDS hand-over tiles: the
// CONSUMER stage loads the Constants' slots itself and appends their values to the channels it reads (the identity
// `(x | c) >> y == x >> ((pass | c) >> y)`, values untouched).  Config 4's first cut shrinks from three channels to one, so
// its tiles are twice as long for the same LDS (half the rounds, barriers and per-tile bookkeeping) and the producer /
// consumer drop two LDS writes / reads per frame pair.
#ifndef FD_PIPE_ELIDE
#define FD_PIPE_ELIDE 1   // A/B switch: 0 = every channel of a cut travels through LDS
#endif
template <class X> struct ConstTail {
    static constexpr int K = 0;
    template <class W> static FD_D void visit(X& x, W& w) { w.on = false; x.visit(w); }
    static FD_D void fill2(const X&, v2f*) {}
    static FD_D void fill(const X&, float*) {}
};
template <class XL, int M> struct ConstTail<Stack<XL, Constant<M>>> {
    using S = Stack<XL, Constant<M>>;
    static constexpr int KL = ConstTail<XL>::K, K = KL + M;
    template <class W> static FD_D void visit(S& s, W& w) {  // Stack::visit order and slot numbering; only the tail's Constants enabled
        w.enter(0); ConstTail<XL>::visit(s.x, w); w.leave();
        w.enter(1); w.on = true; s.y.visit(w); w.on = false; w.leave();
    }
    static FD_D void fill2(const S& s, v2f* tail) {  // tail[0 .. K): XL's trailing constants, then these M
        ConstTail<XL>::fill2(s.x, tail);
        _Pragma("unroll") for (int i = 0; i < M; i++) tail[KL + i] = splat2(s.y.value[i]);
    }
    static FD_D void fill(const S& s, float* tail) {
        ConstTail<XL>::fill(s.x, tail);
        _Pragma("unroll") for (int i = 0; i < M; i++) tail[KL + i] = s.y.value[i];
    }
};

// Seg<G, A, B, HEAD>: the chain stages [A, B) of G, run on G's own state object.
//   in  = what the segment's first stage consumes: the node's own inputs when A == 0, else the hand-over channels;
//   gin = the node's own inputs (the graph's inputs for head-position nodes), valid in EVERY stage: a Binop tail reads
//         its right operand's inputs there.
// The primary template is a whole node.
template <class G, int A, int B, bool HEAD = true>
struct Seg {
    static_assert(A == 0 && B == 1, "a node that is not a chain is one stage");
    static constexpr int IN = G::IN, OUT = G::OUT, cost = Cost<G>::v, weight = Weight<G>::v;
    static constexpr bool USES_GIN = false;
    static constexpr bool HAS_SKIP = HasSkip<G>::v;
    template <int PH> static FD_D void step2(G& g, const v2f* in, const v2f*, v2f* out) { g.template step2<PH>(in, out); }
    template <int PH> static FD_D void step(G& g, const float* in, const float*, float* out) { g.template step<PH>(in, out); }
    template <int PH> static FD_D void skip2(G& g, const v2f* in) { g.template skip2<PH>(in); }  // time-split stages only
    template <int PH> static FD_D void skip(G& g, const float* in) { g.template skip<PH>(in); }
    static FD_D void begin(G& g, int n) { g.begin_block(n); }
    static FD_D void end(G& g) { g.end_simd(); }
    static FD_D bool tripped(const G& g) { return g.tripped(); }
    // visit ALL slots of g in G::visit order (slot numbering unchanged); only this segment's slots are enabled
    template <class W> static FD_D void visit(G& g, W& w) { w.on = true; g.visit(w); }
};
// BiquadT<ID>: stage 0 = the feed-forward half (state x1 x2, coefficients b0 b1 b2; packed, and skippable: the x history of a frame does
// not depend on evaluating the frames before it), stage 1 = the recurrence (y1 y2, a1 a2).  [0, 2) is the whole node.
template <uint64_t NODE_ID, int A, int B, bool HEAD>
struct Seg<BiquadT<NODE_ID>, A, B, HEAD> {
    using G = BiquadT<NODE_ID>;
    static_assert(0 <= A && A < B && B <= 2, "empty or out-of-range chain segment");
    static constexpr bool FF = A == 0, FB = B == 2;
    static constexpr int IN = 1, OUT = 1;
    static constexpr int cost = (FF ? 4 : 0) + (FB ? 10 : 0), weight = cost;   // (instructions per sample of the serial wave: fb's four count double)
    static constexpr bool USES_GIN = false;
    static constexpr bool HAS_SKIP = FF && !FB;
    template <int PH> static FD_D void step2(G& g, const v2f* in, const v2f*, v2f* out) {
        if constexpr (FF && FB) g.template step2<PH>(in, out);
        else if constexpr (FF) out[0] = g.ff2(in[0]);
        else { const float ya = g.fb(in[0].x); out[0] = v2f{ya, g.fb(in[0].y)}; }
    }
    template <int PH> static FD_D void step(G& g, const float* in, const float*, float* out) {
        if constexpr (FF && FB) g.template step<PH>(in, out);
        else if constexpr (FF) out[0] = g.ff(in[0]);
        else out[0] = g.fb(in[0]);
    }
    template <int PH> static FD_D void skip2(G& g, const v2f* in) {
        static_assert(HAS_SKIP, "only the feed-forward half can be skipped");
        g.ff_skip2(in[0]);
    }
    template <int PH> static FD_D void skip(G& g, const float* in) {
        static_assert(HAS_SKIP, "only the feed-forward half can be skipped");
        g.ff_skip(in[0]);
    }
    static FD_D void begin(G&, int) {}
    static FD_D void end(G&) {}
    static FD_D bool tripped(const G&) { return false; }
    template <class W> static FD_D void visit(G& g, W& w) {  // BiquadT::visit order; each half sees its own slots
        w.on = FB; w.f(g.a1, PARAM, "a1"); w.f(g.a2, PARAM, "a2");
        w.on = FF; w.f(g.b0, PARAM, "b0"); w.f(g.b1, PARAM, "b1"); w.f(g.b2, PARAM, "b2");
        w.f(g.x1, STATE, "x1"); w.f(g.x2, STATE, "x2");
        w.on = FB; w.f(g.y1, STATE, "y1"); w.f(g.y2, STATE, "y2");
    }
};
// The FIXED forms of the filters that hold a Biquad (butterpass_hz(f), resonator_hz(f, q): coefficients derived once, biquad.rs:227-380) cut
// the same way -- the holder's own parameters travel with the recurrence half.  (The forms with a cutoff / centre input stay one stage: an input
// may move the coefficients at any sample.)
template <class G, int A, int B>
struct HeldBiquadSeg {
    using BS = Seg<Biquad, A, B, false>;
    static constexpr bool FF = BS::FF, FB = BS::FB;
    static constexpr int IN = 1, OUT = 1, cost = BS::cost, weight = BS::weight;
    static constexpr bool USES_GIN = false, HAS_SKIP = BS::HAS_SKIP;
    template <int PH> static FD_D void step2(G& g, const v2f* in, const v2f* gin, v2f* out) {
        if constexpr (FF && FB) g.template step2<PH>(in, out); else BS::template step2<PH>(g.b, in, gin, out);
    }
    template <int PH> static FD_D void step(G& g, const float* in, const float* gin, float* out) {
        if constexpr (FF && FB) g.template step<PH>(in, out); else BS::template step<PH>(g.b, in, gin, out);
    }
    template <int PH> static FD_D void skip2(G& g, const v2f* in) { BS::template skip2<PH>(g.b, in); }
    template <int PH> static FD_D void skip(G& g, const float* in) { BS::template skip<PH>(g.b, in); }
    static FD_D void begin(G&, int) {}
    static FD_D void end(G&) {}
    static FD_D bool tripped(const G&) { return false; }
};
template <bool HEAD> struct Chain<ButterLowpass<1>, HEAD> { static constexpr int N = 2; };
template <bool HEAD> struct Chain<Resonator<1>, HEAD> { static constexpr int N = 2; };
template <int A, int B, bool HEAD>
struct Seg<ButterLowpass<1>, A, B, HEAD> : HeldBiquadSeg<ButterLowpass<1>, A, B> {
    using H = HeldBiquadSeg<ButterLowpass<1>, A, B>;
    template <class W> static FD_D void visit(ButterLowpass<1>& g, W& w) {  // ButterLowpass::visit order
        w.on = H::FB; w.f(g.cutoff, PARAM, "cutoff"); w.f(g.sr, COEF, "sample_rate"); w.f(g.b.a1, COEF, "a1"); w.f(g.b.a2, COEF, "a2");
        w.on = H::FF; w.f(g.b.b0, COEF, "b0"); w.f(g.b.b1, COEF, "b1"); w.f(g.b.b2, COEF, "b2"); w.f(g.b.x1, STATE, "x1"); w.f(g.b.x2, STATE, "x2");
        w.on = H::FB; w.f(g.b.y1, STATE, "y1"); w.f(g.b.y2, STATE, "y2");
    }
};
template <int A, int B, bool HEAD>
struct Seg<Resonator<1>, A, B, HEAD> : HeldBiquadSeg<Resonator<1>, A, B> {
    using H = HeldBiquadSeg<Resonator<1>, A, B>;
    template <class W> static FD_D void visit(Resonator<1>& g, W& w) {  // Resonator::visit order
        w.on = H::FB; w.f(g.center, PARAM, "center"); w.f(g.q, PARAM, "q"); w.f(g.sr, COEF, "sample_rate"); w.f(g.b.a1, COEF, "a1"); w.f(g.b.a2, COEF, "a2");
        w.on = H::FF; w.f(g.b.b0, COEF, "b0"); w.f(g.b.b1, COEF, "b1"); w.f(g.b.b2, COEF, "b2"); w.f(g.b.x1, STATE, "x1"); w.f(g.b.x2, STATE, "x2");
        w.on = H::FB; w.f(g.b.y1, STATE, "y1"); w.f(g.b.y2, STATE, "y2");
    }
};
static_assert(Chain<Pipe<Noise, ButterLowpass<1>>>::N == 3 && Chain<Pipe<Noise, Resonator<1>>>::N == 3 && Chain<ButterLowpass<2>>::N == 1, "the fixed biquad holders cut like the biquad");
template <class X, class Y, int A, int B, bool HEAD>
struct Seg<Pipe<X, Y>, A, B, HEAD> {
    using G = Pipe<X, Y>;
    static constexpr int NX = Chain<X, HEAD>::N, NY = Chain<Y, false>::N;
    static_assert(0 <= A && A < B && B <= NX + NY, "empty or out-of-range chain segment");
    static constexpr bool HX = A < NX, HY = B > NX;  // the segment has stages inside x / inside y
    using SX = Seg<X, HX ? A : 0, HX ? (B < NX ? B : NX) : NX, HEAD>;
    using SY = Seg<Y, HY ? (A > NX ? A - NX : 0) : 0, HY ? B - NX : NY, false>;
    // a cut right behind x = a Stack with trailing Constants: they stay out of the hand-over (ConstTail)
    static constexpr int KX = ConstTail<X>::K;
    static constexpr bool ELIDE = FD_PIPE_ELIDE != 0 && KX > 0 && KX < X::OUT && NX == 1;
    static constexpr bool ELIDE_OUT = ELIDE && HX && !HY && B == NX;  // this segment ends with x: its trailing constants are not handed over
    static constexpr bool ELIDE_IN = ELIDE && !HX && A == NX;          // this segment starts with y: it supplies x's trailing constants itself
    static constexpr int IN = HX ? SX::IN : (ELIDE_IN ? X::OUT - KX : SY::IN), OUT = HY ? SY::OUT : (ELIDE_OUT ? X::OUT - KX : SX::OUT);
    static constexpr int cost = (HX ? SX::cost : 0) + (HY ? SY::cost : 0);
    static constexpr int weight = (HX ? SX::weight : 0) + (HY ? SY::weight : 0);
    static constexpr bool USES_GIN = HX && SX::USES_GIN;
    static constexpr bool HAS_SKIP = !(HX && HY) && (HX ? SX::HAS_SKIP : SY::HAS_SKIP);  // skip is defined for one-stage segments
    template <int PH> static FD_D void step2(G& g, const v2f* in, const v2f* gin, v2f* out) {
        if constexpr (HX && HY) { v2f t[SX::OUT > 0 ? SX::OUT : 1]; SX::template step2<PH>(g.x, in, gin, t); SY::template step2<PH>(g.y, t, nullptr, out); }
        else if constexpr (ELIDE_OUT) {
            v2f t[X::OUT];
            SX::template step2<PH>(g.x, in, gin, t);
            _Pragma("unroll") for (int c = 0; c < OUT; c++) out[c] = t[c];
        } else if constexpr (ELIDE_IN) {
            v2f t[X::OUT];
            _Pragma("unroll") for (int c = 0; c < IN; c++) t[c] = in[c];
            ConstTail<X>::fill2(g.x, t + IN);
            SY::template step2<PH>(g.y, t, nullptr, out);
        }
        else if constexpr (HX) SX::template step2<PH>(g.x, in, gin, out);
        else SY::template step2<PH>(g.y, in, nullptr, out);
    }
    template <int PH> static FD_D void step(G& g, const float* in, const float* gin, float* out) {
        if constexpr (HX && HY) { float t[SX::OUT > 0 ? SX::OUT : 1]; SX::template step<PH>(g.x, in, gin, t); SY::template step<PH>(g.y, t, nullptr, out); }
        else if constexpr (ELIDE_OUT) {
            float t[X::OUT];
            SX::template step<PH>(g.x, in, gin, t);
            _Pragma("unroll") for (int c = 0; c < OUT; c++) out[c] = t[c];
        } else if constexpr (ELIDE_IN) {
            float t[X::OUT];
            _Pragma("unroll") for (int c = 0; c < IN; c++) t[c] = in[c];
            ConstTail<X>::fill(g.x, t + IN);
            SY::template step<PH>(g.y, t, nullptr, out);
        }
        else if constexpr (HX) SX::template step<PH>(g.x, in, gin, out);
        else SY::template step<PH>(g.y, in, nullptr, out);
    }
    template <int PH> static FD_D void skip2(G& g, const v2f* in) {
        static_assert(!(HX && HY), "skip is defined for one-stage segments");
        if constexpr (HX) SX::template skip2<PH>(g.x, in); else SY::template skip2<PH>(g.y, in);
    }
    template <int PH> static FD_D void skip(G& g, const float* in) {
        static_assert(!(HX && HY), "skip is defined for one-stage segments");
        if constexpr (HX) SX::template skip<PH>(g.x, in); else SY::template skip<PH>(g.y, in);
    }
    static FD_D void begin(G& g, int n) {
        if constexpr (HX) SX::begin(g.x, n);
        if constexpr (HY) SY::begin(g.y, n);
    }
    static FD_D void end(G& g) {
        if constexpr (HX) SX::end(g.x);
        if constexpr (HY) SY::end(g.y);
    }
    static FD_D bool tripped(const G& g) {
        bool t = false;
        if constexpr (HX) t = t || SX::tripped(g.x);
        if constexpr (HY) t = t || SY::tripped(g.y);
        return t;
    }
    template <class W> static FD_D void visit(G& g, W& w) {
        if constexpr (HX) SX::visit(g.x, w); else if constexpr (ELIDE_IN) ConstTail<X>::visit(g.x, w); else { w.on = false; g.x.visit(w); }
        if constexpr (HY) SY::visit(g.y, w); else { w.on = false; g.y.visit(w); }
    }
};
// Binop at the head of the graph with a generator chain on the left: stages 0 .. NX-1 are X's, stage NX is the tail
// (y and the operator).  Same arithmetic as Binop::step2 (x, then y, then O::f), whichever waves run the parts.
template <class O, class X, class Y, int A, int B>
struct Seg<Binop<O, X, Y>, A, B, true> {
    using G = Binop<O, X, Y>;
    static constexpr bool SPLIT = Chain<G, true>::SPLIT;
    static constexpr int NX = SPLIT ? Chain<X, true>::N : 0;
    static_assert(SPLIT ? (0 <= A && A < B && B <= NX + 1) : (A == 0 && B == 1), "empty or out-of-range chain segment");
    static constexpr bool HX = SPLIT ? A < NX : true;     // has stages of x
    static constexpr bool TAIL = SPLIT ? B == NX + 1 : true;  // has y and the operator
    using SX = Seg<X, (SPLIT && HX) ? A : 0, SPLIT ? (HX ? (B < NX ? B : NX) : NX) : Chain<X, true>::N, true>;
    static constexpr int IN = SPLIT ? (A == 0 ? G::IN : Seg<X, 0, (A > 0 ? A : 1), true>::OUT) : G::IN;
    static constexpr int OUT = TAIL ? G::OUT : SX::OUT;
    static constexpr int cost = (HX ? SX::cost : 0) + (TAIL ? Cost<Y>::v + 1 : 0);
    static constexpr int weight = (HX ? SX::weight : 0) + (TAIL ? Weight<Y>::v + 1 : 0);
    static constexpr bool USES_GIN = SPLIT && TAIL && Y::IN > 0;
    static constexpr bool HAS_SKIP = SPLIT && !TAIL && SX::HAS_SKIP;  // stages of the generator operand forward; the tail stage has none
    template <int PH> static FD_D void skip2(G& g, const v2f* in) {
        static_assert(SPLIT && !TAIL, "skip is defined for the generator operand's stages");
        SX::template skip2<PH>(g.x, in);
    }
    template <int PH> static FD_D void skip(G& g, const float* in) {
        static_assert(SPLIT && !TAIL, "skip is defined for the generator operand's stages");
        SX::template skip<PH>(g.x, in);
    }
    template <int PH> static FD_D void step2(G& g, const v2f* in, const v2f* gin, v2f* out) {
        if constexpr (!SPLIT) {
            g.template step2<PH>(in, out);
        } else if constexpr (!TAIL) {
            SX::template step2<PH>(g.x, in, gin, out);
        } else {
            v2f tx[X::OUT], ty[Y::OUT];
            if constexpr (HX) SX::template step2<PH>(g.x, in, gin, tx);
            else { _Pragma("unroll") for (int c = 0; c < X::OUT; c++) tx[c] = in[c]; }
            g.y.template step2<PH>(gin, ty);  // X::IN == 0: the binop's inputs are y's inputs
#pragma unroll
            for (int c = 0; c < Y::OUT; c++) out[c] = O::f(tx[c], ty[c]);
        }
    }
    template <int PH> static FD_D void step(G& g, const float* in, const float* gin, float* out) {
        if constexpr (!SPLIT) {
            g.template step<PH>(in, out);
        } else if constexpr (!TAIL) {
            SX::template step<PH>(g.x, in, gin, out);
        } else {
            float tx[X::OUT], ty[Y::OUT];
            if constexpr (HX) SX::template step<PH>(g.x, in, gin, tx);
            else { _Pragma("unroll") for (int c = 0; c < X::OUT; c++) tx[c] = in[c]; }
            g.y.template step<PH>(gin, ty);
#pragma unroll
            for (int c = 0; c < Y::OUT; c++) out[c] = O::f(tx[c], ty[c]);
        }
    }
    static FD_D void begin(G& g, int n) {
        if constexpr (HX) SX::begin(g.x, n);
        if constexpr (TAIL) g.y.begin_block(n);
    }
    static FD_D void end(G& g) {
        if constexpr (HX) SX::end(g.x);
        if constexpr (TAIL) g.y.end_simd();
    }
    static FD_D bool tripped(const G& g) {
        bool t = false;
        if constexpr (HX) t = t || SX::tripped(g.x);
        if constexpr (TAIL) t = t || g.y.tripped();
        return t;
    }
    template <class W> static FD_D void visit(G& g, W& w) {  // Binop::visit order: x, then y
        if constexpr (HX) SX::visit(g.x, w); else { w.on = false; g.x.visit(w); }
        w.on = TAIL;
        g.y.visit(w);
    }
};

// Waves of the pipeline kernel's workgroup: 4 voice groups x (loader + S compute stages)
template <int NI, int S>
struct PipeGeom {
    static constexpr int WAVES = 4 * (S + (NI > 0 ? 1 : 0));
    static constexpr bool ok = WAVES <= 16 && (NI > 0 || S >= 2);
};

// Tile geometry of a plan.  A tile is SUB frames of 64 voices.  LDS holds, for 4 voice groups: the feed ring (NI input
// channels x D tiles; D = 2, or S + 1 when a later stage reads the graph's inputs -- a Binop tail -- so that the loader's
// tile survives until that stage has used it) and the double-buffered hand-over tiles of the S - 1 cuts (W channels each,
// W = the widest cut).  One channel-tile of the 4 groups is 1 KiB * SUB; the budget is 128 KiB.
template <class G, int S, int K1, int K2, int GPW = 4>
struct PipeTiles {
    static constexpr int N = Chain<G>::N, NI = G::IN;
    using S0 = Seg<G, 0, S == 1 ? N : K1>;
    using S1 = Seg<G, S == 1 ? 0 : K1, S <= 2 ? N : K2>;   // unused when S == 1
    using S2 = Seg<G, S <= 2 ? 0 : K2, N>;                 // unused when S <= 2
    static constexpr int W1 = S >= 2 ? S0::OUT : 0, W2 = S >= 3 ? S1::OUT : 0;
    static constexpr int W = W1 > W2 ? W1 : W2;                                   // hand-over channels per cut (widest)
    static constexpr bool LATE_GIN = (S >= 2 && S1::USES_GIN) || (S >= 3 && S2::USES_GIN);
    static constexpr int D = NI > 0 ? (LATE_GIN ? S + 1 : 2) : 0;                 // feed ring depth
    static constexpr int UNITS4 = NI * D + 2 * W * (S - 1);                       // channel-tiles (of 4 voice groups)
    // a workgroup of 2 or 1 voice groups (heavy graphs on small banks) spends the same LDS on tiles 2 / 4 times as long:
    // half / a quarter of the hand-over rounds and barriers
    static constexpr int UNITS = UNITS4 <= 16 ? (UNITS4 * GPW + 3) / 4 : UNITS4;
    static constexpr int SUB = UNITS <= 2 ? 64 : UNITS <= 4 ? 32 : UNITS <= 8 ? 16 : UNITS <= 16 ? 8 : 0;  // 0 = does not fit
    static constexpr bool ok = SUB >= 8 && PipeGeom<NI, S>::ok;
};

// Where to cut: S stages (2 or 3) with cut points K1 < K2 chosen to minimise the most expensive stage; every stage must
// carry real work and the tiles must fit.
struct PipePlan { int S, K1, K2; };
template <class G, int I = 0>
constexpr void pipe_plan_fill(int* cost) {
    if constexpr (I < Chain<G>::N) {
        cost[I] = Seg<G, I, I + 1>::cost;
        pipe_plan_fill<G, I + 1>(cost);
    }
}
template <class G, int K1 = 1, int K2 = 2>
constexpr void pipe_plan_fits(bool (*fit2)[32], bool (*fit3)[32][32]) {  // which cut sets have tiles that fit
    constexpr int N = Chain<G>::N;
    if constexpr (K1 < N) {
        if constexpr (K2 == K1 + 1) (*fit2)[K1] = PipeTiles<G, 2, K1, N>::ok;
        if constexpr (K2 < N) {
            (*fit3)[K1][K2] = PipeTiles<G, 3, K1, K2>::ok;
            pipe_plan_fits<G, K1, K2 + 1>(fit2, fit3);
        } else {
            pipe_plan_fits<G, K1 + 1, K1 + 2>(fit2, fit3);
        }
    }
}
template <class G>
constexpr PipePlan pipe_plan(int want) {  // want: 0 = best plan, 1 / 2 / 3 = at most one / exactly two / three compute stages
    constexpr int N = Chain<G>::N;
    static_assert(N <= 32, "chain too long");
    // a graph with inputs always gets the loader wave (S = 1 if it cannot or need not be cut)
    const PipePlan fallback = G::IN > 0 && PipeTiles<G, 1, N, N>::ok ? PipePlan{1, N, N} : PipePlan{0, 0, 0};
    if (N < 2 || G::RINGS != 0 || want == 1) return fallback;
    int cost[32] = {0};
    bool fit2[32] = {false}, fit3[32][32] = {{false}};
    pipe_plan_fill<G>(cost);
    pipe_plan_fits<G>(&fit2, &fit3);
    auto sum = [&](int a, int b) { int t = 0; for (int i = a; i < b; i++) t += cost[i]; return t; };
    constexpr int MIN_STAGE = 8;
    PipePlan best = fallback;
    int best_worst = 1 << 30;
    if (want != 3)
        for (int k = 1; k < N; k++) {
            int p = sum(0, k), q = sum(k, N), w = p > q ? p : q;
            if (fit2[k] && p >= MIN_STAGE && q >= MIN_STAGE && w < best_worst) { best = PipePlan{2, k, N}; best_worst = w; }
        }
    // three stages: on request, or for heavy graphs (>= 150 instructions per sample) when they cut the heaviest stage
    // by a third or more.  (On config 3 -- 59 instructions, two stages leave 36 -- the third wave's hand-over traffic and
    // barriers cost more than it hides: 5.8 vs 5.4 ms.)
    const int two_worst = best_worst, total = sum(0, N);
    if (want != 2)
        for (int k1 = 1; k1 < N; k1++)
            for (int k2 = k1 + 1; k2 < N; k2++) {
                int p = sum(0, k1), q = sum(k1, k2), r = sum(k2, N), w = p > q ? (p > r ? p : r) : (q > r ? q : r);
                const bool worth = want == 3 || (total >= 150 && (two_worst == (1 << 30) || 3 * w <= 2 * two_worst));
                if (fit3[k1][k2] && worth && p >= MIN_STAGE && q >= MIN_STAGE && r >= MIN_STAGE && w < best_worst) {
                    best = PipePlan{3, k1, k2};
                    best_worst = w;
                }
            }
    return best;
}

// One stage's work on one tile: frames [lo, hi) of the block that starts at t0 (size / full as in
// AudioNode::process: `full` frames of packed SIMD items, end_simd, then the remainder path).
// Graph inputs come from the feed tile `fin` (written by the loader wave), never from HBM directly: the FIRST stage
// consumes them as its inputs, a later stage that holds a Binop tail reads them as `gin`.
// Hand-over tiles: [channel][frame pair][lane] (v2f).
// FS = floats per frame row of the feed tile (64, or 65 when the loader fills it by transposing planar rows); OL = where
// the LAST stage puts its samples: 0 = HBM, voice-minor; 1 = an LDS tile [channel][frame][FS] that the storer wave of the
// planar pipeline transposes out.
// OL = 2 / 3 (fused mix-down, MIX_SUM / MIX_PAN): the samples go to the wave's mix tile `mx` and leave as the group's partial mix,
// MC frames at a time (mix_flush); `outw` is then the group's partial row [channel][T] and EVERY lane of the wave runs the stage.
// StageCfg: the compile-time shape of one pipe_stage call -- tile length, hand-over width, position in the chain, where the samples go.
template <int SUB_, int W_, bool FIRST_, bool LAST_, int FS_ = 64, int OL_ = 0, bool PF_ = (FD_PIPE_PREFETCH != 0), int MC_ = 0, bool MROLL_ = false>
struct StageCfg {
    static constexpr int SUB = SUB_, W = W_, FS = FS_, OL = OL_, MC = MC_;
    static constexpr bool FIRST = FIRST_, LAST = LAST_, PF = PF_, MROLL = MROLL_;
};
template <class SG, class G, int MODE, class CFG>
FD_D void pipe_stage(G& g, int h, size_t t0, int size, int full, size_t T, size_t V, int lane, float* outw,
                     const float* fin, v2f (*hin)[CFG::SUB / 2][64], v2f (*hout)[CFG::SUB / 2][64], const MixLane* mx = nullptr) {
    constexpr int SUB = CFG::SUB, W = CFG::W, FS = CFG::FS, OL = CFG::OL, MC = CFG::MC;
    constexpr bool FIRST = CFG::FIRST, LAST = CFG::LAST, PF = CFG::PF, MROLL = CFG::MROLL;
    constexpr int NI = SG::IN, NO = SG::OUT, NG = G::IN;
    constexpr bool GIN = !FIRST && SG::USES_GIN;
    constexpr bool MIXO = OL >= 2;
    constexpr int NM = OL >= 3 ? 2 : NO;  // mix channels
    static_assert(!MIXO || (LAST && MC >= 8 && (MC & (MC - 1)) == 0), "mix-down: last stage, chunks of 8 .. 64 frames");
    static_assert(OL < 3 || NO == 1, "the pan mix-down takes a mono graph");
    static_assert(LAST || NO <= W, "hand-over tile too narrow");
    static_assert(FIRST || NI <= W, "hand-over tile too narrow");
    const int lo = h * SUB;
    const int hi = lo + SUB < size ? lo + SUB : size;
    const int shi = hi < full ? hi : full;  // end of the packed part inside this tile
    // Voice-minor output rows leave through buffer stores: the resource's base is this tile's first row (wave-uniform,
    // SALU arithmetic once per tile), the frame's row offset travels in the instruction's SCALAR offset and lane * 4 is
    // the only VGPR -- no per-frame 64-bit vector address arithmetic in a VALU-bound loop (it was one v_lshl_add_u64
    // per frame).  Rows of a tile span at most SUB * V * 4 bytes < 2^32 (V < 2^24 voices per bank).
    __amdgpu_buffer_rsrc_t orow[OL == 0 ? NO : 1];
    if constexpr (OL == 0) {
#pragma unroll
        for (int c = 0; c < NO; c++)
            orow[c] = __builtin_amdgcn_make_buffer_rsrc(outw + ((size_t)c * T + (t0 + lo)) * V, 0, (int)0xffffffffu, 0x00020000);
    }
    const int vrow = (int)(V * sizeof(float));
    auto put = [&](int c, int i, float x) {  // i = frame index inside the block
        if constexpr (OL == 0) {
            __builtin_amdgcn_raw_buffer_store_b32(f2u(x), orow[c], lane * 4, (i - lo) * vrow, FD_PIPE_STORE_AUX);
        } else if constexpr (OL == 1) outw[(c * SUB + (i - lo)) * FS + lane] = x;
        else {
            float* cell = mx->tile + ((i - lo) & (MC - 1)) * MIX_ROW + mx->col;
            cell[c * MC * MIX_ROW] = x;  // (MIX_PAN: NO == 1, the mono sample; the weights come in when the chunk is flushed)
        }
    };
    auto flush_at = [&](int iend) {  // mix-down: the chunk that ends with frame iend - 1 of the block leaves for HBM
        if constexpr (MIXO) {
            const int nf = ((iend - lo - 1) & (MC - 1)) + 1;
            if constexpr (OL >= 3) mix_flush_pan<MC, OL == 3, MROLL>(mx->tile, outw + (t0 + iend - nf), T, nf, lane, mx->w, mx->wlds);
            else mix_flush<NM, MC, MROLL>(mx->tile, outw + (t0 + iend - nf), T, nf, lane);
        }
    };
    if (h == 0) SG::begin(g, size);
    if (lo < shi) {
        const G snap = g;  // tile-start registers, for the rollback below
        // (rolling this loop for heavy stages -- 1 pair per trip instead of 4 -- measured no faster on config 4: 52.9 vs 51.5 ms)
        // PF: the hand-over pairs of an item are read ONE ITEM AHEAD -- each pair's registers are re-armed right after the pair is
        // consumed, ~150-1000 cycles before the next item wants them.  Read where they are used (the compiler hoists them to
        // the item's top, no further) a consumer wave that has its SIMD to itself sits out the LDS round trip at the start of
        // every item (~140 cycles, profiles/r03_ubench_issue_v3.txt "ds_read2st64_b64 + wait"): stage 1 alone 3.55 -> 3.38 ms.
        // With a producer wave on the same SIMD the wait is filled anyway and the longer live ranges cost more than they save
        // (config 3, 65 536 voices: 4.72 vs 4.60 ms), so the four-group pipeline kernel leaves it off.
        v2f ahead[(!FIRST && NI > 0) ? NI : 1][4];
        if constexpr (!FIRST && PF) {
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) ahead[c][k] = hin[c][k][lane];
        }
        for (int i8 = lo; i8 < shi; i8 += 8) {  // one 8-sample SIMD item per trip (lo, shi are multiples of 8) ...
        item_begin(g);
        const int nx = i8 + 8 < shi ? i8 + 8 : i8;  // the tile's last item re-reads itself (never used)
#pragma unroll
        for (int i = i8; i < i8 + 8; i += 2) {  // ... two frames per inner iteration
            v2f pi[NI > 0 ? NI : 1], gi[NG > 0 ? NG : 1], po[NO];
            if constexpr (FIRST) {
#pragma unroll
                for (int c = 0; c < NI; c++) pi[c] = v2f{fin[(c * SUB + (i - lo)) * FS + lane], fin[(c * SUB + (i - lo + 1)) * FS + lane]};
            } else {
                if constexpr (PF) {  // consume the pair read during the previous item; re-arm its registers at once
#pragma unroll
                    for (int c = 0; c < NI; c++) {
                        pi[c] = ahead[c][(i - i8) >> 1];
                        ahead[c][(i - i8) >> 1] = hin[c][((nx - lo) >> 1) + ((i - i8) >> 1)][lane];
                    }
                } else
                {
#pragma unroll
                for (int c = 0; c < NI; c++) pi[c] = hin[c][(i - lo) >> 1][lane];
                }
            }
            if constexpr (GIN) {
#pragma unroll
                for (int c = 0; c < NG; c++) gi[c] = v2f{fin[(c * SUB + (i - lo)) * FS + lane], fin[(c * SUB + (i - lo + 1)) * FS + lane]};
            }
            SG::template step2<PH_SIMD>(g, pi, FIRST ? pi : gi, po);
            if constexpr (LAST) {
#pragma unroll
                for (int c = 0; c < NO; c++) {
                    put(c, i, po[c].x);
                    put(c, i + 1, po[c].y);
                }
            } else {
#pragma unroll
                for (int c = 0; c < NO; c++) hout[c][(i - lo) >> 1][lane] = po[c];
            }
        }
        if constexpr (MIXO) { if (((i8 + 8 - lo) & (MC - 1)) == 0) flush_at(i8 + 8); }
        }
        // (mix-down: the tile holds one chunk, so when ANY lane tripped every lane redoes the tile frame by frame -- step() and
        // step2() agree bit for bit where step2 is exact -- and the chunks already flushed are flushed again)
        bool trip = SG::tripped(g);
        // (... lanes past the end of the bank do not vote: their default-constructed voices land in the padding column whatever they compute)
        if constexpr (MIXO) trip = __builtin_amdgcn_ballot_w64(trip && mx->col < 64) != 0ull;
        if (__builtin_expect(trip, 0)) {  // a packed-path shortcut left its exact domain: redo the tile
            g = snap;
            for (int i = lo; i < shi; i++) {
                float fi[NI > 0 ? NI : 1], gf[NG > 0 ? NG : 1], fo[NO];
                if constexpr (FIRST) {
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = fin[(c * SUB + (i - lo)) * FS + lane];
                } else {
#pragma unroll
                    for (int c = 0; c < NI; c++) fi[c] = reinterpret_cast<const float*>(&hin[c][(i - lo) >> 1][lane])[i & 1];
                }
                if constexpr (GIN) {
#pragma unroll
                    for (int c = 0; c < NG; c++) gf[c] = fin[(c * SUB + (i - lo)) * FS + lane];
                }
                SG::template step<PH_SIMD>(g, fi, FIRST ? fi : gf, fo);
                if constexpr (LAST) {
#pragma unroll
                    for (int c = 0; c < NO; c++) put(c, i, fo[c]);
                    if constexpr (MIXO) { if (((i + 1 - lo) & (MC - 1)) == 0) flush_at(i + 1); }
                } else {
#pragma unroll
                    for (int c = 0; c < NO; c++) reinterpret_cast<float*>(&hout[c][(i - lo) >> 1][lane])[i & 1] = fo[c];
                }
            }
        }
    }
    // end_simd runs once per block, after its last packed item and before its remainder (also when full == 0)
    if (MODE == MODE_PROCESS && h == (full == 0 ? 0 : (full - 1) / SUB)) SG::end(g);
    for (int i = lo > full ? lo : full; i < hi; i++) {
        float fi[NI > 0 ? NI : 1], gf[NG > 0 ? NG : 1], fo[NO];
        if constexpr (FIRST) {
#pragma unroll
            for (int c = 0; c < NI; c++) fi[c] = fin[(c * SUB + (i - lo)) * FS + lane];
        } else {
#pragma unroll
            for (int c = 0; c < NI; c++) fi[c] = reinterpret_cast<const float*>(&hin[c][(i - lo) >> 1][lane])[i & 1];
        }
        if constexpr (GIN) {
#pragma unroll
            for (int c = 0; c < NG; c++) gf[c] = fin[(c * SUB + (i - lo)) * FS + lane];
        }
        SG::template step<(MODE == MODE_PROCESS ? PH_REM : PH_TICK)>(g, fi, FIRST ? fi : gf, fo);
        if constexpr (LAST) {
#pragma unroll
            for (int c = 0; c < NO; c++) put(c, i, fo[c]);
            if constexpr (MIXO) { if (((i + 1 - lo) & (MC - 1)) == 0) flush_at(i + 1); }
        } else {
#pragma unroll
            for (int c = 0; c < NO; c++) reinterpret_cast<float*>(&hout[c][(i - lo) >> 1][lane])[i & 1] = fo[c];
        }
    }
    if constexpr (MIXO) { if (hi > lo && ((hi - lo) & (MC - 1)) != 0) flush_at(hi); }  // the tile's last, short chunk
}

struct RoleOrder { int role[4]; };
template <bool FEED, int S, int W0, int W1, int W2, int GPW>
constexpr RoleOrder role_order() {
    RoleOrder o{{0, 1, 2, 3}};
    constexpr int NW = S + (FEED ? 1 : 0);
    if (GPW != 2 || NW < 3) return o;
    int wt[4] = {0, 0, 0, 0};
    int n = 0;
    if (FEED) wt[n++] = 4;
    wt[n++] = W0;
    if (S >= 2) wt[n++] = W1;
    if (S >= 3) wt[n++] = W2;
    if (NW == 3) {  // slots 0 and 2 share a SIMD, slot 1 has one to itself: the heaviest role goes there
        int h = 0;
        for (int i = 1; i < 3; i++) if (wt[i] > wt[h]) h = i;
        if (h != 1) { o.role[1] = h; o.role[h] = 1; }
        return o;
    }
    // four roles: the pairing {a, b | c, d} with the lightest heavier pair; slots (0, 2) take the first pair
    const int pairs[3][4] = {{0, 2, 1, 3}, {0, 1, 2, 3}, {0, 3, 1, 2}};  // identity first: ties keep the plain order
    int best = 0, best_w = 1 << 30;
    for (int p = 0; p < 3; p++) {
        int a = wt[pairs[p][0]] + wt[pairs[p][1]], b = wt[pairs[p][2]] + wt[pairs[p][3]], m = a > b ? a : b;
        if (m < best_w) { best_w = m; best = p; }
    }
    o.role[0] = pairs[best][0]; o.role[2] = pairs[best][1]; o.role[1] = pairs[best][2]; o.role[3] = pairs[best][3];
    return o;
}

constexpr bool role_order_is(RoleOrder o, int a, int b, int c, int d) { return o.role[0] == a && o.role[1] == b && o.role[2] == c && o.role[3] == d; }
static_assert(role_order_is(role_order<true, 3, 100, 130, 83, 2>(), 0, 1, 2, 3), "config 4, exact: loader + moog | saw + tail");
static_assert(role_order_is(role_order<true, 3, 100, 30, 83, 2>(), 0, 2, 1, 3), "config 4, tolerance mode: loader + saw | moog + tail");
static_assert(role_order_is(role_order<false, 3, 30, 10, 20, 2>(), 1, 0, 2, 3), "three roles: the heaviest gets the SIMD of its own");
static_assert(role_order_is(role_order<true, 3, 100, 30, 83, 4>(), 0, 1, 2, 3), "four groups per workgroup: every SIMD holds one group's roles");

// The pipeline kernel.  Waves of one workgroup, 4 voice groups (w & 3) times NW roles (w >> 2):
//   role 0 (only if the graph has inputs): the LOADER wave.  It does nothing but stream the group's input channels
//     from HBM into the feed ring, one tile ahead.  gfx9-family waves have ONE counter for loads and
//     stores (vmcnt) and mixed pending loads/stores force s_waitcnt vmcnt(0), so a wave that also stores samples
//     stalls on its own stores whenever it waits for an input; the loader never stores, the compute waves never load.
//   then S compute stages (S == 1: the whole graph), stage s one tile behind stage s - 1.
// The hardware places waves w, w+4, w+8, ... of a workgroup on the same SIMD.
// MIX != MIX_NONE (k_render_pipe_mix): `out` is the bank's partial-mix buffer [voice group][mix channel][T] and `panw` the pan
// weights [2][stride] (MIX_PAN); every lane of a live voice group runs (the padded voices of the last group are constructed
// voices with default parameters; their samples land in the tile's padding column).
template <class G, int MODE, int S, int K1, int K2, int GPW = 4, int MIX = MIX_NONE>
FD_D void render_pipe_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                           float* __restrict__ out, size_t T, const void* aux, float* ring, uint32_t ring_cap, const float* __restrict__ panw = nullptr) {
    using TL = PipeTiles<G, S, K1, K2, GPW>;
    constexpr int NI = G::IN;
    constexpr bool FEED = NI > 0;
    constexpr int SUB = TL::SUB, SPB = 64 / SUB, W = TL::W, D = FEED ? TL::D : 1;
    static_assert(TL::ok, "tiles do not fit");
    using S0 = typename TL::S0;
    using S1 = typename TL::S1;
    using S2 = typename TL::S2;
    // GPW = voice groups per workgroup: 4 fills a CU's SIMDs with one workgroup; small banks use 2 or 1 so that every CU
    // gets a workgroup (a heavy graph's waves are latency-bound: spreading them over idle CUs beats stacking them)
    __shared__ float feed[FEED ? GPW : 1][D][FEED ? NI : 1][FEED ? SUB : 1][FEED ? 64 : 1];  // [group][ring slot][channel][frame][lane]
    __shared__ v2f hand[S > 1 ? S - 1 : 1][S > 1 ? GPW : 1][2][S > 1 ? W : 1][S > 1 ? SUB / 2 : 1][S > 1 ? 64 : 1];  // [cut][group][buffer][channel][frame pair][lane]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // w: wave-uniform
    // Two voice groups per workgroup: waves w and w + 4 share a SIMD, i.e. role slots (0, 2) and (1, 3) of a group with
    // four roles, slots 0 and 2 of one with three.  The roles are dealt to the slots so that the heaviest SIMD is as
    // light as possible (config 4, exact: loader + moog | saw + tail; tolerance mode: loader + saw | moog + tail).
    constexpr RoleOrder RO = role_order<FEED, S, S0::weight, S1::weight, S2::weight, GPW>();
    const int grp = w % GPW;
    const int role = RO.role[w / GPW];  // 0 = loader (graphs with inputs), then the compute stages
    // (Half-filled waves -- 32 voices per wave, twice the waves -- were measured for the heavy config-4 voice: 33.7 ms
    // against 31.0 ms.  Its SIMDs are issue-bound on expensive instructions, not latency-bound; profiles/r02_c4_vpw.txt.)
    const size_t v0 = ((size_t)blockIdx.x * GPW + grp) * 64;
    const size_t v = v0 + lane;
    const bool live = v0 < stride;  // a whole group beyond the bank still takes part in the barriers
    const bool active = v < V;
    const size_t ntiles = ((T + 63) / 64) * SPB;
    const size_t rounds = ntiles + (S - 1) + (FEED ? 1 : 0);
    const float* inw = in + v0;  // wave-uniform bases + lane
    constexpr int NM = MIX == MIX_PAN ? 2 : G::OUT;  // mix channels
    constexpr int NTC = MIX == MIX_PAN ? 1 : G::OUT;  // channels of the LDS tile (MIX_PAN parks the mono samples)
    using MG = MixGeom<NTC, GPW, SUB>;
    // more than two waves per SIMD (workgroups of more than 8 waves) leave a wave 128 registers or fewer: the pan weights then live in LDS
    // and the flush's passes stay a loop (as in the 14-wave time-split kernel)
    constexpr bool TIGHT = GPW * ((FEED ? 1 : 0) + S) > 8;
    constexpr bool WLDS = MIX == MIX_PAN && TIGHT;
    constexpr int OLM = MIX == MIX_NONE ? 0 : (MIX == MIX_PAN ? (WLDS ? 4 : 3) : 2), MCM = MIX == MIX_NONE ? 0 : MG::MC;
    static_assert(MIX == MIX_NONE || MG::ok, "mix tile does not fit");
    float* outw = MIX == MIX_NONE ? out + v0 : out + (v0 / 64) * (size_t)NM * T;  // voice-out rows | this group's partial mix
    __shared__ __attribute__((aligned(16))) float mixt[MIX == MIX_NONE ? 1 : GPW][MIX == MIX_NONE ? 4 : MG::FLOATS];
    __shared__ __attribute__((aligned(16))) v2f mixw[WLDS ? GPW : 1][WLDS ? 64 : 1];
    MixLane mxl{&mixt[MIX == MIX_NONE ? 0 : grp][0], active ? lane : 64, {}, nullptr};
    const bool run = live && (MIX != MIX_NONE || active);  // (wave-uniform in a mix-down launch)

    if (FEED && role == 0) {  // ---- loader wave ----
        float rg[FEED ? NI : 1][SUB];
        auto issue = [&](size_t j) {  // frames past the end re-read the last frame (never used): no branches
            const size_t tj = (j / SPB) * 64 + (j % SPB) * SUB;
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int k = 0; k < SUB; k++) {
                    const size_t t = tj + k < T ? tj + k : T - 1;
                    if (MIX != MIX_NONE && !active) rg[c][k] = 0.0f;  // a padded voice of a mix-down launch hears silence
                    else rg[c][k] = __builtin_nontemporal_load(&inw[((size_t)c * T + t) * V + lane]);
                }
        };
        const bool on = run;
        if (on) issue(0);
        for (size_t it = 0; it < rounds; it++) {
            if (on && it < ntiles) {
#pragma unroll
                for (int c = 0; c < NI; c++)
#pragma unroll
                    for (int k = 0; k < SUB; k++) feed[grp][it % D][c][k][lane] = rg[c][k];
                if (it + 1 < ntiles) issue(it + 1);
            }
            __syncthreads();
        }
        return;
    }

    const int stage = role - (FEED ? 1 : 0);
    const size_t first = (size_t)stage + (FEED ? 1 : 0);  // the round in which this stage sees tile 0
    G g{};
    Ctx ctx{static_cast<const Aux*>(aux), ring + (live ? v : 0), ring_cap, stride, 0};
    g.bind(ctx);
    if (live) {
        VLoad ld{slots + v, stride, 0};
        VGate::W<VLoad> gate{&ld, true};
        if (stage == 0) S0::visit(g, gate); else if (stage == 1) S1::visit(g, gate); else S2::visit(g, gate);
    }
    if constexpr (MIX != MIX_NONE) {
        if (stage == S - 1) {  // the wave that owns the group's mix tile: silence in the columns no lane will write
            if (!active) {
#pragma unroll 1
                for (int r = 0; r < NTC * MCM; r++) mxl.tile[r * MIX_ROW + lane] = 0.0f;
            }
            if constexpr (MIX == MIX_PAN) {  // the weights of the quarter this lane adds up (panw is padded to `stride` with zeros)
                if constexpr (WLDS) {
                    if (live) mixw[grp][lane] = v2f{panw[v0 + lane], panw[stride + v0 + lane]};
                    mxl.wlds = &mixw[grp][0];
                } else if (live) {
#pragma unroll
                    for (int j = 0; j < 16; j++) mxl.w[j] = v2f{panw[v0 + (lane & 3) * 16 + j], panw[stride + v0 + (lane & 3) * 16 + j]};
                }
            }
        }
    }
    // The rounds, for the graph type GG: G itself, or its lowpass-specialised twin LpOf<G> (same registers, the
    // packed SVF path 5 operations shorter) when every lane of THIS wave qualifies.  A wave that does not hold the SVF
    // segment sees zeroed coefficients, takes the generic type and runs the same arithmetic for its own segment.
    constexpr bool PFK = FD_PIPE_PREFETCH != 0 && GPW < 4;  // see pipe_stage: prefetch pays when the consumer wave is (nearly) alone on its SIMD
    // the stages' shapes: the LAST one owns the output (HBM rows, or the mix tile of a fused mix-down)
    using C0 = StageCfg<SUB, W, true, S == 1, 64, S == 1 ? OLM : 0, (FD_PIPE_PREFETCH != 0), S == 1 ? MCM : 0, S == 1 && TIGHT>;
    using C1 = StageCfg<SUB, W, false, S == 2, 64, S == 2 ? OLM : 0, PFK, S == 2 ? MCM : 0, S == 2 && TIGHT>;
    using C2 = StageCfg<SUB, W, false, true, 64, OLM, PFK, MCM, TIGHT>;
    auto rounds_of = [&](auto* tag) {
        using GG = typename Pointee<decltype(tag)>::type;
        using TG = PipeTiles<GG, S, K1, K2>;
        using T0 = typename TG::S0;
        using T1 = typename TG::S1;
        using T2 = typename TG::S2;
        GG& gg = reinterpret_cast<GG&>(g);
        auto tile0 = [&](size_t j, size_t t0, int h, int size, int full, const float* fin) {
            if constexpr (S == 1) pipe_stage<T0, GG, MODE, C0>(gg, h, t0, size, full, T, V, lane, outw, fin, nullptr, nullptr, &mxl);
            else pipe_stage<T0, GG, MODE, C0>(gg, h, t0, size, full, T, V, lane, outw, fin, nullptr, hand[0][grp][j & 1]);
        };
        auto tile1 = [&](size_t j, size_t t0, int h, int size, int full, const float* fin) {
            if constexpr (S == 2) pipe_stage<T1, GG, MODE, C1>(gg, h, t0, size, full, T, V, lane, outw, fin, hand[0][grp][j & 1], nullptr, &mxl);
            else if constexpr (S == 3) pipe_stage<T1, GG, MODE, C1>(gg, h, t0, size, full, T, V, lane, outw, fin, hand[0][grp][j & 1], hand[S - 2][grp][j & 1]);
        };
        auto tile2 = [&](size_t j, size_t t0, int h, int size, int full, const float* fin) {
            if constexpr (S == 3) pipe_stage<T2, GG, MODE, C2>(gg, h, t0, size, full, T, V, lane, outw, fin, hand[S - 2][grp][j & 1], nullptr, &mxl);
        };
        if constexpr (S + (FEED ? 1 : 0) >= 3) {
            // One round loop PER ROLE for kernels of three or more roles (the stage is wave-uniform and fixed for the launch): with the
            // stage dispatch inside a common loop the register allocator sees the live ranges of all roles at once -- the config-4 kernel
            // spilled ~40 SGPRs to VGPR lanes in every round's preamble -- and every round pays the dispatch branches: config 4
            // 9.45 -> 9.0 ms.  Two-role kernels keep the common loop (config 3: 4.63 vs 4.74 ms with per-role loops,
            // profiles/r03_ab17_stage_loops.txt).  Every wave executes `rounds` barriers either way.
            auto loop = [&](auto&& tile) {
                for (size_t it = 0; it < rounds; it++) {
                    if (run && it >= first && it - first < ntiles) {
                        const size_t j = it - first;          // the tile this stage works on in this round
                        const size_t t0 = (j / SPB) * 64;
                        const int h = (int)(j % SPB);
                        const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
                        const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
                        const float* fin = nullptr;
                        if constexpr (FEED) fin = &feed[grp][j % D][0][0][0];
                        tile(j, t0, h, size, full, fin);
                    }
                    __syncthreads();  // hand-over point: every role has finished its tile of this round
                }
            };
            if (stage == 0) loop(tile0); else if (stage == 1) loop(tile1); else loop(tile2);
        } else {
            for (size_t it = 0; it < rounds; it++) {
                if (run && it >= first && it - first < ntiles) {
                    const size_t j = it - first;          // the tile this stage works on in this round
                    const size_t t0 = (j / SPB) * 64;
                    const int h = (int)(j % SPB);
                    const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
                    const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
                    const float* fin = nullptr;
                    if constexpr (FEED) fin = &feed[grp][j % D][0][0][0];
                    if (stage == 0) tile0(j, t0, h, size, full, fin);
                    else if (stage == 1) tile1(j, t0, h, size, full, fin);
                    else tile2(j, t0, h, size, full, fin);
                }
                __syncthreads();  // hand-over point: every role has finished its tile of this round
            }
        }
    };
    using GL = typename LpOf<G>::type;
    bool lp = false;
    if constexpr (!SameType<GL, G>::v && MODE == MODE_PROCESS && FD_LP_ENABLE)
        lp = __builtin_amdgcn_ballot_w64(run && !lp_ok(g)) == 0ull && __builtin_amdgcn_ballot_w64(run) != 0ull;
    // The HEAVIEST stage's wave is the critical path of a voice group: its instruction stream is one dependent chain, so every
    // cycle it waits for the VALU behind a sibling's instruction is a cycle added to the round.  VALU arbitration on a SIMD
    // is oldest-first (profiles/r03_ubench_issue_v2.txt: the older of two waves runs unimpeded, the younger gets the
    // leftovers), and wave age follows the role order, not the weight -- so the heaviest stage asks for priority
    // (s_setprio 1: config 3 exact 5.20 -> 4.95 ms before the other changes of the round; the lighter waves fill its gaps).
    {
        constexpr int w0 = S0::weight, w1 = S >= 2 ? S1::weight : 0, w2 = S >= 3 ? S2::weight : 0;
        constexpr int heavy = (w0 >= w1 && w0 >= w2) ? 0 : (w1 >= w2 ? 1 : 2);
        if (S > 1 && stage == heavy) __builtin_amdgcn_s_setprio(1);
    }
    if (lp) rounds_of((GL*)nullptr); else rounds_of((G*)nullptr);
    if (live && active) {
        VStore<false> st{slots + v, stride, 0};
        VGate::W<VStore<false>> gate{&st, true};
        if (stage == 0) S0::visit(g, gate); else if (stage == 1) S1::visit(g, gate); else S2::visit(g, gate);
    }
}

// the pipeline kernel with the fused mix-down: part = [voice groups][mix channels][T] partial mixes, panw = [2][stride] (MIX_PAN)
template <class G, int MODE, int S, int K1, int K2, int GPW, int MIX>
__global__ __launch_bounds__((16 * GPW * PipeGeom<G::IN, S>::WAVES)) FD_PIPE_ATTR void k_render_pipe_mix(float* __restrict__ slots, size_t stride, size_t V,
                                                                                          const float* __restrict__ in, float* __restrict__ part,
                                                                                          size_t T, const void* aux, float* ring, uint32_t ring_cap,
                                                                                          const float* __restrict__ panw) {
    static_assert(MIX == MIX_SUM || (MIX == MIX_PAN && G::OUT == 1), "mix-down: sum of the outputs, or pan of a mono graph");
    render_pipe_body<G, MODE, S, K1, K2, GPW, MIX>(slots, stride, V, in, part, T, aux, ring, ring_cap, panw);
}

template <class G, int MODE, int S, int K1, int K2, int GPW>
__global__ __launch_bounds__((16 * GPW * PipeGeom<G::IN, S>::WAVES)) FD_PIPE_ATTR void k_render_pipe(float* __restrict__ slots, size_t stride, size_t V,
                                                                                      const float* __restrict__ in, float* __restrict__ out,
                                                                                      size_t T, const void* aux, float* ring, uint32_t ring_cap) {
    render_pipe_body<G, MODE, S, K1, K2, GPW>(slots, stride, V, in, out, T, aux, ring, ring_cap);
}

// ---- time-split pipeline for banks that leave most of the chip idle (strong-scaling shards) ------------------------
// The pipeline above gives a voice group as many waves as its chain has stages, and every wave walks all 64 frames of a
// block: the time per frame is the longest stage's instruction count, whatever the number of idle SIMDs around it.  A
// bank of <= 2 voice groups per CU (32 768 voices on an MI355X: the 2-, 4-, 8-GPU shards of the 65 536-voice metric) has
// SIMDs to spare, so here the FEED-FORWARD work of a stage is also split over TIME: a stage whose state advance is cheap
// next to its output (an oscillator: one or two operations of phase recurrence against a ~25-operation sine polynomial)
// runs in several waves; each wave advances the state through all 64 frames of the block (skip2) but evaluates the
// output only for its own share of the frames.  Stages 0 and 1 of a three-stage chain are split like that (NA and NB
// waves), the last stage (the serial filter) is one wave running pipe_stage as before.  Same arithmetic per frame as
// every other kernel: bit-exact.
//   waves of a workgroup (ONE voice group): NA x stage 0 | NB x stage 1 | 1 x stage 2, stage s one block behind stage s-1;
//   hand-over tiles [frame pair][lane] v2f, double-buffered: 2 cuts x 2 x 16 KiB = 64 KiB -> two workgroups per CU.
// Needs: process mode, voice-minor layout, no graph inputs, a 3-stage chain whose first two stages define skip2,
// T a multiple of 64 (launch_render falls back to the pipeline kernel otherwise).
template <class SG, class G, bool FIRST, int W>
FD_D void ts_stage(G& g, int lo, int hi, int lane, v2f (*hin)[32][64], v2f (*hout)[32][64]) {
    constexpr int NI = SG::IN, NO = SG::OUT;
    static_assert(NO <= W && (FIRST || NI <= W), "hand-over tile too narrow");
    // [lo, hi): this wave's frames of the block (multiples of 8; ts_part / Ts3Roles::cut)
    SG::begin(g, 64);
    const G snap = g;
    // The block in 8-frame items; a consumer stage reads every item's four hand-over pairs ONE ITEM AHEAD (each pair's
    // registers re-armed as soon as the pair is consumed).  Read pair by pair at the point of use -- as rounds 1-2 had it --
    // every pair of a SKIPPED item cost a whole LDS round trip (~140 cycles for three VALU instructions): the carrier
    // waves, which skip through two thirds of the block, were the slowest waves of the kernel for that reason alone.
    v2f ahead[(!FIRST && NI > 0) ? NI : 1][4];
    if constexpr (!FIRST) {
#pragma unroll
        for (int c = 0; c < NI; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) ahead[c][k] = hin[c][k][lane];
    }
    for (int i8 = 0; i8 < 64; i8 += 8) {
        const int nx = i8 + 8 < 64 ? i8 + 8 : i8;  // the last item re-reads itself (never used)
        const bool mine = i8 >= lo && i8 < hi;     // wave-uniform
        if (mine) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v2f pi[NI > 0 ? NI : 1], po[NO];
                if constexpr (!FIRST) {
#pragma unroll
                    for (int c = 0; c < NI; c++) {
                        pi[c] = ahead[c][k];
                        ahead[c][k] = hin[c][(nx >> 1) + k][lane];
                    }
                }
                SG::template step2<PH_SIMD>(g, pi, pi, po);
#pragma unroll
                for (int c = 0; c < NO; c++) hout[c][(i8 >> 1) + k][lane] = po[c];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v2f pi[NI > 0 ? NI : 1];
                if constexpr (!FIRST) {
#pragma unroll
                    for (int c = 0; c < NI; c++) {
                        pi[c] = ahead[c][k];
                        ahead[c][k] = hin[c][(nx >> 1) + k][lane];
                    }
                }
                SG::template skip2<PH_SIMD>(g, pi);
            }
        }
    }
    if (__builtin_expect(SG::tripped(g), 0)) {  // a packed-path shortcut left its exact domain: redo the block, frame by frame
        g = snap;
        for (int i = 0; i < 64; i++) {
            float fi[NI > 0 ? NI : 1], fo[NO];
            if constexpr (!FIRST) {
#pragma unroll
                for (int c = 0; c < NI; c++) fi[c] = reinterpret_cast<const float*>(&hin[c][i >> 1][lane])[i & 1];
            }
            if (i >= lo && i < hi) {
                SG::template step<PH_SIMD>(g, fi, fi, fo);
#pragma unroll
                for (int c = 0; c < NO; c++) reinterpret_cast<float*>(&hout[c][i >> 1][lane])[i & 1] = fo[c];
            } else {
                SG::template skip<PH_SIMD>(g, fi);
            }
        }
    }
    SG::end(g);  // end_simd: once per block, after its last packed item
}

template <class G, bool SHAPE = (Chain<G>::N == 3 && G::IN == 0 && G::RINGS == 0)>
struct TsPlan {  // which graphs the time-split kernel takes
    static constexpr bool ok = false;
};
// ... a three-stage chain without inputs or rings whose first two stages can advance their state without producing
// output (skip2 on every node of the stage: Constant, Sine, Pipe, Unop today).  A 3-stage kind headed by a Noise, a
// WaveSynth or a Binop split simply does not qualify (it used to fail to COMPILE).
template <class G> struct TsPlan<G, true> {
    static constexpr bool ok = Seg<G, 0, 1>::HAS_SKIP && Seg<G, 1, 2>::HAS_SKIP;
};
static_assert(HasSkip<Pipe<Constant<1>, Sine>>::v && HasSkip<Unop<Pipe<Constant<1>, SineFast>, UMulScalar>>::v && HasSkip<Noise>::v &&
              !HasSkip<FixedSvf>::v && !HasSkip<Biquad>::v, "HasSkip: oscillator chains and the counter-based Noise yes, filters no");
static_assert(TsPlan<Pipe<Pipe<Unop<Pipe<Constant<1>, Sine>, UAddScalar>, Sine>, FixedSvf>>::ok && !TsPlan<Pipe<Pipe<Pass, Sine>, FixedSvf>>::ok &&
              !TsPlan<Pipe<Sine, FixedSvf>>::ok, "TsPlan: three-stage generator chains only");
static_assert(Chain<Pipe<Noise, Biquad>>::N == 3 && TsPlan<Pipe<Noise, Biquad>>::ok && Seg<Pipe<Noise, Biquad>, 1, 2>::HAS_SKIP && !Seg<Pipe<Noise, Biquad>, 2, 3>::HAS_SKIP,
              "config 2: noise | the biquad's feed-forward half | its recurrence");

static __device__ unsigned int g_ts_arrivals[4096];  // workgroups of k_render_ts<.., 2, 1> seen per CU (role draw; never reset)

template <class G, int NA, int NB>
FD_D void render_ts_body(float* __restrict__ slots, size_t stride, size_t V, float* __restrict__ out, size_t T,
                         const void* aux) {
    using S0 = Seg<G, 0, 1>;
    using S1 = Seg<G, 1, 2>;
    using S2 = Seg<G, 2, 3>;
    constexpr int W = S0::OUT > S1::OUT ? S0::OUT : S1::OUT;
    static_assert(W * 2 * 2 * 16 <= 64, "hand-over tiles must fit 64 KiB");
    __shared__ v2f hand[2][2][W][32][64];  // [cut][buffer][channel][frame pair][lane]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Role of wave w: NA waves of stage 0, NB of stage 1, one filter wave; rank r in [A0 .. A(NA-1), B0 .. B(NB-1), C].
    // Waves w and w + 4 of a workgroup share a SIMD, and two workgroups share a CU when the bank has more than one voice
    // group per CU, so the order of the roles decides how evenly the four SIMDs are loaded:
    //   <2, 2> (one group per CU, 5 waves; waves 0 and 4 share a SIMD): the filter -- the heavy wave, ~18 instructions
    //           per frame against ~11 of an oscillator half -- is not one of the two:       [A0, C, A1, B0, B1]
    //   <2, 1> (two groups per CU, 4 waves each): roles by SIMD id, see below.
    int r;
    if (NA == 2 && NB == 2) r = w == 1 ? 4 : (w > 1 ? w - 1 : w);
    else if (NA == 2 && NB == 1) {
        // Which SIMD wave w of a workgroup gets differs from workgroup to workgroup (the placement walks 0,2,1,3 from
        // wherever the CU's pointer stands), and the two workgroups of a CU are not neighbours in the grid
        // (profiles/r02_census_wave_placement.txt).  So the roles go by what the waves find: every wave reads its SIMD
        // id, the workgroup draws 0 or 1 from a per-CU arrival counter, and SIMD s takes role ts_role[draw][s] -- two
        // workgroups with different draws put an oscillator half next to a whole-block wave on every SIMD.  Any
        // permutation is CORRECT; if the four waves do not sit on four SIMDs the plain order is used.
        __shared__ int s_simd[4], s_draw;
        const uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        if (lane == 0) s_simd[w] = (int)((hw >> 4) & 3);
        if (threadIdx.x == 0) {
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf;  // HW_REG_XCC_ID
            const uint32_t key = (xcc << 8) | ((hw >> 8) & 0xff);                               // XCC, SE, SH, CU
            s_draw = (int)(atomicAdd(&g_ts_arrivals[key & 4095], 1u) & 1u);
        }
        __syncthreads();
        const int mask = (1 << s_simd[0]) | (1 << s_simd[1]) | (1 << s_simd[2]) | (1 << s_simd[3]);
        constexpr int ts_role[2][4] = {{0, 1, 2, 3}, {2, 3, 0, 1}};  // draw 0: A0 A1 B C on SIMD 0..3; draw 1: B C A0 A1
        r = mask == 15 ? ts_role[s_draw][(hw >> 4) & 3] : w;
    } else r = w;
    const int stage = r < NA ? 0 : (r < NA + NB ? 1 : 2);
    const int part = r < NA ? r : (r < NA + NB ? r - NA : 0);
    const int nparts = stage == 0 ? NA : NB;
    const size_t v0 = (size_t)blockIdx.x * 64, v = v0 + lane;
    const bool active = v < V;
    const size_t nblocks = T / 64, rounds = nblocks + 2;
    float* outw = out + v0;
    G g{};
    Ctx ctx{static_cast<const Aux*>(aux), nullptr, 0, stride, 0};
    g.bind(ctx);
    {
        VLoad ld{slots + v, stride, 0};
        VGate::W<VLoad> gate{&ld, true};
        if (stage == 0) S0::visit(g, gate); else if (stage == 1) S1::visit(g, gate); else S2::visit(g, gate);
    }
    auto rounds_of = [&](auto* tag) {
        using GG = typename Pointee<decltype(tag)>::type;
        using T0 = Seg<GG, 0, 1>;
        using T1 = Seg<GG, 1, 2>;
        using T2 = Seg<GG, 2, 3>;
        GG& gg = reinterpret_cast<GG&>(g);
        for (size_t it = 0; it < rounds; it++) {
            if (active && it >= (size_t)stage && it - stage < nblocks) {
                const size_t j = it - stage;  // the block this stage works on in this round
                if (stage == 0) ts_stage<T0, GG, true, W>(gg, 64 / nparts * part, 64 / nparts * (part + 1), lane, nullptr, hand[0][j & 1]);
                else if (stage == 1) ts_stage<T1, GG, false, W>(gg, 64 / nparts * part, 64 / nparts * (part + 1), lane, hand[0][j & 1], hand[1][j & 1]);
                else pipe_stage<T2, GG, MODE_PROCESS, StageCfg<64, W, false, true>>(gg, 0, j * 64, 64, 64, T, V, lane, outw, nullptr, hand[1][j & 1], nullptr);
            }
            __syncthreads();
        }
    };
    using GL = typename LpOf<G>::type;
    bool lp = false;
    if constexpr (!SameType<GL, G>::v && FD_LP_ENABLE)
        lp = __builtin_amdgcn_ballot_w64(active && !lp_ok(g)) == 0ull && __builtin_amdgcn_ballot_w64(active) != 0ull;
    if (stage == 2) __builtin_amdgcn_s_setprio(1);  // the serial filter wave is the round's critical path (see render_pipe_body)
    if (lp) rounds_of((GL*)nullptr); else rounds_of((G*)nullptr);
    if (active && part == 0) {  // the waves of a split stage end with identical state: one of them stores it
        VStore<false> st{slots + v, stride, 0};
        VGate::W<VStore<false>> gate{&st, true};
        if (stage == 0) S0::visit(g, gate); else if (stage == 1) S1::visit(g, gate); else S2::visit(g, gate);
    }
}

template <class G, int NA, int NB>
__global__ __launch_bounds__(64 * (NA + NB + 1)) void k_render_ts(float* __restrict__ slots, size_t stride, size_t V,
                                                                  float* __restrict__ out, size_t T, const void* aux) {
    if constexpr (TsPlan<G>::ok) render_ts_body<G, NA, NB>(slots, stride, V, out, T, aux);
}

// ---- time-split, three ways (round 3) ----------------------------------------------------------------------------------
// What the instruction-issue measurements of round 3 say about the kernel above (profiles/r03_ubench_issue_v*.txt): a
// wave issues one VALU instruction per 4.1-5.1 cycles whatever its neighbours do, and two waves that share a SIMD get in
// each other's way instruction by instruction.  In the 2 + 2 + 1 layout FIVE waves sit on four SIMDs, so one SIMD runs
// two oscillator halves back to back and sets the round time (2.27 ms per 48 000 frames where the filter wave alone
// needs ~1.2).  Here both oscillator stages are split THREE ways (24 + 24 + 16 frames of a block) and the roles are
// placed so that the serial filter wave has a SIMD to itself (one group per CU) or shares it with one oscillator pair:
//   GPW = 1 (<= one voice group per CU), 7 waves:  w: 0  1  2  3  4  5  6      (waves w and w + 4 share a SIMD)
//                                              role: A0 A1 A2 C  B0 B1 B2
//   GPW = 2 (<= two groups per CU), 14 waves = two such groups; the filter waves of the two groups land on the two SIMDs
//           that hold three waves (w % 4 = 2, 3), the four-wave SIMDs hold oscillator thirds only.
// The thirds of a stage each advance the state through the whole block (skip2) and evaluate their own frames; per frame
// the arithmetic is that of every other kernel -- bit-exact (tests/test_gpu_time_split.py).
template <int GPW> struct Ts3Roles;
// Seven waves left the four SIMDs of a CU loaded 48 | 48 | 32 oscillator frames per block | the filter alone, and the two 48s set the round time
// (profiles/r03_strong_scaling_shards.txt: the filter wave alone 1.21 ms, the kernel 1.98).  With the second oscillator stage in QUARTERS
// the eighth wave sits next to the filter:  w: 0  1  2  3  4  5  6  7   ->   40 | 40 | 32 | filter + 16.
//                                        role: A0 A1 A2 C  B0 B1 B2 B3
template <> struct Ts3Roles<1> {  // wave -> (group, stage, part); parts per stage
    static constexpr int WAVES = 8;
    static constexpr int grp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    static constexpr int stg[8] = {0, 0, 0, 2, 1, 1, 1, 1};
    static constexpr int prt[8] = {0, 1, 2, 0, 0, 1, 2, 3};
    static constexpr int cut[2][5] = {{0, 24, 48, 64, 64}, {0, 16, 32, 48, 64}};   // frames of a block per part: [cut[stage][part], cut[stage][part + 1])
};
template <> struct Ts3Roles<2> {
    // SIMD = w % 4:   SIMD 0: w 0 4 8 12   SIMD 1: w 1 5 9 13   SIMD 2: w 2 6 10   SIMD 3: w 3 7 11
    static constexpr int WAVES = 14;
    static constexpr int grp[14] = {0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 0};
    static constexpr int stg[14] = {0, 0, 2, 2, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1};
    static constexpr int prt[14] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 1, 1, 2, 2};
    static constexpr int cut[2][5] = {{0, 24, 48, 64, 64}, {0, 24, 48, 64, 64}};
    // SIMD 0: g0 A0, g0 B0, g0 A2, g1 B2   SIMD 1: g1 A0, g1 B0, g1 A2, g0 B2   SIMD 2: g0 C, g0 A1, g0 B1   SIMD 3: g1 C, g1 A1, g1 B1
};

template <class G, int GPW, int MIX = MIX_NONE>
FD_D void render_ts3_body(float* __restrict__ slots, size_t stride, size_t V, float* __restrict__ out, size_t T, const void* aux, const float* __restrict__ panw = nullptr) {
    using S0 = Seg<G, 0, 1>;
    using S1 = Seg<G, 1, 2>;
    using S2 = Seg<G, 2, 3>;
    using R = Ts3Roles<GPW>;
    constexpr int W = S0::OUT > S1::OUT ? S0::OUT : S1::OUT;
    static_assert(W * 2 * 2 * 16 * GPW <= 128, "hand-over tiles must fit 128 KiB");
    __shared__ v2f hand[GPW][2][2][W][32][64];  // [group][cut][buffer][channel][frame pair][lane]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = R::grp[w], stage = R::stg[w], part = R::prt[w];
    const size_t v0 = ((size_t)blockIdx.x * GPW + grp) * 64, v = v0 + lane;
    const bool live = v0 < stride, active = v < V;   // a whole group past the bank still takes part in the barriers
    const size_t nblocks = T / 64, rounds = nblocks + 2;
    // fused mix-down (see render_pipe_body): the filter wave parks its samples in the group's mix tile, `out` holds the partial mixes
    constexpr int NM = MIX == MIX_PAN ? 2 : G::OUT;
    constexpr int NTC = MIX == MIX_PAN ? 1 : G::OUT;
    using MG = MixGeom<NTC, GPW, 64>;
    constexpr bool WLDS = MIX == MIX_PAN && GPW == 2;  // 14 waves per workgroup: 128 registers per wave, the weights live in LDS
    constexpr int OLM = MIX == MIX_NONE ? 0 : (MIX == MIX_PAN ? (WLDS ? 4 : 3) : 2), MCM = MIX == MIX_NONE ? 0 : MG::MC;
    static_assert(MIX == MIX_NONE || MG::ok, "mix tile does not fit");
    float* outw = MIX == MIX_NONE ? out + v0 : out + (v0 / 64) * (size_t)NM * T;
    __shared__ __attribute__((aligned(16))) float mixt[MIX == MIX_NONE ? 1 : GPW][MIX == MIX_NONE ? 4 : MG::FLOATS];
    __shared__ __attribute__((aligned(16))) v2f mixw[WLDS ? GPW : 1][WLDS ? 64 : 1];
    MixLane mxl{&mixt[MIX == MIX_NONE ? 0 : grp][0], active ? lane : 64, {}, nullptr};
    const bool run = live && (MIX != MIX_NONE || active);
    G g{};
    Ctx ctx{static_cast<const Aux*>(aux), nullptr, 0, stride, 0};
    g.bind(ctx);
    if (live) {
        VLoad ld{slots + v, stride, 0};
        VGate::W<VLoad> gate{&ld, true};
        if (stage == 0) S0::visit(g, gate); else if (stage == 1) S1::visit(g, gate); else S2::visit(g, gate);
    }
    if constexpr (MIX != MIX_NONE) {
        if (stage == 2) {
            if (!active) {
#pragma unroll 1
                for (int r = 0; r < NTC * MCM; r++) mxl.tile[r * MIX_ROW + lane] = 0.0f;
            }
            if constexpr (MIX == MIX_PAN) {  // the weights of the quarter this lane adds up (panw is padded to `stride` with zeros)
                if constexpr (WLDS) {
                    if (live) mixw[grp][lane] = v2f{panw[v0 + lane], panw[stride + v0 + lane]};
                    mxl.wlds = &mixw[grp][0];
                } else if (live) {
#pragma unroll
                    for (int j = 0; j < 16; j++) mxl.w[j] = v2f{panw[v0 + (lane & 3) * 16 + j], panw[stride + v0 + (lane & 3) * 16 + j]};
                }
            }
        }
    }
    auto rounds_of = [&](auto* tag) {
        using GG = typename Pointee<decltype(tag)>::type;
        using T0 = Seg<GG, 0, 1>;
        using T1 = Seg<GG, 1, 2>;
        using T2 = Seg<GG, 2, 3>;
        GG& gg = reinterpret_cast<GG&>(g);
        auto loop = [&](auto&& block) {
            for (size_t it = 0; it < rounds; it++) {
                if (run && it >= (size_t)stage && it - stage < nblocks)
                    block(it - stage);  // the block this stage works on in this round
                __syncthreads();
            }
        };
        if (stage == 0) loop([&](size_t j) { ts_stage<T0, GG, true, W>(gg, R::cut[0][part], R::cut[0][part + 1], lane, nullptr, hand[grp][0][j & 1]); });
        else if (stage == 1) loop([&](size_t j) { ts_stage<T1, GG, false, W>(gg, R::cut[1][part], R::cut[1][part + 1], lane, hand[grp][0][j & 1], hand[grp][1][j & 1]); });
        else loop([&](size_t j) { pipe_stage<T2, GG, MODE_PROCESS, StageCfg<64, W, false, true, 64, OLM, FD_PIPE_PREFETCH != 0, MCM, (GPW == 2)>>(gg, 0, j * 64, 64, 64, T, V, lane, outw, nullptr, hand[grp][1][j & 1], nullptr, &mxl); });
    };
    using GL = typename LpOf<G>::type;
    bool lp = false;
    if constexpr (!SameType<GL, G>::v && FD_LP_ENABLE)
        lp = __builtin_amdgcn_ballot_w64(run && !lp_ok(g)) == 0ull && __builtin_amdgcn_ballot_w64(run) != 0ull;
    if (stage == 2) __builtin_amdgcn_s_setprio(1);  // the serial filter wave is the round's critical path
    if (lp) rounds_of((GL*)nullptr); else rounds_of((G*)nullptr);
    if (live && active && part == 0) {  // the waves of a split stage end with identical state: one of them stores it
        VStore<false> st{slots + v, stride, 0};
        VGate::W<VStore<false>> gate{&st, true};
        if (stage == 0) S0::visit(g, gate); else if (stage == 1) S1::visit(g, gate); else S2::visit(g, gate);
    }
}

template <class G, int GPW>
__global__ __launch_bounds__(64 * Ts3Roles<GPW>::WAVES) void k_render_ts3(float* __restrict__ slots, size_t stride, size_t V,
                                                                          float* __restrict__ out, size_t T, const void* aux) {
    if constexpr (TsPlan<G>::ok) render_ts3_body<G, GPW>(slots, stride, V, out, T, aux);
}
// ... with the fused mix-down (see k_render_pipe_mix)
template <class G, int GPW, int MIX>
__global__ __launch_bounds__(64 * Ts3Roles<GPW>::WAVES) void k_render_ts3_mix(float* __restrict__ slots, size_t stride, size_t V,
                                                                              float* __restrict__ part, size_t T, const void* aux,
                                                                              const float* __restrict__ panw) {
    if constexpr (TsPlan<G>::ok) render_ts3_body<G, GPW, MIX>(slots, stride, V, part, T, aux, panw);
}

// ---- the pipeline kernel for the PLANAR layout ([voice][channel][frame_stride], the reference's BufferArray rows) ----
// Same stages, same hand-over tiles; what changes is how samples enter and leave the workgroup:
//   * the LOADER wave reads each voice's row in float4 runs (16 lanes x 16 B = one 256-B run per voice row) and writes
//     them TRANSPOSED into the feed tile (rows padded to 65 floats: the 4-B transposed writes hit distinct banks);
//   * the last compute stage writes its samples into an LDS out tile instead of HBM;
//   * a STORER wave, one round behind, transposes that tile back and stores float4 runs into the voices' rows.
// The compute waves touch no global memory at all.  Needs 16-byte aligned rows (frame_stride % 4 == 0).
constexpr int PFS = 65;  // floats per frame row of the transposed tiles
template <class G>
struct PlanarPlan {
    static constexpr PipePlan P2 = pipe_plan<G>(2);  // the best two-stage cut, if the graph has one
    static constexpr int N = Chain<G>::N;
    template <int S, int K1> struct Tiles {
        using S0 = Seg<G, 0, S == 1 ? N : K1>;
        using S1 = Seg<G, S == 1 ? 0 : K1, N>;
        static constexpr int W = S >= 2 ? S0::OUT : 0;
        static constexpr bool LATE_GIN = S >= 2 && S1::USES_GIN;
        static constexpr int D = G::IN > 0 ? (LATE_GIN ? S + 1 : 2) : 0;
        static constexpr int UNITS = G::IN * D + 2 * W * (S - 1) + 2 * G::OUT;
        // one unit = SUB frames x 65 floats x 4 voice groups = SUB * 1040 B; gfx950 gives a workgroup up to 160 KB of LDS
        static constexpr int SUB = UNITS <= 4 ? 32 : UNITS <= 9 ? 16 : UNITS <= 18 ? 8 : 0;
        static constexpr int WAVES = S + 1 + (G::IN > 0 ? 1 : 0);  // per voice group
        static constexpr bool ok = SUB >= 8 && WAVES <= 4 && G::OUT > 0;
    };
    static constexpr bool two_ok() {
        if constexpr (P2.S == 2) return Tiles<2, P2.K1>::ok; else return false;
    }
    static constexpr bool two = two_ok();
    static constexpr int S = two ? 2 : (Tiles<1, N>::ok ? 1 : 0);
    static constexpr int K1 = two ? P2.K1 : N;
    using T = Tiles<(S >= 1 ? S : 1), K1>;
};

template <class G, int MODE, int S, int K1, int GPW = 4>
FD_D void render_pipe_planar_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                                  float* __restrict__ out, size_t T, size_t fstride, const void* aux, float* ring, uint32_t ring_cap) {
    using TL = typename PlanarPlan<G>::template Tiles<S, K1>;
    constexpr int NI = G::IN, NO = G::OUT;
    constexpr bool FEED = NI > 0;
    constexpr int SUB = TL::SUB, SPB = 64 / SUB, W = TL::W, D = FEED ? TL::D : 1;
    constexpr int LPR = SUB / 4;        // lanes per voice row (one float4 each)
    constexpr int VPP = 64 / LPR;       // voice rows per pass
    constexpr int PASSES = 64 / VPP;    // = SUB / 4
    static_assert(TL::ok, "tiles do not fit");
    using S0 = typename TL::S0;
    using S1 = typename TL::S1;
    __shared__ float feed[FEED ? GPW : 1][D][FEED ? NI : 1][FEED ? SUB : 1][FEED ? PFS : 1];
    __shared__ v2f hand[S > 1 ? GPW : 1][2][S > 1 ? W : 1][S > 1 ? SUB / 2 : 1][S > 1 ? 64 : 1];
    __shared__ float otile[GPW][2][NO][SUB][PFS];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = w % GPW, role = w / GPW;
    const size_t v0 = ((size_t)blockIdx.x * GPW + grp) * 64;
    const size_t v = v0 + lane;
    const bool live = v0 < stride;
    const bool active = v < V;
    const size_t ntiles = ((T + 63) / 64) * SPB;
    const size_t rounds = ntiles + (size_t)S + (FEED ? 1 : 0);  // ... + the storer's round
    const int lr = lane / LPR, fq = (lane % LPR) * 4;            // this lane's row inside a pass, its first frame in the tile
    auto tile_t = [&](size_t j) { return (j / SPB) * 64 + (j % SPB) * SUB; };

    if (FEED && role == 0) {  // ---- loader wave ----
        float4 rg[FEED ? NI : 1][PASSES];
        auto issue = [&](size_t j) {
            const size_t t = tile_t(j) + fq;
#pragma unroll
            for (int c = 0; c < NI; c++)
#pragma unroll
                for (int p = 0; p < PASSES; p++) {
                    const size_t gv = v0 + p * VPP + lr;
                    float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (gv < V && t < T) {
                        const float* src = in + (gv * NI + c) * fstride + t;
                        if (t + 3 < T) {
                            x = *reinterpret_cast<const float4*>(src);
                        } else {
                            x.x = src[0];
                            if (t + 1 < T) x.y = src[1];
                            if (t + 2 < T) x.z = src[2];
                        }
                    }
                    rg[c][p] = x;
                }
        };
        if (live) issue(0);
        for (size_t it = 0; it < rounds; it++) {
            if (live && it < ntiles) {
#pragma unroll
                for (int c = 0; c < NI; c++)
#pragma unroll
                    for (int p = 0; p < PASSES; p++) {
                        float* dst = &feed[grp][it % D][c][fq][p * VPP + lr];
                        dst[0] = rg[c][p].x;
                        dst[PFS] = rg[c][p].y;
                        dst[2 * PFS] = rg[c][p].z;
                        dst[3 * PFS] = rg[c][p].w;
                    }
                if (it + 1 < ntiles) issue(it + 1);
            }
            __syncthreads();
        }
        return;
    }

    const int stage = role - (FEED ? 1 : 0);  // S = the storer
    if (stage == S) {  // ---- storer wave: tile j leaves one round after the last compute stage wrote it ----
        const size_t first = (size_t)S + (FEED ? 1 : 0);
        for (size_t it = 0; it < rounds; it++) {
            if (live && it >= first && it - first < ntiles) {
                const size_t j = it - first;
                const size_t t = tile_t(j) + fq;
#pragma unroll
                for (int c = 0; c < NO; c++)
#pragma unroll
                    for (int p = 0; p < PASSES; p++) {
                        const size_t gv = v0 + p * VPP + lr;
                        const float* src = &otile[grp][j & 1][c][fq][p * VPP + lr];
                        const float4 x = make_float4(src[0], src[PFS], src[2 * PFS], src[3 * PFS]);
                        if (gv < V && t < T) {
                            float* dst = out + (gv * NO + c) * fstride + t;
                            if (t + 3 < T) {
                                *reinterpret_cast<float4*>(dst) = x;
                            } else {
                                dst[0] = x.x;
                                if (t + 1 < T) dst[1] = x.y;
                                if (t + 2 < T) dst[2] = x.z;
                            }
                        }
                    }
            }
            __syncthreads();
        }
        return;
    }

    const size_t first = (size_t)stage + (FEED ? 1 : 0);
    G g;
    Ctx ctx{static_cast<const Aux*>(aux), ring + (live ? v : 0), ring_cap, stride, 0};
    g.bind(ctx);
    if (live) {
        VLoad ld{slots + v, stride, 0};
        VGate::W<VLoad> gate{&ld, true};
        if (stage == 0) S0::visit(g, gate); else S1::visit(g, gate);
    }
    for (size_t it = 0; it < rounds; it++) {
        if (live && active && it >= first && it - first < ntiles) {
            const size_t j = it - first;
            const size_t t0 = (j / SPB) * 64;
            const int h = (int)(j % SPB);
            const int size = (int)((T - t0) < 64 ? (T - t0) : 64);
            const int full = MODE == MODE_PROCESS ? (size & ~7) : 0;
            const float* fin = nullptr;
            if constexpr (FEED) fin = &feed[grp][j % D][0][0][0];
            float* ot = &otile[grp][j & 1][0][0][0];
            if (stage == 0) {
                if constexpr (S == 1) pipe_stage<S0, G, MODE, StageCfg<SUB, W, true, true, PFS, 1>>(g, h, t0, size, full, T, V, lane, ot, fin, nullptr, nullptr);
                else pipe_stage<S0, G, MODE, StageCfg<SUB, W, true, false, PFS, 1>>(g, h, t0, size, full, T, V, lane, ot, fin, nullptr, hand[grp][j & 1]);
            } else {
                if constexpr (S == 2) pipe_stage<S1, G, MODE, StageCfg<SUB, W, false, true, PFS, 1>>(g, h, t0, size, full, T, V, lane, ot, fin, hand[grp][j & 1], nullptr);
            }
        }
        __syncthreads();
    }
    if (live && active) {
        VStore<false> st{slots + v, stride, 0};
        VGate::W<VStore<false>> gate{&st, true};
        if (stage == 0) S0::visit(g, gate); else S1::visit(g, gate);
    }
}

template <class G, int MODE, int S, int K1, int GPW>
__global__ __launch_bounds__((64 * GPW * PlanarPlan<G>::template Tiles<S, K1>::WAVES)) void k_render_pipe_planar(
    float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, size_t T, size_t fstride,
    const void* aux, float* ring, uint32_t ring_cap) {
    render_pipe_planar_body<G, MODE, S, K1, GPW>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
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
// heavy run-time compiled graphs also get the pipeline kernel for workgroups of 1 / 2 voice groups (small banks)
template <class G>
struct JitPipeSmall {
    static constexpr PipePlan P = pipe_plan<G>(0);
    static constexpr bool on = P.S >= 1 && Cost<G>::v >= 150;
    template <int GPW> static constexpr int threads() { return on ? 16 * GPW * PipeGeom<G::IN, (P.S >= 1 ? P.S : 1)>::WAVES : 64; }
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
    constexpr PipePlan P = pipe_plan<G>(0);  // pipeline kernel plan of run-time compiled graphs (0 stages = not used)
    meta[5] = P.S;
    meta[6] = P.S >= 1 ? 64 * PipeGeom<G::IN, (P.S >= 1 ? P.S : 1)>::WAVES : 0;
    meta[7] = PlanarPlan<G>::S >= 1 ? 256 * PlanarPlan<G>::T::WAVES : 0;  // threads of the planar pipeline kernel (0 = none)
    meta[8] = SameType<typename FastOf<G>::type, G>::v ? 0 : 1;            // the graph has a tolerance-mode variant
    meta[9] = JitPipeSmall<G>::on ? 1 : 0;                                 // heavy: jit_pipe_g1 / _g2 exist for small banks
    meta[10] = PipeMinT<G>::v;                                             // launch length from which the pipeline kernel is taken
    meta[11] = TsPlan<G>::ok ? 1 : 0;                                      // a three-stage generator chain: small banks take the time-split kernels (jit_ts3_g1 / _g2)
    meta[12] = WideChain<G>::on ? WideChain<G>::W : 0;                     // a wide sum of generators: waves of the chain kernel (jit_wide_*) small banks take, 0 = none
}
// the three-way time-split kernels for run-time compiled graphs (small banks of three-stage generator chains; a module of their own with the
// mix-down kernels, compiled on first use): empty when the graph does not qualify
template <class G, int GPW>
FD_D void jit_ts3_body(float* __restrict__ slots, size_t stride, size_t V, float* __restrict__ out, size_t T, const void* aux) {
    if constexpr (TsPlan<G>::ok) render_ts3_body<G, GPW>(slots, stride, V, out, T, aux);
}
// ... and with the fused mix-down (k_render_ts3_mix): MIX_SUM for graphs of one or two outputs, MIX_PAN for mono graphs
template <class G, int GPW, int MIX>
FD_D void jit_ts3_mix_body(float* __restrict__ slots, size_t stride, size_t V, float* __restrict__ part, size_t T, const void* aux, const float* __restrict__ panw) {
    if constexpr (TsPlan<G>::ok && (MIX == MIX_SUM ? G::OUT <= 2 : G::OUT == 1)) render_ts3_body<G, GPW, MIX>(slots, stride, V, part, T, aux, panw);
}

// pipeline kernel entry for run-time compiled graphs: empty when the graph has no plan
template <class G, int MODE>
FD_D void jit_pipe_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                        float* __restrict__ out, size_t T, const void* aux, float* ring, uint32_t ring_cap) {
    constexpr PipePlan P = pipe_plan<G>(0);
    if constexpr (P.S >= 1) render_pipe_body<G, MODE, P.S, P.K1, P.K2>(slots, stride, V, in, out, T, aux, ring, ring_cap);
}
// the same for workgroups of GPW = 1 / 2 voice groups: heavy graphs on small banks (see launch_render_pipe); empty otherwise
template <class G, int MODE, int GPW>
FD_D void jit_pipe_small_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                              float* __restrict__ out, size_t T, const void* aux, float* ring, uint32_t ring_cap) {
    constexpr PipePlan P = pipe_plan<G>(0);
    if constexpr (JitPipeSmall<G>::on) render_pipe_body<G, MODE, P.S, P.K1, P.K2, GPW>(slots, stride, V, in, out, T, aux, ring, ring_cap);
}
// ... with the fused mix-down (fdsp_bank_process_mix on run-time compiled graphs; compiled on first use, four voice groups per workgroup --
// the partial mixes do not depend on the launch geometry); empty when the graph has no plan / MIX_PAN on a graph that is not mono
template <class G, int MODE, int MIX>
FD_D void jit_pipe_mix_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                            float* __restrict__ part, size_t T, const void* aux, float* ring, uint32_t ring_cap, const float* __restrict__ panw) {
    constexpr PipePlan P = pipe_plan<G>(0);
    if constexpr (P.S >= 1 && (MIX == MIX_SUM || G::OUT == 1)) {
        // (graphs of four or more outputs: a mix tile of 8 frames does not fit beside the pipeline's tiles -- the host refuses such a
        // launch up front, fd_jit.hip jit_mix_channels_ok, and this body stays empty so that the module still builds)
        if constexpr (MixGeom<(MIX == MIX_PAN ? 1 : G::OUT), 4, PipeTiles<G, P.S, P.K1, P.K2, 4>::SUB>::ok)
            render_pipe_body<G, MODE, P.S, P.K1, P.K2, 4, MIX>(slots, stride, V, in, part, T, aux, ring, ring_cap, panw);
    }
}
constexpr bool jit_mix_channels_ok(int channels) { return (31 * 1024) / 4 / (channels * MIX_ROW * 4) >= 8; }  // MixGeom<channels, 4, .>::ok, for the host
static_assert(jit_mix_channels_ok(3) && !jit_mix_channels_ok(4) && MixGeom<3, 4, 64>::ok && !MixGeom<4, 4, 64>::ok, "fused mix-down: at most three output channels at four voice groups per workgroup");
template <class G, int MODE>
FD_D void jit_events_mix_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ part, size_t T,
                              const double* __restrict__ ev, const int* __restrict__ fade, double time0, double sr, const void* aux, float* ring,
                              uint32_t ring_cap) {
    if constexpr (G::OUT <= 2) render_events_body<G, MODE, true>(slots, stride, V, in, part, T, ev, fade, time0, sr, aux, ring, ring_cap);
}
template <class G, int MODE>
FD_D void jit_pipe_planar_body(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in,
                               float* __restrict__ out, size_t T, size_t fstride, const void* aux, float* ring, uint32_t ring_cap) {
    using PP = PlanarPlan<G>;
    if constexpr (PP::S >= 1) render_pipe_planar_body<G, MODE, PP::S, PP::K1>(slots, stride, V, in, out, T, fstride, aux, ring, ring_cap);
}
template <class G>
struct JitPipePlanarThreads {
    static constexpr int v = PlanarPlan<G>::S >= 1 ? 256 * PlanarPlan<G>::T::WAVES : 64;
};
template <class G>
struct JitPipeThreads {
    static constexpr PipePlan P = pipe_plan<G>(0);
    static constexpr int v = P.S >= 1 ? 64 * PipeGeom<G::IN, (P.S >= 1 ? P.S : 1)>::WAVES : 64;
};

}  // namespace fd
