// fd_nodes.hpp -- device-side node library of the MI355X voice-bank engine.
//
// One wavefront lane evaluates one voice.  A voice graph is a C++ type built from the templates below, which
// mirror FunDSP's statically typed combinator tree (reference src/audionode.rs: Pipe :1375, Stack :1496,
// Binop :850, Unop :1232, Constant :465) and its leaf DSP nodes.  Because the whole tree is one type, the
// compiler fuses a voice into straight-line VALU code with every per-voice coefficient and every piece of
// IIR state held in VGPRs for the whole launch -- the block temporaries (`BufferArray`) the CPU combinators
// pass through memory never exist here.
//
// Per-voice data lives in HBM as a structure of arrays `slots[slot][voice]`; every node enumerates its fields
// through `visit(v)` in a fixed order, so one visitor pattern provides coalesced load / store of the lane's
// registers, host-side introspection (slot names) and state snapshots (FunDSP nodes are `Clone`).
//
// Semantics per node follow the reference `tick` (scalar path) and, where the reference overrides it with a
// different arithmetic, its `process` (block path): `step<true>` = sample inside a full 8-sample SIMD item of
// a block, `step<false>` = per-sample `tick` (block remainder, or tick mode).  Citations are reference file:line.
#pragma once

#include "fd_math.hpp"

#define FD_D __device__ __forceinline__

namespace fd {

enum FieldKind { PARAM = 0, COEF = 1, STATE = 2 };

// Which reference path a sample belongs to (template argument of step / step2):
//   PH_SIMD  process mode, sample inside a full 8-sample SIMD item of the block  -> a node's `process` arithmetic
//   PH_REM   process mode, one of the trailing `size & 7` samples: nodes that end their `process` with
//            process_remainder (Sine, WaveSynth, Shaper) use `tick` here; nodes whose `process` walks the whole
//            block themselves (EnvelopeIn, Noise, Constant, Panner) keep their `process` arithmetic
//   PH_TICK  tick mode: AudioNode::tick for every sample
constexpr int PH_SIMD = 0, PH_REM = 1, PH_TICK = 2;

// Per-launch context handed to every node by bind(): shared wavetable data and this bank's delay-ring memory.
// Ring memory is laid out [ring node][position][voice] (voice-minor, like the audio I/O): all lanes of a wave write the
// same ring position in the same instruction, so ring writes are coalesced 256-B rows; reads at per-voice delays gather.
struct Aux;
struct Ctx {
    const Aux* aux;
    float* ring;         // base of this LANE's column: ring + voice
    uint32_t ring_cap;   // positions per ring node
    size_t vstride;      // floats between consecutive positions (= padded voice count)
    int next_ring;       // running index handed out to ring nodes in visit order
    // Where a node reports that the capacity fixed at bank creation is too small for what it was asked to hold (the
    // reference would resize its buffer): the largest number of positions any node wanted.  One word behind the bank's
    // ring memory, set by the lifecycle kernels only (nullptr in the render kernels); the host checks it after every
    // call that can change a length and fails that call (fd_capi.hip check_ring_need).
    uint32_t* ring_need = nullptr;
    FD_HD float* claim_ring() { return ring + (size_t)(next_ring++) * ring_cap * vstride; }
    FD_HD void want_positions(uint32_t want) const {
        if (ring_need && want > ring_cap) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicMax(ring_need, want);
#else
            if (*ring_need < want) *ring_need = want;
#endif
        }
    }
};

// step2<PH>(in, out): two consecutive frames at once, channel c of frames (n, n+1) packed in one <2 x float>.
// Feed-forward nodes (Constant, Unop, Binop, the sine polynomial of Sine::process) implement it with packed f32
// arithmetic, which halves their instruction count; nodes whose samples depend serially on each other use this
// default, which is two `step` calls.  Either way each frame's arithmetic is identical to `step`.
#define FD_STEP2_VIA_STEP                                                      \
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {           \
        float i0[IN > 0 ? IN : 1], i1[IN > 0 ? IN : 1], o0[OUT], o1[OUT];      \
        _Pragma("unroll") for (int c = 0; c < IN; c++) {                       \
            i0[c] = in[c].x;                                                   \
            i1[c] = in[c].y;                                                   \
        }                                                                      \
        this->template step<PH>(i0, o0);                                     \
        this->template step<PH>(i1, o1);                                     \
        _Pragma("unroll") for (int c = 0; c < OUT; c++) out[c] = v2f{o0[c], o1[c]}; \
    }

constexpr int SVF_LOWPASS = 0, SVF_HIGHPASS = 1, SVF_BANDPASS = 2, SVF_NOTCH = 3, SVF_PEAK = 4, SVF_ALLPASS = 5,
              SVF_BELL = 6, SVF_LOWSHELF = 7, SVF_HIGHSHELF = 8;

// ---------------------------------------------------------------------------------------------------------
// shared coefficient math (host + device): the C ABI exposes the same constructors to host callers
// ---------------------------------------------------------------------------------------------------------
struct SvfCoefs { float a1, a2, a3, m0, m1, m2; };
struct BiquadCoefs { float a1, a2, b0, b1, b2; };

// SvfCoefs::{lowpass .. highshelf}  svf.rs:28-221
FD_HD SvfCoefs svf_coefs(int mode, float sr, float cutoff, float q, float gain) {
    SvfCoefs c;
    float t = tanf_musl(F32_PI * cutoff / sr);
    float a = 0.0f, g = t, k = 1.0f / q;
    if (mode >= SVF_BELL) {
        a = __builtin_sqrtf(gain);
        if (mode == SVF_BELL) k = 1.0f / (q * a);
        if (mode == SVF_LOWSHELF) g = t / __builtin_sqrtf(a);
        if (mode == SVF_HIGHSHELF) g = t * __builtin_sqrtf(a);
    }
    c.a1 = 1.0f / (1.0f + g * (g + k));
    c.a2 = g * c.a1;
    c.a3 = g * c.a2;
    switch (mode) {
    case SVF_LOWPASS: c.m0 = 0.0f; c.m1 = 0.0f; c.m2 = 1.0f; break;
    case SVF_HIGHPASS: c.m0 = 1.0f; c.m1 = -k; c.m2 = -1.0f; break;
    case SVF_BANDPASS: c.m0 = 0.0f; c.m1 = 1.0f; c.m2 = 0.0f; break;
    case SVF_NOTCH: c.m0 = 1.0f; c.m1 = -k; c.m2 = 0.0f; break;
    case SVF_PEAK: c.m0 = 1.0f; c.m1 = -k; c.m2 = -2.0f; break;
    case SVF_ALLPASS: c.m0 = 1.0f; c.m1 = -2.0f * k; c.m2 = 0.0f; break;
    case SVF_BELL: c.m0 = 1.0f; c.m1 = k * (a * a - 1.0f); c.m2 = 0.0f; break;
    case SVF_LOWSHELF: c.m0 = 1.0f; c.m1 = k * (a - 1.0f); c.m2 = a * a - 1.0f; break;
    default: c.m0 = a * a; c.m1 = k * (1.0f - a) * a; c.m2 = 1.0f - a * a; break;
    }
    return c;
}

// BiquadCoefs::{butter_lowpass, resonator, lowpass, highpass, bell}  biquad.rs:27-116
constexpr int BQ_BUTTER = 0, BQ_RESONATOR = 1, BQ_LOWPASS = 2, BQ_HIGHPASS = 3, BQ_BELL = 4;
FD_HD BiquadCoefs biquad_coefs(int kind, float sr, float f0, float q, float gain) {
    BiquadCoefs c;
    if (kind == BQ_BUTTER) {
        float f = tanf_musl(f0 * F32_PI / sr);
        float a0r = 1.0f / (1.0f + F32_SQRT_2 * f + f * f);
        c.a1 = (2.0f * f * f - 2.0f) * a0r;
        c.a2 = (1.0f - F32_SQRT_2 * f + f * f) * a0r;
        c.b0 = f * f * a0r;
        c.b1 = 2.0f * c.b0;
        c.b2 = c.b0;
        return c;
    }
    if (kind == BQ_RESONATOR) {
        float r = expf_musl(-F32_PI * f0 / (q * sr));
        c.a1 = -2.0f * r * cosf_musl(F32_TAU * f0 / sr);
        c.a2 = r * r;
        c.b0 = __builtin_sqrtf(1.0f - r * r) * 0.5f;
        c.b1 = 0.0f;
        c.b2 = -c.b0;
        return c;
    }
    float omega = F32_TAU * f0 / sr;
    float alpha = sinf_musl(omega) / (2.0f * q);
    float beta = cosf_musl(omega);
    if (kind == BQ_LOWPASS) {
        float a0r = 1.0f / (1.0f + alpha);
        c.a1 = -2.0f * beta * a0r;
        c.a2 = (1.0f - alpha) * a0r;
        c.b1 = (1.0f - beta) * a0r;
        c.b0 = c.b1 * 0.5f;
        c.b2 = c.b0;
    } else if (kind == BQ_HIGHPASS) {
        float a0r = 1.0f / (1.0f + alpha);
        c.a1 = -2.0f * beta * a0r;
        c.a2 = (1.0f - alpha) * a0r;
        c.b0 = (1.0f + beta) * 0.5f * a0r;
        c.b1 = (-1.0f - beta) * a0r;
        c.b2 = c.b0;
    } else {
        float a = __builtin_sqrtf(gain);
        float a0r = 1.0f / (1.0f + alpha / a);
        c.a1 = -2.0f * beta * a0r;
        c.a2 = (1.0f - alpha / a) * a0r;
        c.b0 = (1.0f + alpha * a) * a0r;
        c.b1 = c.a1;
        c.b2 = (1.0f - alpha * a) * a0r;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------------
// leaves
// ---------------------------------------------------------------------------------------------------------

// Constant<N>  audionode.rs:465-523 (ID 2).  The value is per voice.
template <int N>
struct Constant {
    static constexpr int IN = 0, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 2;
    float value[N];
    template <class V> FD_HD void visit(V& v) {
        _Pragma("unroll") for (int i = 0; i < N; i++) v.fi(value[i], PARAM, "value", i);
    }
    FD_HD void init() {
        for (int i = 0; i < N; i++) value[i] = 0.0f;
    }
    FD_HD void update(double) {}
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float*, float* out) {
        for (int i = 0; i < N; i++) out[i] = value[i];
    }
    template <int PH> FD_HD void step2(const v2f*, v2f* out) {
        for (int i = 0; i < N; i++) out[i] = splat2(value[i]);
    }
    // skip / skip2: advance the node's STATE over one / two frames without producing their output (time-split stages,
    // fd_device.hpp ts_stage).  Only nodes whose state advance is cheap next to their output define them.
    template <int PH> FD_HD void skip(const float*) {}
    template <int PH> FD_HD void skip2(const v2f*) {}
};

// Pass  audionode.rs:408-436 (ID 48)
struct Pass {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 48;
    template <class V> FD_HD void visit(V&) {}
    FD_HD void init() {}
    FD_HD void update(double) {}
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) { out[0] = in[0]; }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) { out[0] = in[0]; }
};

// Sine<f32>  oscillator.rs:21-102 (ID 21)
struct Sine {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 21;
    float phase, sample_duration, has_phase, initial_phase;
    uint64_t hash;
    float tmax;  // transient guard of the packed sine path (not a slot)
    // A phase of exactly -0.0 is the one argument wide_sin2 gets wrong (+0.0 instead of -0.0, see fd_math.hpp).  The
    // unwrapped phase can only be -0.0 inside a block if it is -0.0 at the block start (-0 + d is -0 only for d = -0),
    // so such a block is sent down the rollback path up front.
    FD_HD void begin_block(int) { tmax = f2u(phase) == 0x80000000u ? __builtin_inff() : 0.0f; }
    FD_HD bool tripped() const { return !(tmax < 8192.0f); }
    FD_HD void bind(Ctx&) {}
    template <class V> FD_HD void visit(V& v) {
        v.f(phase, STATE, "phase");
        v.f(sample_duration, COEF, "sample_duration");
        v.f(has_phase, PARAM, "has_initial_phase");
        v.f(initial_phase, PARAM, "initial_phase");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void init() {  // Sine::new :30-35 (Default -> hash 0, no initial phase; reset; sr = 44100 via update)
        has_phase = 0.0f;
        initial_phase = 0.0f;
        hash = 0;
        reset();
    }
    FD_HD void update(double sr) { sample_duration = (float)(1.0 / sr); }                   // :62-64
    FD_HD void reset() { phase = has_phase != 0.0f ? initial_phase : (float)rnd1(hash); }  // :55-60
    FD_HD uint64_t ping(bool probe, uint64_t h) {                                            // audionode.rs:156-161
        if (!probe) {  // set_hash :94-97
            hash = h;
            reset();
        }
        return atto(h, ID);
    }
    FD_HD void end_simd() { phase = phase - __builtin_floorf(phase); }  // :85 (one wrap after the SIMD items)
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_SIMD) {  // process :74-86: phase runs unwrapped inside the block, f32x8 (wide) sin
            float tmp = phase;
            phase += in[0] * sample_duration;
            out[0] = wide_sinf(tmp * F32_TAU);
        } else {  // tick :67-72: wrapped phase, libm sinf
            float p = phase;
            phase += in[0] * sample_duration;
            phase -= __builtin_floorf(phase);
            out[0] = sinf_musl(p * F32_TAU);
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {  // serial f32 phase recurrence for the two frames, then ONE packed polynomial evaluation
            v2f d = in[0] * sample_duration;
            float t0 = phase;
            phase += d.x;
            float t1 = phase;
            phase += d.y;
            out[0] = wide_sin2(v2f{t0, t1} * F32_TAU, tmax);
        } else {
            float o0, o1, i0 = in[0].x, i1 = in[0].y;
            this->template step<PH>(&i0, &o0);
            this->template step<PH>(&i1, &o1);
            out[0] = v2f{o0, o1};
        }
    }
    // the phase recurrence alone (the same operations step / step2 perform on `phase`); PH_SIMD only
    template <int PH> FD_HD void skip(const float* in) { phase += in[0] * sample_duration; }
    template <int PH> FD_HD void skip2(const v2f* in) {
        v2f d = in[0] * sample_duration;
        phase += d.x;
        phase += d.y;
    }
};

// Sine in tolerance mode (FDSP_MATH_FAST): the phase recurrence of Sine::process is kept operation for operation, the
// f32x8 sine polynomial is replaced by fast_sin (fd_math.hpp; within 1.2e-7 of it).  tick / remainder samples unchanged.
struct SineFast : Sine {
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_SIMD) {
            float tmp = phase;
            phase += in[0] * sample_duration;
            out[0] = fast_sin1(tmp * F32_TAU);
        } else {
            Sine::template step<PH>(in, out);
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {
            v2f d = in[0] * sample_duration;
            float t0 = phase;
            phase += d.x;
            float t1 = phase;
            phase += d.y;
            out[0] = v2f{fast_sin1(t0 * F32_TAU), fast_sin1(t1 * F32_TAU)};
        } else {
            Sine::template step2<PH>(in, out);
        }
    }
};

// Sine with its packed path evaluated as two PLAIN wide_sin1 calls (same operations per frame, bit-identical): for a
// producer stage that shares a SIMD with a prioritised consumer wave -- a 2-cycle plain op holds the VALU half as long as a
// 4-cycle packed one, so it stands in the consumer's way half as long (fd_device.hpp FD_PIPE_PRODUCER_PLAIN).
struct SinePlain : Sine {
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {
            v2f d = in[0] * sample_duration;
            float t0 = phase;
            phase += d.x;
            float t1 = phase;
            phase += d.y;
            out[0] = v2f{wide_sin1(t0 * F32_TAU, tmax), wide_sin1(t1 * F32_TAU, tmax)};
        } else {
            Sine::template step2<PH>(in, out);
        }
    }
};

// Dsf<N>  oscillator.rs:103-208 (ID 55): discrete summation formula oscillator (Moorer 1976), tick only.
// NIN = 1 (frequency) or 2 (frequency, roughness).
template <int NIN>
struct Dsf {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 55;
    float phase, roughness, harmonic_spacing, sample_duration, has_phase, initial_phase;
    uint64_t hash;
    template <class V> FD_HD void visit(V& v) {
        v.f(phase, STATE, "phase");
        v.f(roughness, NIN > 1 ? STATE : PARAM, "roughness");
        v.f(harmonic_spacing, PARAM, "harmonic_spacing");
        v.f(sample_duration, COEF, "sample_duration");
        v.f(has_phase, PARAM, "has_initial_phase");
        v.f(initial_phase, PARAM, "initial_phase");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD static float clamp_roughness(float r) {  // set_roughness :154-157: clamp(0.0001, 0.9999, r)
        return __builtin_fminf(__builtin_fmaxf(r, 0.0001f), 0.9999f);
    }
    FD_HD void init() {  // dsf_saw_r(0.5) defaults; Dsf::new :132-146
        harmonic_spacing = 1.0f; roughness = 0.5f; has_phase = 0.0f; initial_phase = 0.0f; hash = 0;
        reset();
    }
    FD_HD void update(double sr) {  // :168-170; the host-set roughness is clamped like set_roughness does
        sample_duration = (float)(1.0 / sr);
        roughness = clamp_roughness(roughness);
    }
    FD_HD void reset() { phase = has_phase != 0.0f ? initial_phase : (float)rnd1(hash); }  // :161-166
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :198-201
            hash = h;
            reset();
        }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :172-187, dsf :105-113
        if (NIN > 1) roughness = clamp_roughness(in[1]);
        phase += in[0] * sample_duration;
        phase -= __builtin_floorf(phase);
        const float n = __builtin_floorf(22050.0f / in[0] / harmonic_spacing);
        const float f = phase * F32_TAU, d = phase * F32_TAU * harmonic_spacing, r = roughness;
        out[0] = (sinf_musl(f) - r * sinf_musl(f - d) -
                  powf_musl(r, n + 1.0f) * (sinf_musl(f + (n + 1.0f) * d) - r * sinf_musl(f + n * d))) /
                 (1.0f + r * r - 2.0f * r * cosf_musl(d));
    }
    FD_STEP2_VIA_STEP
};

// Noise  noise.rs:173-234 (ID 20).  Integer-exact; process == tick sample for sample.
struct Noise {
    static constexpr int IN = 0, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 20;
    uint32_t state;
    float has_seed;
    uint64_t seed, hash;
    template <class V> FD_HD void visit(V& v) {
        v.u32(state, STATE, "state");
        v.f(has_seed, PARAM, "has_seed");
        v.u64(seed, PARAM, "seed");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void init() {  // Noise::new = Default :179-183 (state 0 until pinged or reset)
        state = 0;
        has_seed = 0.0f;
        seed = 0;
        hash = 0;
    }
    FD_HD void update(double) {}
    FD_HD void reset() {  // :192-195
        uint64_t h = has_seed != 0.0f ? seed : hash;
        state = (uint32_t)(h ^ (h >> 32));
    }
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :226-229
            hash = h;
            reset();
        }
        return atto(h, ID);
    }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float*, float* out) {  // :197-202
        state += 1u;
        out[0] = (float)(hash32x(state) >> 8) * (2.0f / (float)((1 << 24) - 1)) - 1.0f;
    }
    FD_STEP2_VIA_STEP
    // the state is a sample counter: frames another wave evaluates are skipped by counting (time-split stages, fd_device.hpp)
    template <int PH> FD_HD void skip(const float*) { state += 1u; }
    template <int PH> FD_HD void skip2(const v2f*) { state += 2u; }
};

// SVF core shared by FixedSvf and Svf:  svf.rs:995-1006 / :829-843
// one v_max3_f32 vmax, |v1|, |v2| per frame (written as max(max(vmax, |v1|), |v2|): the other association made the
// compiler pair frames up -- a v_max_f32 per frame plus a v_max3 per two)
#define FD_SVF_TRACK(vmax, v1, v2) vmax = __builtin_fmaxf(__builtin_fmaxf(vmax, __builtin_fabsf(v1)), __builtin_fabsf(v2))
struct SvfCore {
    float a1, a2, a3, m0, m1, m2, ic1eq, ic2eq;
    // The reference's operations in the reference's order (svf.rs:995-1006): every scalar path uses this form.
    FD_HD float tick(float v0) {
        float v3 = v0 - ic2eq;
        float v1 = a1 * ic1eq + a2 * v3;
        float v2 = ic2eq + a2 * ic1eq + a3 * v3;
        ic1eq = 2.0f * v1 - ic1eq;
        ic2eq = 2.0f * v2 - ic2eq;
        return m0 * v0 + m1 * v1 + m2 * v2;
    }
    // Packed-path form: `2*v - ic` as fma(2, v, -ic).  2*v is exact in binary floating point, so the fused and the
    // unfused form round the same real number once -- identical bits -- EXCEPT when 2*v overflows (|v| >= 2^127): the
    // reference's product becomes inf there while the fused form stays finite.  `vmax` accumulates max(|v1|, |v2|)
    // (one v_max3_f32); the caller checks it per tile and re-renders the tile with tick() if it reached 2^127.
    FD_HD float tick_fused(float v0, float& vmax) {
        float v3 = v0 - ic2eq;
        float v1 = a1 * ic1eq + a2 * v3;
        float v2 = ic2eq + a2 * ic1eq + a3 * v3;
        FD_SVF_TRACK(vmax, v1, v2);
        ic1eq = __builtin_fmaf(2.0f, v1, -ic1eq);
        ic2eq = __builtin_fmaf(2.0f, v2, -ic2eq);
        return m0 * v0 + m1 * v1 + m2 * v2;
    }
    // ... and when m0 = m1 = +0.0 and m2 = 1.0 (LowpassMode): `m0*v0 + m1*v1 + m2*v2` = (+-0 + +-0) + v2 is v2 itself,
    // bit for bit, whenever v0, v1 are finite and v2 is not -0.0 -- FixedSvfLp guards both (see there).
    FD_HD float tick_lp(float v0, float& vmax) {
        // The two state equations as ONE <2 x float> computation (same operations, same order, per component): a lone
        // wave issues one VALU instruction per ~4-5 cycles whatever its width (profiles/r03_ubench_issue.txt), and this
        // recurrence is what the filter wave's issue slots go to -- 8 slots per frame instead of 11.
        //   lane 0: v1 = a1*ic1 + a2*v3          lane 1: v2 = (ic2 + a2*ic1) + a3*v3
        const float v3 = v0 - ic2eq;
        const float u = a1 * ic1eq;
        const float s = ic2eq + a2 * ic1eq;
        const v2f q = v2f{v3, v3} * v2f{a2, a3};
        const v2f v = v2f{u, s} + q;
        FD_SVF_TRACK(vmax, v.x, v.y);
        const v2f ic = __builtin_elementwise_fma(splat2(2.0f), v, -v2f{ic1eq, ic2eq});
        ic1eq = ic.x;
        ic2eq = ic.y;
        return v.y;
    }
    FD_HD void set(const SvfCoefs& c) {
        a1 = c.a1; a2 = c.a2; a3 = c.a3; m0 = c.m0; m1 = c.m1; m2 = c.m2;
    }
};

// FixedSvf<f32, M>  svf.rs:861-1031 (ID 43).  The mode is a per-voice parameter: the recurrence is mode
// independent (17 flops with generic m0..m2, exactly the reference arithmetic); only `update` branches on it.
struct FixedSvf {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 43;
    float mode, cutoff, q, gain, sr;
    SvfCore c;
    float vmax;  // transient guard of the packed path (not a slot): SvfCore::tick_fused
    template <class V> FD_HD void visit(V& v) {
        v.f(mode, PARAM, "mode");
        v.f(cutoff, PARAM, "cutoff");
        v.f(q, PARAM, "q");
        v.f(gain, PARAM, "gain");
        v.f(sr, COEF, "sample_rate");
        v.f(c.a1, COEF, "a1"); v.f(c.a2, COEF, "a2"); v.f(c.a3, COEF, "a3");
        v.f(c.m0, COEF, "m0"); v.f(c.m1, COEF, "m1"); v.f(c.m2, COEF, "m2");
        v.f(c.ic1eq, STATE, "ic1eq");
        v.f(c.ic2eq, STATE, "ic2eq");
    }
    FD_HD void init() {
        mode = (float)SVF_LOWPASS; cutoff = 440.0f; q = 1.0f; gain = 1.0f;
        c.ic1eq = 0.0f; c.ic2eq = 0.0f;
    }
    FD_HD void update(double sample_rate) {  // set_sample_rate :989-992 -> update_frequency -> update :236-239
        sr = (float)sample_rate;
        c.set(svf_coefs((int)mode, sr, cutoff, q, gain));
    }
    FD_HD void reset() { c.ic1eq = 0.0f; c.ic2eq = 0.0f; }  // :984-987
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) { vmax = 0.0f; }
    FD_HD bool tripped() const { return !(vmax < 0x1p127f); }
    FD_HD void bind(Ctx&) { vmax = 0.0f; }
    template <int PH> FD_HD void step(const float* in, float* out) { out[0] = c.tick(in[0]); }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {  // a caller of the packed path checks tripped() per tile and redoes the tile with step()
            const float a = c.tick_fused(in[0].x, vmax);
            const float b = c.tick_fused(in[0].y, vmax);
            out[0] = v2f{a, b};
        } else {
            out[0] = v2f{c.tick(in[0].x), c.tick(in[0].y)};
        }
    }
};

// FixedSvf in a wave whose lanes are ALL lowpass filters (m0 = m1 = +0.0, m2 = 1.0) and none of which starts the
// launch with ic2eq = -0.0: the packed path returns v2 directly (5 of the 17 flops are `0*v0 + 0*v1 + 1*v2`).
// Exactness: with those coefficients the reference's sum is (+-0 + +-0) + v2, which is v2 bit for bit unless
//   (a) v2 is -0.0 -- impossible here: v2 = (ic2 + a2*ic1) + a3*v3 is -0.0 only if ic2 is, and ic2' = fma(2, v2, -ic2)
//       is -0.0 only if v2 is -0.0 while ic2 is +0.0, so a launch that starts with ic2 != -0.0 never produces one; or
//   (b) v0 or v1 is infinite (0 * inf = NaN in the reference) -- then ic1eq / ic2eq become and stay non-finite, so a
//       tile whose END state is finite had finite v0, v1, v2 throughout; tripped() checks exactly that and the caller
//       re-renders the tile from its snapshot with the generic arithmetic (`step` below is the inherited generic one).
// The render kernels pick this type per wave (lp_ok below), never the host.
struct FixedSvfLp : FixedSvf {
    FD_HD bool tripped() const {  // vmax < 2^127 also rules out infinite v1 / v2; the state test catches NaNs
        return !(vmax < 0x1p127f && __builtin_fabsf(c.ic1eq) < __builtin_inff() && __builtin_fabsf(c.ic2eq) < __builtin_inff());
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {
            const float a = c.tick_lp(in[0].x, vmax);
            const float b = c.tick_lp(in[0].y, vmax);
            out[0] = v2f{a, b};
        } else {
            FixedSvf::template step2<PH>(in, out);
        }
    }
};
FD_HD bool svf_is_plain_lowpass(const FixedSvf& f) {
    return f2u(f.c.m0) == 0u && f2u(f.c.m1) == 0u && f2u(f.c.m2) == 0x3f800000u && f2u(f.c.ic2eq) != 0x80000000u;
}

// Svf<f32, M> with parameter inputs  svf.rs:748-855 (ID 36).  NIN = 3 (audio, cutoff, q) or 4 (+ gain).
template <int NIN>
struct Svf {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 36;
    float mode, cutoff, q, gain, sr;
    SvfCore c;
    template <class V> FD_HD void visit(V& v) {
        v.f(mode, PARAM, "mode");
        v.f(cutoff, STATE, "cutoff");
        v.f(q, STATE, "q");
        v.f(gain, STATE, "gain");
        v.f(sr, COEF, "sample_rate");
        v.f(c.a1, STATE, "a1"); v.f(c.a2, STATE, "a2"); v.f(c.a3, STATE, "a3");
        v.f(c.m0, STATE, "m0"); v.f(c.m1, STATE, "m1"); v.f(c.m2, STATE, "m2");
        v.f(c.ic1eq, STATE, "ic1eq");
        v.f(c.ic2eq, STATE, "ic2eq");
    }
    FD_HD void init() {
        mode = (float)(NIN == 4 ? SVF_BELL : SVF_LOWPASS); cutoff = 440.0f; q = 1.0f; gain = 1.0f;
        c.ic1eq = 0.0f; c.ic2eq = 0.0f;
    }
    FD_HD void update(double sample_rate) {  // :823-826
        sr = (float)sample_rate;
        c.set(svf_coefs((int)mode, sr, cutoff, q, gain));
    }
    FD_HD void reset() { c.ic1eq = 0.0f; c.ic2eq = 0.0f; }  // :818-821
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        // update_inputs :299-313 (3 inputs) / :588-606 (4 inputs): recompute only when an input changed
        bool changed = in[1] != cutoff || in[2] != q;
        if (NIN == 4) changed = changed || in[3] != gain;
        if (changed) {
            cutoff = in[1];
            q = in[2];
            if (NIN == 4) gain = in[3];
            c.set(svf_coefs((int)mode, sr, cutoff, q, gain));
        }
        out[0] = c.tick(in[0]);
    }
    FD_STEP2_VIA_STEP
};

// Biquad<f32>  biquad.rs:136-218 (ID 15): DF1, coefficients are raw parameters (set_sample_rate keeps them).
// One lane of BiquadBank<f32x8> (biquad_bank.rs:73-84, ID 98) performs exactly this arithmetic.
template <uint64_t NODE_ID>
struct BiquadT {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = NODE_ID;
    float a1, a2, b0, b1, b2, x1, x2, y1, y2;
    template <class V> FD_HD void visit(V& v) {
        v.f(a1, PARAM, "a1"); v.f(a2, PARAM, "a2");
        v.f(b0, PARAM, "b0"); v.f(b1, PARAM, "b1"); v.f(b2, PARAM, "b2");
        v.f(x1, STATE, "x1"); v.f(x2, STATE, "x2"); v.f(y1, STATE, "y1"); v.f(y2, STATE, "y2");
    }
    FD_HD void init() { a1 = a2 = b0 = b1 = b2 = 0.0f; reset(); }
    FD_HD void update(double) {}                        // :179-181
    FD_HD void reset() { x1 = x2 = y1 = y2 = 0.0f; }   // :172-177
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    FD_HD float tick(float x0) {  // :184-194
        float y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
        x2 = x1;
        x1 = x0;
        y2 = y1;
        y1 = y0;
        return y0;
    }
    template <int PH> FD_HD void step(const float* in, float* out) { out[0] = tick(in[0]); }
    // The DF1 expression at its own seam (biquad.rs:186-188: `b0*x0 + b1*x1 + b2*x2 - a1*y1 - a2*y2`, evaluated left to right):
    //   p  = (b0*x0 + b1*x1) + b2*x2      feed-forward: no y in it, so two frames are one packed computation (ff2) and the x history of
    //                                      any later frame is known without evaluating the frames before it (ff_skip2)
    //   y0 = (p - a1*y1) - a2*y2          the recurrence: four dependent-issue operations per sample (fb)
    // The same operations in the same order as tick(), so every split below is bit-identical to it.  A serial wave issues one VALU
    // instruction per ~4.4 cycles whatever it depends on: tick() twice is 18 of them per frame pair, ff2 + 2 x fb is 13;
    // and as a CHAIN OF TWO STAGES (fd_device.hpp Seg<BiquadT>) the recurrence wave carries 8.
    FD_HD v2f ff2(v2f x) {
        const v2f p = (x * b0 + v2f{x1, x.x} * b1) + v2f{x2, x1} * b2;
        x2 = x.x;
        x1 = x.y;
        return p;
    }
    FD_HD float ff(float x0) {
        const float p = (b0 * x0 + b1 * x1) + b2 * x2;
        x2 = x1;
        x1 = x0;
        return p;
    }
    FD_HD void ff_skip2(v2f x) { x2 = x.x; x1 = x.y; }
    FD_HD void ff_skip(float x0) { x2 = x1; x1 = x0; }
    FD_HD float fb(float p) {
        const float y0 = (p - a1 * y1) - a2 * y2;
        y2 = y1;
        y1 = y0;
        return y0;
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        const v2f p = ff2(in[0]);
        const float ya = fb(p.x);
        out[0] = v2f{ya, fb(p.y)};
    }
};
using Biquad = BiquadT<15>;

// BiquadBank<f32x8>  biquad_bank.rs:14-130 (ID 98) as ONE voice: 8 channels in, 8 out, each lane its own coefficients
// (Setting::biquad(..).index(i)).  Config 2 instead spreads the lanes of many banks over voices.
struct BiquadBank {
    static constexpr int IN = 8, OUT = 8, RINGS = 0;
    static constexpr uint64_t ID = 98;
    BiquadT<98> lane[8];
    template <class V> FD_HD void visit(V& v) { for (int i = 0; i < 8; i++) { v.enter(i); lane[i].visit(v); v.leave(); } }
    FD_HD void init() { for (int i = 0; i < 8; i++) lane[i].init(); }
    FD_HD void update(double) {}
    FD_HD void reset() { for (int i = 0; i < 8; i++) lane[i].reset(); }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        _Pragma("unroll") for (int i = 0; i < 8; i++) out[i] = lane[i].tick(in[i]);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {  // every lane: packed feed-forward half, then its two recurrence steps
        _Pragma("unroll") for (int i = 0; i < 8; i++) lane[i].template step2<PH>(in + i, out + i);
    }
};

// ButterLowpass<f32, N>  biquad.rs:227-300 (ID 16), N = 1 (fixed) or 2 (cutoff input)
template <int NIN>
struct ButterLowpass {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 16;
    float cutoff, sr;
    Biquad b;
    template <class V> FD_HD void visit(V& v) {
        v.f(cutoff, NIN > 1 ? STATE : PARAM, "cutoff");
        v.f(sr, COEF, "sample_rate");
        v.f(b.a1, NIN > 1 ? STATE : COEF, "a1"); v.f(b.a2, NIN > 1 ? STATE : COEF, "a2");
        v.f(b.b0, NIN > 1 ? STATE : COEF, "b0"); v.f(b.b1, NIN > 1 ? STATE : COEF, "b1");
        v.f(b.b2, NIN > 1 ? STATE : COEF, "b2");
        v.f(b.x1, STATE, "x1"); v.f(b.x2, STATE, "x2"); v.f(b.y1, STATE, "y1"); v.f(b.y2, STATE, "y2");
    }
    FD_HD void set_cutoff(float f) {  // :246-250
        BiquadCoefs c = biquad_coefs(BQ_BUTTER, sr, f, 0.0f, 0.0f);
        b.a1 = c.a1; b.a2 = c.a2; b.b0 = c.b0; b.b1 = c.b1; b.b2 = c.b2;
        cutoff = f;
    }
    FD_HD void init() { cutoff = 440.0f; b.init(); }
    FD_HD void update(double sample_rate) {  // :263-267
        sr = (float)sample_rate;
        set_cutoff(cutoff);
    }
    FD_HD void reset() { b.reset(); }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :269-277
        if (NIN > 1) {
            if (in[1] != cutoff) set_cutoff(in[1]);
        }
        out[0] = b.tick(in[0]);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if constexpr (NIN == 1) {  // fixed coefficients: the biquad's own packed form (feed-forward half as one <2 x float> computation)
            b.template step2<PH>(in, out);
        } else {                   // a cutoff input may move the coefficients between the two frames
            float i0[NIN], i1[NIN], o0, o1;
            _Pragma("unroll") for (int c = 0; c < NIN; c++) { i0[c] = in[c].x; i1[c] = in[c].y; }
            this->template step<PH>(i0, &o0);
            this->template step<PH>(i1, &o1);
            out[0] = v2f{o0, o1};
        }
    }
};

// Resonator<f32, N>  biquad.rs:310-380 (ID 17), N = 1 (fixed) or 3 (center, q inputs)
template <int NIN>
struct Resonator {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 17;
    float center, q, sr;
    Biquad b;
    template <class V> FD_HD void visit(V& v) {
        v.f(center, NIN > 1 ? STATE : PARAM, "center");
        v.f(q, NIN > 1 ? STATE : PARAM, "q");
        v.f(sr, COEF, "sample_rate");
        v.f(b.a1, NIN > 1 ? STATE : COEF, "a1"); v.f(b.a2, NIN > 1 ? STATE : COEF, "a2");
        v.f(b.b0, NIN > 1 ? STATE : COEF, "b0"); v.f(b.b1, NIN > 1 ? STATE : COEF, "b1");
        v.f(b.b2, NIN > 1 ? STATE : COEF, "b2");
        v.f(b.x1, STATE, "x1"); v.f(b.x2, STATE, "x2"); v.f(b.y1, STATE, "y1"); v.f(b.y2, STATE, "y2");
    }
    FD_HD void set_center_q(float c0, float q0) {  // :330-335
        BiquadCoefs c = biquad_coefs(BQ_RESONATOR, sr, c0, q0, 0.0f);
        b.a1 = c.a1; b.a2 = c.a2; b.b0 = c.b0; b.b1 = c.b1; b.b2 = c.b2;
        center = c0;
        q = q0;
    }
    FD_HD void init() { center = 440.0f; q = 1.0f; b.init(); }
    FD_HD void update(double sample_rate) {  // :349-352
        sr = (float)sample_rate;
        set_center_q(center, q);
    }
    FD_HD void reset() { b.reset(); }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :354-366
        if (NIN >= 3) {
            if (in[1] != center || in[2] != q) set_center_q(in[1], in[2]);
        }
        out[0] = b.tick(in[0]);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if constexpr (NIN == 1) {  // fixed coefficients: the biquad's own packed form
            b.template step2<PH>(in, out);
        } else {
            float i0[NIN], i1[NIN], o0, o1;
            _Pragma("unroll") for (int c = 0; c < NIN; c++) { i0[c] = in[c].x; i1[c] = in[c].y; }
            this->template step<PH>(i0, &o0);
            this->template step<PH>(i1, &o1);
            out[0] = v2f{o0, o1};
        }
    }
};

// Moog<f32, N>  moog.rs:17-117 (ID 60).  N = 1 (fixed cutoff/q) or 3 (audio, cutoff Hz, Q inputs; the
// coefficients including a sinf are then recomputed EVERY sample, unconditionally: moog.rs:83-85).
template <int NIN>
struct Moog {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 60;
    float q, cutoff, sr, rez, p, k, s0, s1, s2, s3, px, ps0, ps1, ps2;
    template <class V> FD_HD void visit(V& v) {
        constexpr FieldKind PK = NIN > 1 ? STATE : PARAM, CK = NIN > 1 ? STATE : COEF;
        v.f(cutoff, PK, "cutoff"); v.f(q, PK, "q");
        v.f(sr, COEF, "sample_rate");
        v.f(rez, CK, "rez"); v.f(p, CK, "p"); v.f(k, CK, "k");
        v.f(s0, STATE, "s0"); v.f(s1, STATE, "s1"); v.f(s2, STATE, "s2"); v.f(s3, STATE, "s3");
        v.f(px, STATE, "px"); v.f(ps0, STATE, "ps0"); v.f(ps1, STATE, "ps1"); v.f(ps2, STATE, "ps2");
    }
    FD_HD void set_cutoff_q(float cutoff_, float q_) {  // :48-57
        cutoff = cutoff_;
        q = q_;
        float c = 2.0f * cutoff / sr;
        p = c * (1.8f - 0.8f * c);
        k = 2.0f * sinf_musl(c * F32_PI * 0.5f) - 1.0f;
        float t1 = (1.0f - p) * 1.386249f;
        float t2 = 12.0f + t1 * t1;
        rez = q * (t2 + 6.0f * t1) / (t2 - 6.0f * t1);
    }
    FD_HD void init() {  // prelude.rs:551-553 moog() = Moog::new(1000.0, 0.1)
        cutoff = 1000.0f;
        q = 0.1f;
        reset();
    }
    FD_HD void update(double sample_rate) {  // :76-79
        sr = (float)sample_rate;
        set_cutoff_q(cutoff, q);
    }
    FD_HD void reset() { s0 = s1 = s2 = s3 = px = ps0 = ps1 = ps2 = 0.0f; }  // :65-74
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    uint32_t wmax;  // transient guard of the packed path (not a slot): largest |tanh argument| bits seen, see tanhf_common
    FD_HD void begin_block(int) { wmax = 0u; }
    FD_HD bool tripped() const { return wmax > TANH_COMMON_MAX_BITS; }
    FD_HD void bind(Ctx&) { wmax = 0u; }
    // Packed path (full SIMD items of a process block): the same recurrence with the saturator evaluated by tanhf_common --
    // exact for |argument| <= 7.5; beyond (or NaN) the tile is re-rendered through step() from the caller's snapshot.
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {
            float o[2];
            bool same = NIN == 1;
#if defined(__HIP_DEVICE_COMPILE__)
            if (NIN > 1) {
                // (cutoff, q) of BOTH samples against the stored pair in one wave-uniform test: four xor/or instructions,
                // one compare and a scalar branch per PAIR, where the per-sample form below costs a compare pair, an
                // exec-mask save, a branch and a restore per SAMPLE on the stage's single wave
                const uint32_t chg = (f2u(in[1].x) ^ f2u(cutoff)) | (f2u(in[2].x) ^ f2u(q)) | (f2u(in[1].y) ^ f2u(cutoff)) | (f2u(in[2].y) ^ f2u(q));
                same = __builtin_amdgcn_ballot_w64(chg != 0u) == 0ull;
            }
#endif
            if (__builtin_expect(same, 1)) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    float x = -rez * s3 + (j ? in[0].y : in[0].x);
                    s0 = (x + px) * p - k * s0;
                    s1 = (s0 + ps0) * p - k * s1;
                    s2 = (s1 + ps1) * p - k * s2;
                    s3 = tanhf_common((s2 + ps2) * p - k * s3, wmax);
                    px = x;
                    ps0 = s0;
                    ps1 = s1;
                    ps2 = s2;
                    o[j] = s3;
                }
                out[0] = v2f{o[0], o[1]};
                return;
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float i0 = j ? in[0].y : in[0].x;
                if (NIN > 1) {
                    const float i1 = j ? in[1].y : in[1].x, i2 = j ? in[2].y : in[2].x;
                    if (f2u(i1) != f2u(cutoff) || f2u(i2) != f2u(q)) set_cutoff_q(i1, i2);
                }
                // (the five products with the previous sample's state as packed multiplies were tried: every packed result feeds
                // the next instruction of this serial chain and costs a wait state -- 101 issue slots per sample against 99)
                float x = -rez * s3 + i0;
                s0 = (x + px) * p - k * s0;
                s1 = (s0 + ps0) * p - k * s1;
                s2 = (s1 + ps1) * p - k * s2;
                s3 = tanhf_common((s2 + ps2) * p - k * s3, wmax);
                px = x;
                ps0 = s0;
                ps1 = s1;
                ps2 = s2;
                o[j] = s3;
            }
            out[0] = v2f{o[0], o[1]};
            return;
        }
        float i0[NIN], i1[NIN], o0, o1;
#pragma unroll
        for (int c = 0; c < NIN; c++) {
            i0[c] = in[c].x;
            i1[c] = in[c].y;
        }
        this->template step<PH>(i0, &o0);
        this->template step<PH>(i1, &o1);
        out[0] = v2f{o0, o1};
    }
    template <int PH> FD_HD void step(const float* in, float* out) {  // :82-100
        // The reference recomputes (p, k, rez) from (cutoff, q) on EVERY sample (moog.rs:83-85).  They are a pure
        // function of the BITS of the two inputs and the sample rate, so recomputing only when an input's bit pattern
        // differs from the stored one yields the same registers bit for bit and keeps the f64 sine out of the
        // steady-state loop.  The test is on bits, not values: +0.0 == -0.0 as values, but cutoff = -0.0 gives
        // c = -0.0 and p = -0.0 where +0.0 gives +0.0 (and the stored `cutoff` word itself is readable state).
        if (NIN > 1) {
            if (f2u(in[1]) != f2u(cutoff) || f2u(in[2]) != f2u(q)) set_cutoff_q(in[1], in[2]);
        }
        float x = -rez * s3 + in[0];
        s0 = (x + px) * p - k * s0;
        s1 = (s0 + ps0) * p - k * s1;
        s2 = (s1 + ps1) * p - k * s2;
        s3 = tanhf_musl((s2 + ps2) * p - k * s3);
        px = x;
        ps0 = s0;
        ps1 = s1;
        ps2 = s2;
        out[0] = s3;
    }
};

// Moog in tolerance mode (FDSP_MATH_FAST): the ladder recurrence is kept operation for operation (unfused), only the
// saturating tanh of the last stage is fast_tanh1.  The coefficient formulas run when (cutoff, q) change, not per
// sample, and stay exact.  tick / remainder samples unchanged.
template <int NIN>
struct MoogFast : Moog<NIN> {
    using B = Moog<NIN>;
    static constexpr int IN = B::IN, OUT = B::OUT;
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_SIMD) {
            if (NIN > 1) {
                if (f2u(in[1]) != f2u(this->cutoff) || f2u(in[2]) != f2u(this->q)) this->set_cutoff_q(in[1], in[2]);
            }
            float x = -this->rez * this->s3 + in[0];
            this->s0 = (x + this->px) * this->p - this->k * this->s0;
            this->s1 = (this->s0 + this->ps0) * this->p - this->k * this->s1;
            this->s2 = (this->s1 + this->ps1) * this->p - this->k * this->s2;
            this->s3 = fast_tanh1((this->s2 + this->ps2) * this->p - this->k * this->s3);
            this->px = x;
            this->ps0 = this->s0;
            this->ps1 = this->s1;
            this->ps2 = this->s2;
            out[0] = this->s3;
        } else {
            B::template step<PH>(in, out);
        }
    }
    FD_STEP2_VIA_STEP
};

// Fir<N>  fir.rs:14-89 (ID 52)
template <int N>
struct Fir {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 52;
    float w[N], v[N];
    template <class V> FD_HD void visit(V& vis) {
        _Pragma("unroll") for (int i = 0; i < N; i++) vis.fi(w[i], PARAM, "w", i);
        _Pragma("unroll") for (int i = 0; i < N; i++) vis.fi(v[i], STATE, "v", i);
    }
    FD_HD void init() {
        for (int i = 0; i < N; i++) w[i] = 0.0f;
        reset();
    }
    FD_HD void update(double) {}
    FD_HD void reset() {
        for (int i = 0; i < N; i++) v[i] = 0.0f;
    }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :57-70
        for (int i = 0; i + 1 < N; i++) v[i] = v[i + 1];
        v[N - 1] = in[0];
        float output = 0.0f;
        for (int i = 0; i < N; i++) output += w[i] * v[i];
        out[0] = output;
    }
    FD_STEP2_VIA_STEP
};

// Tick<N>  delay.rs:19-65 (ID 9): one-sample delay
template <int N>
struct Tick {
    static constexpr int IN = N, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 9;
    float buffer[N];
    template <class V> FD_HD void visit(V& v) {
        _Pragma("unroll") for (int i = 0; i < N; i++) v.fi(buffer[i], STATE, "buffer", i);
    }
    FD_HD void init() { reset(); }
    FD_HD void update(double) {}
    FD_HD void reset() {
        for (int i = 0; i < N; i++) buffer[i] = 0.0f;
    }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :47-52
        for (int i = 0; i < N; i++) {
            float o = buffer[i];
            buffer[i] = in[i];
            out[i] = o;
        }
    }
    FD_STEP2_VIA_STEP
};


// ---------------------------------------------------------------------------------------------------------
// wavetable oscillator, envelope, panner (BASELINE config 4)
// ---------------------------------------------------------------------------------------------------------

// Shared wavetable data (Arc<Wavetable> in the reference, wavetable.rs:82-84): one table set per waveform, in HBM
// (saw: 40 tables, 41 024 floats = 160 KiB -- lives in L2; LDS cannot hold it next to anything else).
constexpr int WT_MAX_TABLES = 48, WT_SETS = 8;  // 0 saw, 1 square, 2 triangle, 3 user, 4 organ, 5 soft saw, 6 hammond, 7 user
struct WtSet {
    int n;
    float pitch[WT_MAX_TABLES];
    int off[WT_MAX_TABLES];
    int len[WT_MAX_TABLES];
    const float* data;
};
// shared sample buffers (the reference's Arc<Wave>: wave.rs), [channel][length] f32 in HBM
constexpr int WAVE_SLOTS = 8;
struct WaveBuf {
    const float* data;
    uint32_t channels, length;
};
struct Aux {
    WtSet wt[WT_SETS];
    WaveBuf wave[WAVE_SLOTS];
};

// A load through a pointer that itself came out of a struct in memory (table / sample-buffer descriptors): say that it
// points to HBM, or the compiler emits flat loads, which count on the LDS counter too and fence the hand-over stores.
FD_HD float gload(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(1))) float*)p;
#else
    return *p;
#endif
}
FD_HD float optimal4x44(float a0, float a1, float a2, float a3, float x) {  // wavetable.rs:24-38
    float z = x - (float)0.5;
    float even1 = a2 + a1, odd1 = a2 - a1, even2 = a3 + a0, odd2 = a3 - a0;
    float c0 = even1 * (float)0.4656725512077848 + even2 * (float)0.03432729708429672;
    float c1 = odd1 * (float)0.5374383075356016 + odd2 * (float)0.1542946255730746;
    float c2 = even1 * (float)-0.25194210134021744 + even2 * (float)0.2519474493593906;
    float c3 = odd1 * (float)-0.46896069955075126 + odd2 * (float)0.15578800670302476;
    float c4 = even1 * (float)0.00986988334359864 + even2 * (float)-0.00989340017126506;
    return (((c4 * z + c3) * z + c2) * z + c1) * z + c0;
}
// The same interpolator for TWO frames at once (component-wise the identical operations in the identical order, no contraction):
// the oscillator stage of config 4 is VALU-issue bound and 448 of its 714 VALU instructions per 8-frame item were these
// multiplies and adds, one frame at a time (profiles/r03_c4_stage0.txt).
FD_HD v2f optimal4x44_2(v2f a0, v2f a1, v2f a2, v2f a3, v2f x) {
    v2f z = x - (float)0.5;
    v2f even1 = a2 + a1, odd1 = a2 - a1, even2 = a3 + a0, odd2 = a3 - a0;
    v2f c0 = even1 * (float)0.4656725512077848 + even2 * (float)0.03432729708429672;
    v2f c1 = odd1 * (float)0.5374383075356016 + odd2 * (float)0.1542946255730746;
    v2f c2 = even1 * (float)-0.25194210134021744 + even2 * (float)0.2519474493593906;
    v2f c3 = odd1 * (float)-0.46896069955075126 + odd2 * (float)0.15578800670302476;
    v2f c4 = even1 * (float)0.00986988334359864 + even2 * (float)-0.00989340017126506;
    return (((c4 * z + c3) * z + c2) * z + c1) * z + c0;
}
FD_HD float clamp01f(float x) {  // math.rs:136-138
    x = x > 0.0f ? x : 0.0f;
    return x < 1.0f ? x : 1.0f;
}
FD_HD int wt_table_index(const WtSet* t, int hint, float frequency) {  // :189-211
    if (frequency >= t->pitch[hint] && frequency <= t->pitch[hint + 1]) return hint;
    int i0 = 0, i1 = t->n - 3;
    while (i0 < i1) {
        int i = (i0 + i1) >> 1;
        if (t->pitch[i] > frequency) {
            i1 = i;
        } else if (t->pitch[i + 1] > frequency) {
            i0 = i;
            break;
        } else {
            i0 = i + 1;
        }
    }
    return i0;
}

// WaveSynth<U1>  wavetable.rs:249-359 (ID 34).  SET selects the shared table set (0 saw, 1 square, 2 triangle).
//
// The descriptor of the current table pair (pitch bounds, offsets, masks) is cached in registers and refreshed only
// when the frequency leaves the hinted table's range -- exactly the condition under which the reference's
// table_index() leaves its fast path (:190-192) -- so a steady voice does no dependent descriptor loads, only the
// 2 x 4 data gathers per sample (served by L2: the saw set is 160 KiB).  step2 issues the 16 gathers of two frames
// back to back so their latencies overlap.
// Device tables are stored circularly padded -- [t[len-1], t[0..len-1], t[0], t[1]] -- so the four interpolation taps
// t[i1-1..i1+2] of Wavetable::at are ONE contiguous (unaligned) 16-byte gather per lane instead of four.
struct Tap4 { float a0, a1, a2, a3, w; };
// the pieces of wt_tap: index + interpolation weight, then the four floats from HBM / from the kernel's LDS copy
FD_HD float wt_tap_index(uint32_t mask, float phase, uint32_t& i1) {
    float p = (float)(mask + 1u) * phase;
    uint32_t i = (uint32_t)p;
    i1 = i & mask;
    return p - (float)i;
}
FD_HD Tap4 wt_tap_mem(const float* __restrict__ at, float w) {  // (by value: a Tap4 passed by reference ended up as a memory object)
    Tap4 t;
    t.w = w;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u q = *(const __attribute__((address_space(1))) f4u*)at;
    t.a0 = q.x; t.a1 = q.y; t.a2 = q.z; t.a3 = q.w;
#else
    float q[4];
    __builtin_memcpy(q, at, 16);
    t.a0 = q[0]; t.a1 = q[1]; t.a2 = q[2]; t.a3 = q[3];
#endif
    return t;
}
FD_HD Tap4 wt_tap(const float* __restrict__ tab, uint32_t mask, float phase) {  // loads of Wavetable::at :154-166
    float p = (float)(mask + 1u) * phase;
    uint32_t i1 = (uint32_t)p;
    Tap4 t;
    t.w = p - (float)i1;
    i1 = i1 & mask;
    // padded layout: tab[i1 + 0..3] = t[i1-1], t[i1], t[i1+1], t[i1+2]
#if defined(__HIP_DEVICE_COMPILE__)
    // through an explicit global-address-space pointer: on a generic pointer (the table address comes out of a struct
    // in memory) the 4-byte-aligned 16-byte load is split into four dword gathers before the address space is inferred
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u q = *(const __attribute__((address_space(1))) f4u*)(tab + i1);
    t.a0 = q.x; t.a1 = q.y; t.a2 = q.z; t.a3 = q.w;
#else
    float q[4];
    __builtin_memcpy(q, tab + i1, 16);
    t.a0 = q[0]; t.a1 = q[1]; t.a2 = q[2]; t.a3 = q[3];
#endif
    return t;
}
FD_HD float tap_eval(const Tap4& t) { return optimal4x44(t.a0, t.a1, t.a2, t.a3, t.w); }
FD_HD v2f tap_eval2(const Tap4 a, const Tap4 b) {  // frames n and n + 1 of one table
    return optimal4x44_2(v2f{a.a0, b.a0}, v2f{a.a1, b.a1}, v2f{a.a2, b.a2}, v2f{a.a3, b.a3}, v2f{a.w, b.w});
}

template <int SET, int NOUT = 1>  // WaveSynth<U2> also outputs the wrapped phase (wavetable.rs:318-324, 343-345)
struct WaveSynth {
    static constexpr int IN = 1, OUT = NOUT, RINGS = 0;
    static constexpr uint64_t ID = 34;
    float phase, sample_duration, has_phase, initial_phase;
    uint32_t hint;
    uint64_t hash;
    // transients
    const WtSet* wt;
    int item_pos;
    float item_w;
    int c_table;  // cached descriptor of tables c_table+1 / c_table+2; -1 = invalid
    float c_p0, c_p1;
    const float *c_tab1, *c_tab2;
    uint32_t c_mask1, c_mask2;
    // the taps gathered ahead for the next frame pair (packed path), valid for exactly the wrapped phases pf_p0 / pf_p1
    bool pf_ok;
    uint32_t pf_p0, pf_p1;
    Tap4 pf_a1, pf_a2, pf_b1, pf_b2;
    template <class V> FD_HD void visit(V& v) {
        v.f(phase, STATE, "phase");
        v.u32(hint, STATE, "table_hint");
        v.f(sample_duration, COEF, "sample_duration");
        v.f(has_phase, PARAM, "has_initial_phase");
        v.f(initial_phase, PARAM, "initial_phase");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx& a) {
        wt = &a.aux->wt[SET];
        c_table = -1;
        pf_drop();
    }
    FD_HD void pf_drop() {  // nothing gathered ahead is valid any more
        pf_ok = false;
    }
    FD_HD void init() {  // WaveSynth::new :270-281: phase 0.0 WITHOUT reset
        phase = 0.0f;
        hint = 0;
        has_phase = 0.0f;
        initial_phase = 0.0f;
        hash = 0;
    }
    FD_HD void update(double sr) { sample_duration = 1.0f / (float)sr; }                    // :299-302
    FD_HD void reset() { phase = has_phase != 0.0f ? initial_phase : (float)rnd1(hash); }  // :292-297
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :304-307
            hash = h;
            reset();
        }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) { item_pos = 0; }
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() { phase = phase - __builtin_floorf(phase); }  // :345
    // == table_index(hint, f0) (:189-211) + crossfade weight (:216-220), with the descriptor cache
    FD_HD float select(float f0) {
        if (!(c_table == (int)hint && f0 >= c_p0 && f0 <= c_p1)) {
            int t = wt_table_index(wt, (int)hint, f0);
            c_table = t;
            c_p0 = wt->pitch[t];
            c_p1 = wt->pitch[t + 1];
            c_tab1 = wt->data + wt->off[t + 1];
            c_tab2 = wt->data + wt->off[t + 2];
            c_mask1 = (uint32_t)wt->len[t + 1] - 1u;
            c_mask2 = (uint32_t)wt->len[t + 2] - 1u;
            hint = (uint32_t)t;
            pf_drop();  // taps gathered ahead came from the previous table pair
        }
        return clamp01f((f0 - c_p0) / (c_p1 - c_p0));
    }
    FD_HD void taps(float ph, Tap4& t1, Tap4& t2) const {  // Wavetable::at of both tables at one phase (:154-166)
        uint32_t i1, i2;
        const float w1 = wt_tap_index(c_mask1, ph, i1), w2 = wt_tap_index(c_mask2, ph, i2);
        t1 = wt_tap_mem(c_tab1 + i1, w1);
        t2 = wt_tap_mem(c_tab2 + i2, w2);
    }
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_SIMD) {  // process :327-348
            // table pair and crossfade from LANE 0's frequency for the whole 8-sample item
            if ((item_pos & 7) == 0) item_w = select(__builtin_fabsf(in[0]));
            item_pos++;
            phase += in[0] * sample_duration;
            float ph = phase - __builtin_floorf(phase);  // wide's inherent f32x8::floor (true floor)
            Tap4 t1, t2;
            taps(ph, t1, t2);
            out[0] = (1.0f - item_w) * tap_eval(t1) + item_w * tap_eval(t2);
            if (NOUT > 1) out[1] = ph;
        } else {  // tick :310-324: increment + wrap BEFORE reading
            float frequency = in[0];
            phase += frequency * sample_duration;
            phase -= __builtin_floorf(phase);
            float w = select(__builtin_fabsf(frequency));
            Tap4 t1, t2;
            taps(phase, t1, t2);
            out[0] = (1.0f - w) * tap_eval(t1) + w * tap_eval(t2);
            if (NOUT > 1) out[1] = phase;
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD && NOUT == 1) {
            if ((item_pos & 7) == 0) item_w = select(__builtin_fabsf(in[0].x));
            item_pos += 2;
            v2f d = in[0] * sample_duration;
            phase += d.x;
            float ph0 = phase - __builtin_floorf(phase);
            phase += d.y;
            float ph1 = phase - __builtin_floorf(phase);
#if defined(__HIP_DEVICE_COMPILE__)
            // The gathers are served by L2 (a saw set is 160 KiB) and the stage has ~35 instructions per frame: issued
            // where they are used, their round trip (~600 cycles per pair) is 2/3 of the stage's time (config 4: stage 0
            // alone 10.4 ms, 33 issue slots per frame).  The next pair's phases are this pair's plus the same two
            // increments whenever the frequency input repeats, so its taps are gathered NOW, before this pair's are
            // evaluated, and the next call takes them if its phases have exactly the predicted bits (wave-uniform test;
            // any other case -- a moving frequency, a new table pair, a rollback -- gathers in place as before).  The
            // values are the table's for those phases either way.
            Tap4 a1, a2, b1, b2;
            const bool hit = pf_ok && f2u(ph0) == pf_p0 && f2u(ph1) == pf_p1;
            if (__builtin_amdgcn_ballot_w64(!hit) == 0ull) {
                a1 = pf_a1; a2 = pf_a2; b1 = pf_b1; b2 = pf_b2;
            } else {
                taps(ph0, a1, a2);
                taps(ph1, b1, b2);
            }
            {
                float np = phase + d.x;
                const float q0 = np - __builtin_floorf(np);
                np += d.y;
                const float q1 = np - __builtin_floorf(np);
                uint32_t ia1, ia2, ib1, ib2;
                const float wa1 = wt_tap_index(c_mask1, q0, ia1), wa2 = wt_tap_index(c_mask2, q0, ia2);
                const float wb1 = wt_tap_index(c_mask1, q1, ib1), wb2 = wt_tap_index(c_mask2, q1, ib2);
                pf_a1 = wt_tap_mem(c_tab1 + ia1, wa1); pf_a2 = wt_tap_mem(c_tab2 + ia2, wa2);
                pf_b1 = wt_tap_mem(c_tab1 + ib1, wb1); pf_b2 = wt_tap_mem(c_tab2 + ib2, wb2);
                pf_p0 = f2u(q0);
                pf_p1 = f2u(q1);
                pf_ok = true;
            }
#else
            // 16 independent gathers in flight before any of them is consumed
            Tap4 a1, a2, b1, b2;
            taps(ph0, a1, a2);
            taps(ph1, b1, b2);
#endif
            out[0] = (1.0f - item_w) * tap_eval2(a1, b1) + item_w * tap_eval2(a2, b2);
        } else {
            float o0[NOUT], o1[NOUT], i0 = in[0].x, i1 = in[0].y;
            this->template step<PH>(&i0, o0);
            this->template step<PH>(&i1, o1);
            for (int c = 0; c < NOUT; c++) out[c] = v2f{o0[c], o1[c]};
        }
    }
};

// PhaseSynth  wavetable.rs:358-430 (ID 35): table read driven by a PHASE input; the table pair follows the frequency
// implied by the phase step (at most Nyquist), 0.5 cycles for the first sample after reset.  No process override.
template <int SET>
struct PhaseSynth {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 35;
    float phase, phase_ready, sample_rate;
    uint32_t hint;
    const WtSet* wt;
    template <class V> FD_HD void visit(V& v) {
        v.f(phase, STATE, "phase");
        v.f(phase_ready, STATE, "phase_ready");
        v.u32(hint, STATE, "table_hint");
        v.f(sample_rate, COEF, "sample_rate");
    }
    FD_HD void bind(Ctx& a) { wt = &a.aux->wt[SET]; }
    FD_HD void init() { phase = 0.0f; phase_ready = 0.0f; hint = 0; sample_rate = 44100.0f; }
    FD_HD void update(double sr) { sample_rate = (float)sr; }
    FD_HD void reset() { phase_ready = 0.0f; }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        float ph = in[0];
        ph = ph - __builtin_floorf(ph);
        float delta;
        if (phase_ready != 0.0f) {
            delta = __builtin_fminf(__builtin_fabsf(ph - phase),
                               __builtin_fminf(__builtin_fabsf(ph - 1.0f - phase), __builtin_fabsf(ph + 1.0f - phase)));
        } else {
            phase_ready = 1.0f;
            delta = 0.5f;
        }
        const float f0 = delta * sample_rate;
        const int t = wt_table_index(wt, (int)hint, f0);  // Wavetable::read :213-226
        hint = (uint32_t)t;
        const float w = clamp01f((f0 - wt->pitch[t]) / (wt->pitch[t + 1] - wt->pitch[t]));
        Tap4 t1 = wt_tap(wt->data + wt->off[t + 1], (uint32_t)wt->len[t + 1] - 1u, ph);
        Tap4 t2 = wt_tap(wt->data + wt->off[t + 2], (uint32_t)wt->len[t + 2] - 1u, ph);
        out[0] = (1.0f - w) * tap_eval(t1) + w * tap_eval(t2);
        phase = ph;
    }
    FD_STEP2_VIA_STEP
};

FD_HD float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }  // math.rs:169-178

// adsr_live(attack, decay, sustain, release) = EnvelopeIn<f32, closure, U1, f32>  (adsr.rs:21-70,
// prelude.rs:626-639, envelope.rs:185-358; ID 53).  The Rust closure + its two atomics become plain per-voice state.
struct AdsrLive {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 53;
    float attack, decay, sustain, release, interval;       // params
    float sd;                                               // coef
    float attacked, attack_start, release_start;            // closure state
    float t, t0, t1, v0, v1, value, value_d;                // EnvelopeIn state
    uint64_t t_hash, hash;
    // block-walk transients (process path)
    int blk_i, blk_size, remaining, loop_len;
    bool full_seg;
    // the next segment, prepared at the head of the block (speculate)
    bool s_valid, s_full;
    int s_loop_len;
    float s_t1, s_v1, s_value, s_value_d;
    uint64_t s_hash;
    // the planned block (step2's packed path, see plan_block)
    bool fast, f_trip;
    int fb;      // sample of the block at which the prepared segment takes over; -1: the current chunk covers the block
    float f_cb;  // the gate at that sample
    template <class V> FD_HD void visit(V& v) {
        v.f(attack, PARAM, "attack"); v.f(decay, PARAM, "decay"); v.f(sustain, PARAM, "sustain");
        v.f(release, PARAM, "release"); v.f(interval, PARAM, "interval");
        v.f(sd, COEF, "sample_duration");
        v.f(attacked, STATE, "attacked"); v.f(attack_start, STATE, "attack_start"); v.f(release_start, STATE, "release_start");
        v.f(t, STATE, "t"); v.f(t0, STATE, "t_0"); v.f(t1, STATE, "t_1");
        v.f(v0, STATE, "value_0"); v.f(v1, STATE, "value_1"); v.f(value, STATE, "value"); v.f(value_d, STATE, "value_d");
        v.u64(t_hash, STATE, "t_hash");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() {
        attack = 0.01f; decay = 0.1f; sustain = 0.6f; release = 0.2f;
        interval = (float)0.002;  // envelope2: F::from_f64(0.002)
        attacked = 0.0f; attack_start = 0.0f; release_start = -1.0f;
        v0 = v1 = value = value_d = 0.0f;
        hash = 0;
        reset();
    }
    FD_HD void update(double sr) { sd = (float)(1.0 / sr); }  // envelope.rs:300-302
    FD_HD void reset() {  // :293-298 (closure state is NOT reset)
        t = 0.0f; t0 = 0.0f; t1 = 0.0f;
        t_hash = hash;
    }
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :346-349: no reset
            hash = h;
            t_hash = h;
        }
        return atto(h, ID);
    }
    // the closure, adsr.rs:36-56, in two parts: the trigger logic (the only place its state changes) ...
    FD_HD bool closure_triggers(float control) const {
        return (release_start >= 0.0f && control > 0.0f) || (release_start < 0.0f && control <= 0.0f);
    }
    // ... and its arithmetic for the CURRENT trigger state
    FD_HD float closure_value(float time) const {
        if (attacked == 0.0f) return 0.0f;
        float tt = time - attack_start, ads;
        if (tt < attack) {
            ads = lerpf(0.0f, 1.0f, tt / attack);
        } else {
            float decay_time = tt - attack;
            ads = decay_time < decay ? lerpf(1.0f, sustain, decay_time / decay) : sustain;
        }
        if (release_start < 0.0f) return ads;
        float a = release_start + release, b = release_start;
        return ads * clamp01f((time - a) / (b - a));
    }
    FD_HD float closure(float time, float control) {
        if (release_start >= 0.0f && control > 0.0f) {
            attacked = 1.0f;
            attack_start = time;
            release_start = -1.0f;
        } else if (release_start < 0.0f && control <= 0.0f) {
            release_start = time;
        }
        return closure_value(time);
    }
    FD_HD void next_segment(float input) {  // envelope.rs:252-278
        if (t0 == 0.0f && t1 == 0.0f) {
            v0 = closure(t0, input);
        } else {
            t0 = t1;
            v0 = v1;
        }
        float next_interval = lerpf(0.75f, 1.25f, (float)rnd1(t_hash)) * interval;
        t1 = t0 + next_interval;
        v1 = closure(t1, input);
        t_hash = t_hash * 6364136223846793005ULL + 1ULL;
        float u = (t - t0) / (t1 - t0);
        value = lerpf(v0, v1, u);
        float samples = next_interval / sd;
        value_d = (v1 - v0) / samples;
    }
    FD_HD void begin_block(int size) { blk_i = 0; blk_size = size; remaining = 0; full_seg = false; loop_len = 0; s_valid = false; fast = false; }
    // A planned block is exact unless the gate triggers at the take-over sample (or the plan could not be made): then the
    // caller re-renders from its snapshot through step(), which walks the block the reference's way
    FD_HD bool tripped() const { return fast && (f_trip || (fb >= 0 && blk_i > fb && closure_triggers(f_cb))); }
    FD_HD void end_simd() {
        if (fast) {  // the bookkeeping the sample-by-sample walk does at its two chunk ends, once
            t += (float)(long long)loop_len * sd;
            if (fb >= 0) {
                t0 = t1; v0 = v1;
                t1 = s_t1; v1 = s_v1; t_hash = s_hash;
                t += (float)(long long)s_loop_len * sd;
            }
            remaining = 0;
            s_valid = false;
            fast = false;
        }
    }
    // Voices reach the ends of their ~2 ms segments at different samples, so in a wave of 64 some lane needs
    // next_segment -- two closure evaluations, seven divisions, a 64-bit hash -- at almost every sample, and the wave
    // pays for it every time.  Everything about a voice's next segment except the trigger logic is known at the head
    // of the block: where the current chunk ends (start_chunk), t there, the jittered interval (t_hash), and the
    // closure's VALUE at the new t_1 as long as the gate does not trigger at that sample.  So all lanes prepare their
    // next segment together, once per block, with the very operations next_segment / start_chunk would perform; at the
    // boundary a lane whose gate sample does not trigger takes the prepared registers, any other lane (gate edge,
    // first segment, a second boundary in the same block at low sample rates) runs next_segment as before.
    FD_HD void speculate() {
        s_valid = false;
        if (!(full_seg && loop_len < blk_size - blk_i) || (t0 == 0.0f && t1 == 0.0f)) return;
        const float nt = t + (float)(long long)loop_len * sd;  // t at the boundary (the update at the end of step)
        const float next_interval = lerpf(0.75f, 1.25f, (float)rnd1(t_hash)) * interval;
        s_t1 = t1 + next_interval;
        s_v1 = closure_value(s_t1);
        s_hash = t_hash * 6364136223846793005ULL + 1ULL;
        const float u = (nt - t1) / (s_t1 - t1);
        s_value = lerpf(v1, s_v1, u);
        const float samples = next_interval / sd;
        s_value_d = (s_v1 - v1) / samples;
        const float c = __builtin_ceilf((s_t1 - nt) / sd);  // start_chunk at the boundary sample
        const long long left = (long long)c;
        const int room = blk_size - (blk_i + loop_len);
        const bool huge = left < 0 || left > (long long)room;
        s_loop_len = huge ? room : (int)left;
        s_full = !huge && s_loop_len == (int)left;
        s_valid = true;
    }
    FD_HD void start_chunk() {  // one iteration head of the `while i < size` loop, envelope.rs:323-326
        float c = __builtin_ceilf((t1 - t) / sd);
        long long left = (long long)c;
        int room = blk_size - blk_i;
        bool huge = left < 0 || left > (long long)room;  // `as usize` of a negative value is huge
        loop_len = huge ? room : (int)left;
        full_seg = !huge && loop_len == (int)left;
        remaining = loop_len;
    }
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_TICK) {  // tick :305-313
            if (t >= t1) next_segment(in[0]);
            out[0] = value;
            value += value_d;
            t += sd;
        } else {  // process :315-340, walked sample by sample (the whole block, no remainder path)
            if (blk_i == 0) {
                fast = false;
                if (t >= t1) next_segment(in[0]);
                start_chunk();
                speculate();
            } else {
                leave_fast();
            }
            for (int guard = 0; remaining == 0 && guard < 4; guard++) {  // chunk exhausted before this sample
                if (full_seg && s_valid && !closure_triggers(in[0])) {  // the segment prepared at the head of the block
                    t0 = t1; v0 = v1;
                    t1 = s_t1; v1 = s_v1; t_hash = s_hash;
                    value = s_value; value_d = s_value_d;
                    loop_len = s_loop_len; full_seg = s_full; remaining = loop_len;
                } else {
                    if (full_seg) next_segment(in[0]);
                    start_chunk();
                }
                s_valid = false;
            }
            out[0] = value;
            value += value_d;
            remaining--;
            blk_i++;
            if (remaining == 0) t += (float)(long long)loop_len * sd;
        }
    }
    // The packed path.  With the next segment prepared at the head of the block (speculate) the walk above still costs the
    // whole wave a masked trip through the take-over code at every sample where ANY of its 64 voices ends a segment --
    // every second sample at 48 kHz (64 voices, ~96-sample segments).  But once the head of the block has run, the block's
    // outputs are decided but for one thing, the gate at the take-over sample: a voice plays its current line up to sample
    // fb and the prepared one from there (segments are longer than a block, so there is at most one take-over; a second
    // one, or a first segment, is left to the walk).  So the packed path runs the two-line form without branches -- a
    // compare and three selects per sample --, remembers the gate at fb, and commits the segment registers once, in
    // end_simd.  tripped() reports a triggering gate (the first segment end after every gate edge) and the caller
    // re-renders through step(); leave_fast() hands a half-planned block over to the walk (rollback of a later tile).
    FD_HD void plan_block(float input) {
        if (t >= t1) next_segment(input);
        start_chunk();
        speculate();
        fast = true;
        fb = loop_len < blk_size ? loop_len : -1;
        f_trip = fb >= 0 && !(s_valid && s_loop_len == blk_size - loop_len);
        f_cb = 0.0f;
    }
    FD_HD void leave_fast() {
        if (!fast) return;
        fast = false;
        if (fb >= 0 && blk_i >= fb) t += (float)(long long)loop_len * sd;  // the first chunk ended at sample fb - 1
        if (fb >= 0 && blk_i > fb) {  // ... and the prepared segment took over at fb (no trigger: tripped() was false then)
            t0 = t1; v0 = v1;
            t1 = s_t1; v1 = s_v1; t_hash = s_hash;
            loop_len = s_loop_len; full_seg = s_full; remaining = s_loop_len - (blk_i - fb);
            s_valid = false;
        } else {
            remaining = loop_len - blk_i;
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (PH == PH_SIMD) {
            if (blk_i == 0 && (blk_size & 7) == 0) plan_block(in[0].x);  // (blocks with remainder samples are walked)
            if (__builtin_expect(fast, 1)) {
                const int rel = fb - blk_i;
                const bool at0 = rel == 0, at1 = rel == 1;
                value = at0 ? s_value : value;
                value_d = at0 ? s_value_d : value_d;
                f_cb = at0 ? in[0].x : f_cb;
                const float o0 = value;
                value += value_d;
                value = at1 ? s_value : value;
                value_d = at1 ? s_value_d : value_d;
                f_cb = at1 ? in[0].y : f_cb;
                const float o1 = value;
                value += value_d;
                blk_i += 2;
                out[0] = v2f{o0, o1};
                return;
            }
        }
        float i0 = in[0].x, i1 = in[0].y, o0, o1;
        this->template step<PH>(&i0, &o0);
        this->template step<PH>(&i1, &o1);
        out[0] = v2f{o0, o1};
    }
};

// Envelope<f32, E, R>  envelope.rs:17-179 (ID 14): control-rate closure E(t) sampled at jittered ~2 ms intervals and
// linearly interpolated (envelope / lfo, prelude32.rs:581-611).  A Rust closure has no device form, so E is a FUNCTOR
// type FN: `static constexpr int OUT`, `visit(v)` for its own parameters (may be empty), `eval(float t, float* out)`.
// Two ship with the engine (the closures of the reference's doc examples); run-time compiled graphs can bring their
// own (fdsp_graph_compile_src: C++ source of the functor compiled with the graph).
struct EnvExp {  // lfo(|t| a * exp(-t * k))   (prelude32.rs:602 with a = k = 1)
    static constexpr int OUT = 1;
    float a, k;
    template <class V> FD_HD void visit(V& v) { v.f(a, PARAM, "a"); v.f(k, PARAM, "k"); }
    FD_HD void init() { a = 1.0f; k = 1.0f; }
    FD_HD void eval(float t, float* out) const { out[0] = a * expf_musl(-t * k); }
};
struct EnvSineHz {  // lfo(|t| lerp11(lo, hi, sin_hz(hz, t)))   (math.rs:204-206, 462-464)
    static constexpr int OUT = 1;
    float hz, lo, hi;
    template <class V> FD_HD void visit(V& v) { v.f(hz, PARAM, "hz"); v.f(lo, PARAM, "lo"); v.f(hi, PARAM, "hi"); }
    FD_HD void init() { hz = 1.0f; lo = -1.0f; hi = 1.0f; }
    FD_HD void eval(float t, float* out) const {
        const float u = sinf_musl(t * hz * F32_TAU) * 0.5f + 0.5f;
        out[0] = lo * (1.0f - u) + hi * u;
    }
};
// EnvelopeIn<f32, E, I, R>  envelope.rs:185-358 (ID 53): like Envelope, but the closure also sees the node's inputs
// E(t, &inputs) (envelope2 / lfo2 / envelope_in / lfo_in, prelude32.rs:625-745).  adsr_live above is the instance whose
// closure carries state; this is the general form for stateless functors: FN::IN inputs, FN::OUT outputs,
// `eval(float t, const float* in, float* out)`.
struct EnvInExp {  // lfo2(|t, speed| exp(-t * speed))   (prelude32.rs:623)
    static constexpr int IN = 1, OUT = 1;
    template <class V> FD_HD void visit(V&) {}
    FD_HD void init() {}
    FD_HD void eval(float t, const float* in, float* out) const { out[0] = expf_musl(-t * in[0]); }
};
template <class FN>
struct EnvelopeIn {
    static constexpr int IN = FN::IN, OUT = FN::OUT, RINGS = 0;
    static constexpr uint64_t ID = 53;
    FN fn;
    float interval, sd;
    float t, t0, t1, v0[OUT], v1[OUT], value[OUT], value_d[OUT];
    uint64_t t_hash, hash;
    int blk_i, blk_size, remaining, loop_len;  // block-walk transients (process path)
    bool full_seg;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); fn.visit(v); v.leave();
        v.f(interval, PARAM, "interval");
        v.f(sd, COEF, "sample_duration");
        v.f(t, STATE, "t"); v.f(t0, STATE, "t_0"); v.f(t1, STATE, "t_1");
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(v0[i], STATE, "value_0", i);
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(v1[i], STATE, "value_1", i);
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(value[i], STATE, "value", i);
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(value_d[i], STATE, "value_d", i);
        v.u64(t_hash, STATE, "t_hash");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() {  // EnvelopeIn::new :221-241 with interval 0.002
        fn.init();
        interval = (float)0.002;
        hash = 0;
        for (int i = 0; i < OUT; i++) { v0[i] = v1[i] = value[i] = value_d[i] = 0.0f; }
        reset();
    }
    FD_HD void update(double sr) { sd = (float)(1.0 / sr); }  // :300-302
    FD_HD void reset() { t = 0.0f; t0 = 0.0f; t1 = 0.0f; t_hash = hash; }  // :293-298
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :346-349: no reset
            hash = h;
            t_hash = h;
        }
        return atto(h, ID);
    }
    FD_HD void next_segment(const float* in) {  // :244-278
        if (t0 == 0.0f && t1 == 0.0f) {
            fn.eval(t0, in, v0);
        } else {
            t0 = t1;
            for (int i = 0; i < OUT; i++) v0[i] = v1[i];
        }
        const float next_interval = lerpf(0.75f, 1.25f, (float)rnd1(t_hash)) * interval;
        t1 = t0 + next_interval;
        fn.eval(t1, in, v1);
        t_hash = t_hash * 6364136223846793005ULL + 1ULL;
        const float u = (t - t0) / (t1 - t0);
        const float samples = next_interval / sd;
        for (int i = 0; i < OUT; i++) {
            value[i] = lerpf(v0[i], v1[i], u);
            value_d[i] = (v1[i] - v0[i]) / samples;
        }
    }
    FD_HD void begin_block(int size) { blk_i = 0; blk_size = size; remaining = 0; full_seg = false; loop_len = 0; }
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    FD_HD void start_chunk() {  // one iteration head of the `while i < size` loop :323-326
        float c = __builtin_ceilf((t1 - t) / sd);
        long long left = (long long)c;
        int room = blk_size - blk_i;
        bool huge = left < 0 || left > (long long)room;  // `as usize` of a negative value is huge
        loop_len = huge ? room : (int)left;
        full_seg = !huge && loop_len == (int)left;
        remaining = loop_len;
    }
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_TICK) {  // tick :305-313
            if (t >= t1) next_segment(in);
            for (int i = 0; i < OUT; i++) { out[i] = value[i]; value[i] += value_d[i]; }
            t += sd;
        } else {  // process :315-340, walked sample by sample (the whole block, no remainder path)
            if (blk_i == 0) {
                if (t >= t1) next_segment(in);
                start_chunk();
            }
            for (int guard = 0; remaining == 0 && guard < 4; guard++) {  // chunk exhausted before this sample (`i < size`)
                if (full_seg) next_segment(in);
                start_chunk();
            }
            for (int i = 0; i < OUT; i++) { out[i] = value[i]; value[i] += value_d[i]; }
            remaining--;
            blk_i++;
            if (remaining == 0) t += (float)(long long)loop_len * sd;
        }
    }
    FD_STEP2_VIA_STEP
};

template <class FN>
struct Envelope {
    static constexpr int IN = 0, OUT = FN::OUT, RINGS = 0;
    static constexpr uint64_t ID = 14;
    FN fn;
    float interval, sd;
    float t, t0, t1, v0[OUT], v1[OUT], value[OUT], value_d[OUT];
    uint64_t t_hash, hash;
    int blk_i, blk_size, remaining, loop_len;  // block-walk transients (process path)
    bool full_seg;
    // the next segment and the chunk after it, prepared at the head of the block (speculate; see AdsrLive)
    bool s_valid, s_cvalid, s_full;
    int s_loop_len;
    float s_t1, s_v1[OUT], s_value[OUT], s_value_d[OUT];
    uint64_t s_hash;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); fn.visit(v); v.leave();
        v.f(interval, PARAM, "interval");
        v.f(sd, COEF, "sample_duration");
        v.f(t, STATE, "t"); v.f(t0, STATE, "t_0"); v.f(t1, STATE, "t_1");
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(v0[i], STATE, "value_0", i);
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(v1[i], STATE, "value_1", i);
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(value[i], STATE, "value", i);
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(value_d[i], STATE, "value_d", i);
        v.u64(t_hash, STATE, "t_hash");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() {  // Envelope::new :58-76 with interval 0.002 (prelude32.rs:591)
        fn.init();
        interval = (float)0.002;
        hash = 0;
        for (int i = 0; i < OUT; i++) { v0[i] = v1[i] = value[i] = value_d[i] = 0.0f; }
        t = t0 = t1 = 0.0f;
        t_hash = 0;
    }
    // The reference constructs with the closure in place and calls reset() (which evaluates it) last; here the functor's
    // parameters arrive after construction, so every update() that precedes the first sample re-runs that reset.
    FD_HD void update(double sr) {  // :124-126
        sd = (float)(1.0 / sr);
        if (t == 0.0f && t1 == 0.0f) reset();
    }
    FD_HD void reset() {  // :114-122
        t = 0.0f; t0 = 0.0f; t1 = 0.0f;
        t_hash = hash;
        fn.eval(t0, v0);
        for (int i = 0; i < OUT; i++) v1[i] = v0[i];
    }
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :165-168: no reset
            hash = h;
            t_hash = h;
        }
        return atto(h, ID);
    }
    FD_HD void next_segment() {  // :79-98
        s_valid = false;  // whatever was prepared refers to the segment that ends here
        t0 = t1;
        for (int i = 0; i < OUT; i++) v0[i] = v1[i];
        const float next_interval = lerpf(0.75f, 1.25f, (float)rnd1(t_hash)) * interval;
        t1 = t0 + next_interval;
        fn.eval(t1, v1);
        t_hash = t_hash * 6364136223846793005ULL + 1ULL;
        const float u = (t - t0) / (t1 - t0);
        const float samples = next_interval / sd;
        for (int i = 0; i < OUT; i++) {
            value[i] = lerpf(v0[i], v1[i], u);
            value_d[i] = (v1[i] - v0[i]) / samples;
        }
    }
    FD_HD void begin_block(int size) { blk_i = 0; blk_size = size; remaining = 0; full_seg = false; loop_len = 0; s_valid = s_cvalid = false; }
    // The closure of an Envelope sees nothing but time, so a voice's next segment is fully known at the head of the block:
    // all lanes evaluate theirs together (one closure call per voice per block instead of one divergent call per lane per
    // boundary sample), with the operations next_segment / start_chunk perform, and take the registers at the boundary.
    FD_HD void speculate() {
        s_valid = s_cvalid = false;
        if (!full_seg) return;
        const float nt = t + (float)(long long)loop_len * sd;  // t at the boundary (the update at the end of step)
        const float next_interval = lerpf(0.75f, 1.25f, (float)rnd1(t_hash)) * interval;
        s_t1 = t1 + next_interval;
        fn.eval(s_t1, s_v1);
        s_hash = t_hash * 6364136223846793005ULL + 1ULL;
        const float u = (nt - t1) / (s_t1 - t1);
        const float samples = next_interval / sd;
        for (int i = 0; i < OUT; i++) {
            s_value[i] = lerpf(v1[i], s_v1[i], u);
            s_value_d[i] = (s_v1[i] - v1[i]) / samples;
        }
        const float c = __builtin_ceilf((s_t1 - nt) / sd);  // start_chunk at the sample after the boundary
        const long long left = (long long)c;
        const int room = blk_size - (blk_i + loop_len);
        const bool huge = left < 0 || left > (long long)room;
        s_loop_len = huge ? room : (int)left;
        s_full = !huge && s_loop_len == (int)left;
        s_valid = true;
    }
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    FD_HD void start_chunk() {  // one iteration head of the `while i < size` loop :137-140
        float c = __builtin_ceilf((t1 - t) / sd);
        long long left = (long long)c;
        int room = blk_size - blk_i;
        bool huge = left < 0 || left > (long long)room;  // `as usize` of a negative value is huge
        loop_len = huge ? room : (int)left;
        full_seg = !huge && loop_len == (int)left;
        remaining = loop_len;
    }
    template <int PH> FD_HD void step(const float*, float* out) {
        if (PH == PH_TICK) {  // tick :128-136
            if (t >= t1) next_segment();
            for (int i = 0; i < OUT; i++) { out[i] = value[i]; value[i] += value_d[i]; }
            t += sd;
        } else {  // process :138-163, walked sample by sample (the whole block, no remainder path)
            if (blk_i == 0) {
                if (t >= t1) next_segment();
                start_chunk();
                speculate();
            }
            for (int guard = 0; remaining == 0 && guard < 4; guard++) {  // chunk exhausted before this sample
                if (full_seg) next_segment();
                if (s_cvalid) {  // the chunk prepared with the segment that was just taken
                    loop_len = s_loop_len; full_seg = s_full; remaining = loop_len;
                } else {
                    start_chunk();
                }
                s_cvalid = false;
            }
            for (int i = 0; i < OUT; i++) { out[i] = value[i]; value[i] += value_d[i]; }
            remaining--;
            blk_i++;
            if (remaining == 0) {
                t += (float)(long long)loop_len * sd;
                if (full_seg) {  // :154-156: unconditional, also when the block ends here (EnvelopeIn differs: `i < size`)
                    if (s_valid) {  // the segment prepared at the head of the block
                        t0 = t1; t1 = s_t1; t_hash = s_hash;
                        for (int i = 0; i < OUT; i++) { v0[i] = v1[i]; v1[i] = s_v1[i]; value[i] = s_value[i]; value_d[i] = s_value_d[i]; }
                        s_valid = false;
                        s_cvalid = true;
                    } else {
                        next_segment();
                    }
                    full_seg = false;
                }
            }
        }
    }
    FD_STEP2_VIA_STEP
};


// Panner<N>  pan.rs:26-93 (ID 49): mono -> stereo, equal power.  N = 1: fixed pan; N = 2: pan on input 1, weights
// recomputed every sample in tick and in process alike (:55-61, :70-76).
template <int NIN>
struct PannerT {
    static constexpr int IN = NIN, OUT = 2, RINGS = 0;
    static constexpr uint64_t ID = 49;
    float pan, lw, rw;
    template <class V> FD_HD void visit(V& v) {
        v.f(pan, PARAM, "pan");
        v.f(lw, NIN > 1 ? STATE : COEF, "left_weight");
        v.f(rw, NIN > 1 ? STATE : COEF, "right_weight");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { pan = 0.0f; weights(0.0f); }
    FD_HD void weights(float value) {  // pan_weights :13-17
        float c = value > -1.0f ? value : -1.0f;
        c = c < 1.0f ? c : 1.0f;
        float angle = (c + 1.0f) * (F32_PI * 0.25f);
        lw = cosf_musl(angle);
        rw = sinf_musl(angle);
    }
    FD_HD void update(double) { if (NIN == 1) weights(pan); }
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :55-62 / :63-77
        if (NIN > 1) {
            weights(in[1]);
            out[0] = PH == PH_TICK ? lw * in[0] : in[0] * lw;
            out[1] = PH == PH_TICK ? rw * in[0] : in[0] * rw;
        } else {
            out[0] = lw * in[0];
            out[1] = rw * in[0];
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        if (NIN > 1) {
            float i0[2] = {in[0].x, in[1].x}, i1[2] = {in[0].y, in[1].y}, o0[2], o1[2];
            this->template step<PH>(i0, o0);
            this->template step<PH>(i1, o1);
            out[0] = v2f{o0[0], o1[0]};
            out[1] = v2f{o0[1], o1[1]};
        } else {
            out[0] = in[0] * lw;
            out[1] = in[0] * rw;
        }
    }
};
using Panner = PannerT<1>;

// ---------------------------------------------------------------------------------------------------------
// waveshapers (shape.rs:11-201) -- the shape KIND is a per-voice parameter (one kernel for all shapes)
// ---------------------------------------------------------------------------------------------------------
constexpr int SH_CLIP = 0, SH_CLIPTO = 1, SH_TANH = 2, SH_ATAN = 3, SH_SOFTSIGN = 4, SH_CRUSH = 5, SH_SOFTCRUSH = 6,
              SH_ADAPTIVE_TANH = 7,  // Adaptive<Tanh>, the form the reference's own graphs use
              SH_ADAPTIVE = 8;       // + inner kind: Adaptive<S> for any of the seven shapes (shape.rs:162-201)

FD_HD float smooth9f(float x) {  // math.rs:431-437
    float x2 = x * x;
    return ((((70.0f * x - 315.0f) * x + 540.0f) * x - 420.0f) * x + 126.0f) * x2 * x2 * x;
}
FD_HD float rs_clamp(float lo, float hi, float x) {  // math.rs:130-132: x.max(lo).min(hi)
    x = x > lo ? x : lo;  // f32::max / min (NaN-free inputs assumed equivalent)
    return x < hi ? x : hi;
}

struct Shape {
    float kind, p0, p1;        // params: Clip(p0) ClipTo(p0,p1) Tanh(p0) Atan(p0) Softsign(p0) Crush(p0) SoftCrush(p0); Adaptive<S>: S's own
    float smoothing, state;    // Adaptive only: smoothing is a host-computed coefficient (f64 pow, shape.rs:197-200)
    template <class V> FD_HD void visit(V& v) {
        v.f(kind, PARAM, "shape"); v.f(p0, PARAM, "shape_p0"); v.f(p1, PARAM, "shape_p1");
        v.f(smoothing, PARAM, "shape_smoothing");
        v.f(state, STATE, "shape_state");
    }
    FD_HD void init() { kind = (float)SH_TANH; p0 = 1.0f; p1 = 0.0f; smoothing = 0.0f; state = 0.0f; }  // Adaptive::new: state 0.0
    FD_HD void reset() { if ((int)kind >= SH_ADAPTIVE_TANH) state = 1.0e-3f; }                            // shape.rs:193-196
    // Shape::shape (scalar path)
    FD_HD float shape(float input) {
        int k = (int)kind;
        if (k >= SH_ADAPTIVE_TANH) {  // Adaptive<S> :185-192: level estimate, then the inner shape on input / sqrt(level)
            state = smoothing * state + (1.0f - smoothing) * (1.0e-6f + input * input);
            input = input / __builtin_sqrtf(state);
            k = k == SH_ADAPTIVE_TANH ? SH_TANH : k - SH_ADAPTIVE;
        }
        switch (k) {
        case SH_CLIP: return rs_clamp(-1.0f, 1.0f, input * p0);                                  // :48-51
        case SH_CLIPTO: return rs_clamp(p0, p1, input);                                           // :64-67
        case SH_TANH: return tanhf_musl(input * p0);                                              // :82-85
        case SH_ATAN: return atanf_musl(input * (p0 * F32_PI * 0.5f)) * (2.0f / F32_PI);          // :93-96
        case SH_SOFTSIGN: { float x = input * p0; return x / (1.0f + __builtin_fabsf(x)); }      // :109-112, math.rs:386
        case SH_CRUSH: return roundf_musl(input * p0) / p0;                                       // :125-128
        default: { float x = input * p0; float y = __builtin_floorf(x); return (y + smooth9f(x - y)) / p0; }  // SoftCrush :141-146
        }
    }
    // Shape::simd, one lane (the f32x8 path of Shaper::process)
    FD_HD float simd(float input) {
        switch ((int)kind) {
        case SH_ATAN: return wide_atanf(input * (p0 * F32_PI * 0.5f)) * (2.0f / F32_PI);          // :98-100
        case SH_SOFTSIGN: return input * p0 / (1.0f + __builtin_fabsf(input) * p0);               // :114-116
        case SH_CRUSH: return __builtin_rintf(input * p0) / p0;                                   // :130-132 (wide round: half to even)
        case SH_SOFTCRUSH: {                                                                      // :148-152, vector floor lib.rs:326-328
            float x = input * p0;
            float y = __builtin_rintf(x - 0.4999999f);
            return (y + smooth9f(x - y)) / p0;
        }
        default: return shape(input);  // Clip / ClipTo: fast_max/fast_min == clamp for non-NaN; Tanh, Adaptive<S>: per-lane shape()
        }
    }
};

// Shaper<S>  shape.rs:205-247 (ID 42)
struct Shaper {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 42;
    Shape sh;
    template <class V> FD_HD void visit(V& v) { sh.visit(v); }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { sh.init(); }
    FD_HD void update(double) {}
    FD_HD void reset() { sh.reset(); }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        out[0] = PH == PH_SIMD ? sh.simd(in[0]) : sh.shape(in[0]);
    }
    FD_STEP2_VIA_STEP
};

// ---------------------------------------------------------------------------------------------------------
// phase oscillators without a process override: Ramp, PolySaw, PolySquare, PolyPulse (oscillator.rs:441-760)
// ---------------------------------------------------------------------------------------------------------
FD_HD float polyblep(float t, float dt) {  // oscillator.rs:512-523
    if (t < dt) {
        float z = t / dt;
        return z + z - z * z - 1.0f;
    } else if (t > 1.0f - dt) {
        float z = (t - 1.0f) / dt;
        return z + z + z * z + 1.0f;
    }
    return 0.0f;
}

constexpr int OSC_RAMP = 0, OSC_POLYSAW = 1, OSC_POLYSQUARE = 2, OSC_POLYPULSE = 3;
template <int KIND>
struct PhaseOsc {
    static constexpr int IN = KIND == OSC_POLYPULSE ? 2 : 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = KIND == OSC_RAMP ? 94 : KIND == OSC_POLYSAW ? 95 : KIND == OSC_POLYSQUARE ? 96 : 97;
    float phase, sample_duration, has_phase, initial_phase;
    uint64_t hash;
    template <class V> FD_HD void visit(V& v) {
        v.f(phase, STATE, "phase");
        v.f(sample_duration, COEF, "sample_duration");
        v.f(has_phase, PARAM, "has_initial_phase");
        v.f(initial_phase, PARAM, "initial_phase");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { has_phase = 0.0f; initial_phase = 0.0f; hash = 0; reset(); }
    FD_HD void update(double sr) { sample_duration = (float)(1.0 / sr); }
    FD_HD void reset() { phase = has_phase != 0.0f ? initial_phase : (float)rnd1(hash); }
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) { hash = h; reset(); }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        const float ph = phase;
        const float delta = in[0] * sample_duration;
        phase += delta;
        phase -= __builtin_floorf(phase);
        if (KIND == OSC_RAMP) {  // :478-483
            out[0] = ph;
        } else if (KIND == OSC_POLYSAW) {  // :570-577
            out[0] = 2.0f * ph - 1.0f - polyblep(ph, delta);
        } else {  // PolySquare :646-659 (width 0.5) / PolyPulse :729-739 (width = input 1)
            const float width = KIND == OSC_POLYPULSE ? in[IN - 1] : 0.5f;
            const float square = ph < width ? 1.0f : -1.0f;
            const float half = ph - width;
            out[0] = square + polyblep(ph, delta) - polyblep(half - __builtin_floorf(half), delta);
        }
    }
    FD_STEP2_VIA_STEP
};

// Rossler (oscillator.rs:323-375, ID 73) and Lorenz (:382-435, ID 74) attractors, explicit Euler
template <bool LORENZ>
struct Chaos {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = LORENZ ? 74 : 73;
    float x, y, z, sr;
    uint64_t hash;
    template <class V> FD_HD void visit(V& v) {
        v.f(x, STATE, "x"); v.f(y, STATE, "y"); v.f(z, STATE, "z");
        v.f(sr, COEF, "sample_rate");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { hash = 0; reset(); }
    FD_HD void update(double sample_rate) { sr = (float)sample_rate; }
    FD_HD void reset() {  // lerp(0.0, 1.0, rnd1(hash) as f32)
        float t = (float)rnd1(hash);
        x = 0.0f * (1.0f - t) + 1.0f * t;
        y = 1.0f;
        z = 1.0f;
    }
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) { hash = h; reset(); }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (LORENZ) {  // :407-417
            float dx = 10.0f * (y - x);
            float dy = x * (28.0f - z) - y;
            float dz = x * y - (8.0f / 3.0f) * z;
            float dt = in[0] / sr;
            x += dx * dt; y += dy * dt; z += dz * dt;
            out[0] = x * 0.05107f;
        } else {  // :348-358
            float dx = -y - z;
            float dy = x + 0.15f * y;
            float dz = 0.2f + z * (x - 10.0f);
            float dt = 2.91f * in[0] / sr;
            x += dx * dt; y += dy * dt; z += dz * dt;
            out[0] = x * 0.05757f;
        }
    }
    FD_STEP2_VIA_STEP
};

// ---------------------------------------------------------------------------------------------------------
// biquads with nonlinear feedback / state shaping (biquad.rs:494-920): FbBiquad (ID 88), FixedFbBiquad (ID 90),
// DirtyBiquad (ID 89), FixedDirtyBiquad (ID 91).  Mode (resonator/lowpass/highpass/bell) and shape are per voice.
// NIN = 1: fixed; NIN = 3: (audio, center, q); NIN = 4: (audio, center, q, gain).
// ---------------------------------------------------------------------------------------------------------
template <bool DIRTY, int NIN>
struct NlBiquad {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = DIRTY ? (NIN == 1 ? 91 : 89) : (NIN == 1 ? 90 : 88);
    float mode, center, q, gain, sr;
    float a1, a2, b0, b1, b2, s1, s2;
    Shape sh1, sh2;
    template <class V> FD_HD void visit(V& v) {
        constexpr FieldKind PK = NIN > 1 ? STATE : PARAM, CK = NIN > 1 ? STATE : COEF;
        v.f(mode, PARAM, "mode");
        v.f(center, PK, "center"); v.f(q, PK, "q"); v.f(gain, PK, "gain");
        v.f(sr, COEF, "sample_rate");
        v.f(a1, CK, "a1"); v.f(a2, CK, "a2"); v.f(b0, CK, "b0"); v.f(b1, CK, "b1"); v.f(b2, CK, "b2");
        v.f(s1, STATE, "s1"); v.f(s2, STATE, "s2");
        v.enter(0); sh1.visit(v); v.leave();
        if (DIRTY) { v.enter(1); sh2.visit(v); v.leave(); }
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void coefs() {  // BiquadMode::update :404-490 (kinds: BQ_RESONATOR/LOWPASS/HIGHPASS/BELL)
        BiquadCoefs c = biquad_coefs((int)mode, sr, center, q, gain);
        a1 = c.a1; a2 = c.a2; b0 = c.b0; b1 = c.b1; b2 = c.b2;
    }
    FD_HD void init() {  // ::new :505-521: center 440, q 1, gain 1
        mode = (float)BQ_LOWPASS; center = 440.0f; q = 1.0f; gain = 1.0f; s1 = 0.0f; s2 = 0.0f;
        sh1.init(); sh2.init();
    }
    FD_HD void update(double sample_rate) { sr = (float)sample_rate; coefs(); }
    FD_HD void reset() { s1 = 0.0f; s2 = 0.0f; sh1.reset(); if (DIRTY) sh2.reset(); }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (NIN == 3) {  // :549-557: squared-difference change detection
            float dc = in[1] - center, dq = in[2] - q;
            if (dc * dc + dq * dq != 0.0f) { center = in[1]; q = in[2]; coefs(); }
        }
        if (NIN == 4) {  // :558-572
            float dc = in[1] - center, dq = in[2] - q, dg = in[3] - gain;
            if (dc * dc + dq * dq + dg * dg != 0.0f) { center = in[1]; q = in[2]; gain = in[3]; coefs(); }
        }
        const float x0 = in[0];
        const float y0 = b0 * x0 + s1;
        if (DIRTY) {  // :789-796
            const float n1 = sh1.shape(s2 + b1 * x0 - y0 * a1);
            const float n2 = sh2.shape(b2 * x0 - y0 * a2);
            s1 = n1;
            s2 = n2;
        } else {  // :577-581
            const float fb = sh1.shape(y0);
            s1 = s2 + b1 * x0 - fb * a1;
            s2 = b2 * x0 - fb * a2;
        }
        out[0] = y0;
    }
    FD_STEP2_VIA_STEP
};

// ---------------------------------------------------------------------------------------------------------
// delay lines (delay.rs): Delay, Tap<U1>, TapLinear<U1>, AllNest<U1, X>.  Ring memory: see Ctx.
// ---------------------------------------------------------------------------------------------------------
FD_HD float splinef(float y0, float y1, float y2, float y3, float x) {  // Catmull-Rom, math.rs:360-366
    return y1 + x * 0.5f * (y2 - y0 + x * (2.0f * y0 - 5.0f * y1 + 4.0f * y2 - y3 + x * (3.0f * (y1 - y2) + y3 - y0)));
}
FD_HD uint32_t next_pow2_u32(uint32_t x) {  // usize::next_power_of_two
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// Delay  delay.rs:72-139 (ID 13): fixed delay of round(time * sr) samples; ring length = that + 1.
struct Delay {
    static constexpr int IN = 1, OUT = 1, RINGS = 1;
    static constexpr uint64_t ID = 13;
    float time;           // seconds (prelude delay(t: f32) -> Delay::new(t as f64))
    float last_sr;        // the rate the ring was sized for: a CHANGE resizes and resets (delay.rs:105-113)
    uint32_t len, i;
    float* ring;
    size_t vs;
    uint32_t cap;
    Ctx owner;  // for want_positions()
    template <class V> FD_HD void visit(V& v) {
        v.f(time, PARAM, "time");
        v.f(last_sr, COEF, "sized_for_sample_rate");
        v.u32(len, COEF, "length");
        v.u32(i, STATE, "i");
    }
    FD_HD void bind(Ctx& c) { ring = c.claim_ring(); vs = c.vstride; cap = c.ring_cap; owner = c; }
    FD_HD void clear() {
        for (uint32_t k = 0; k < cap; k++) ring[(size_t)k * vs] = 0.0f;
    }
    FD_HD void init() { time = 0.0f; last_sr = 0.0f; len = 1; i = 0; }
    FD_HD void update(double sr) {
        const double wd = __builtin_round((double)time * sr) + 1.0;
        uint32_t want = wd < 4294967295.0 ? (wd > 1.0 ? (uint32_t)wd : 1u) : 0xFFFFFFFFu;
        owner.want_positions(want);      // capacity is fixed at bank creation (fdsp_bank_create_ring): the host fails the
        want = want > cap ? cap : want;  // call that asked for more; until then the delay is the longest that fits
        if (last_sr != (float)sr || want != len) {
            last_sr = (float)sr;
            len = want;
            reset();
        }
    }
    FD_HD void reset() { i = 0; clear(); }  // :100-103
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :116-124
        ring[(size_t)i * vs] = in[0];
        i += 1;
        if (i >= len) i = 0;
        out[0] = ring[(size_t)i * vs];
    }
    FD_STEP2_VIA_STEP
};

// Pluck  oscillator.rs:210-317 (ID 58): Karplus-Strong string = delay line -> Fir<U3> damping -> Allpole tuning, fed back.
// Two ring nodes: ring 0 holds the EXCITATION, i.e. the stream `Rnd::from_u64(hash).f32_in(-1.0, 1.0)` that
// initialize_line draws (:257-261).  funutd's generator is a third-party crate whose source is not under
// /root/reference, so the host uploads that stream (fdsp_bank_set_ring, ring 0) -- the Rust shim calls funutd itself;
// everything after the draw (mean removal in f64, loop gain, damping, tuning, the loop) is reproduced here.  Ring 1
// is the working line.  `gain` uses the device library's f64 pow (policy of halfway_coeff: f64 result rounded to f32).
struct Pluck {
    static constexpr int IN = 1, OUT = 1, RINGS = 2;
    static constexpr uint64_t ID = 58;
    float frequency, gain_per_second, damping;   // params (constructor arguments)
    float gain, w[3], eta;                        // derived
    float fv[3], ax1, ay1;                        // Fir / Allpole state
    float initialized, last_freq;
    uint32_t len, pos;
    uint64_t sr_bits, hash;                       // sample_rate is f64 in the reference (:224)
    float *raw, *line;
    size_t vs;
    uint32_t cap;
    template <class V> FD_HD void visit(V& v) {
        v.f(frequency, PARAM, "frequency");
        v.f(gain_per_second, PARAM, "gain_per_second");
        v.f(damping, PARAM, "high_frequency_damping");
        v.f(gain, COEF, "gain");
        _Pragma("unroll") for (int i = 0; i < 3; i++) v.fi(w[i], COEF, "w", i);
        v.f(eta, STATE, "eta");
        _Pragma("unroll") for (int i = 0; i < 3; i++) v.fi(fv[i], STATE, "v", i);
        v.f(ax1, STATE, "x1");
        v.f(ay1, STATE, "y1");
        v.f(initialized, STATE, "initialized");
        v.f(last_freq, STATE, "initialized_for_frequency");
        v.u32(len, STATE, "length");
        v.u32(pos, STATE, "pos");
        v.u64(sr_bits, COEF, "sample_rate");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx& c) { raw = c.claim_ring(); line = c.claim_ring(); vs = c.vstride; cap = c.ring_cap; }
    FD_HD static double bits2d(uint64_t b) { return __builtin_bit_cast(double, b); }
    FD_HD void init() {
        frequency = 440.0f; gain_per_second = 0.5f; damping = 0.5f;
        fv[0] = fv[1] = fv[2] = 0.0f; ax1 = ay1 = 0.0f; eta = 0.0f;
        initialized = 0.0f; last_freq = 0.0f; len = 1; pos = 0; hash = 0;
        sr_bits = __builtin_bit_cast(uint64_t, 44100.0);
    }
    FD_HD void update(double sr) {  // Pluck::new :229-242 (parameter-derived parts) + set_sample_rate :279-285
        const float g = 1.0f - damping;  // fir3(1.0 - high_frequency_damping)  prelude.rs:863-867
        const float alpha = (g + 1.0f) / 2.0f, beta = (1.0f - alpha) / 2.0f;
        w[0] = beta; w[1] = alpha; w[2] = beta;
        gain = (float)pow_f64((double)gain_per_second, 1.0 / (double)frequency);
        if (bits2d(sr_bits) != sr || last_freq != frequency) {
            sr_bits = __builtin_bit_cast(uint64_t, sr);
            initialized = 0.0f;
        }
    }
    FD_HD void reset() { fv[0] = fv[1] = fv[2] = 0.0f; initialized = 0.0f; }  // :274-277
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :307-310
            hash = h;
            initialized = 0.0f;
        }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    FD_HD void initialize_line() {  // :244-268
        const double total_delay = bits2d(sr_bits) / (double)frequency - 1.0;
        const double loop_delay = __builtin_floor(total_delay - 0.2);
        const double allpass_delay = total_delay - loop_delay;
        ax1 = ay1 = 0.0f;                           // tuning.reset()
        const float d = (float)allpass_delay;       // tuning.set_delay: eta = (1 - d) / (1 + d)  filter.rs:292-295
        eta = (1.0f - d) / (1.0f + d);
        uint32_t n = loop_delay > 0.0 ? (uint32_t)loop_delay : 0u;
        n = n > cap ? cap : n;                      // ring capacity is fixed at bank creation
        n = n < 1u ? 1u : n;
        len = n;
        double mean = 0.0;
        for (uint32_t i = 0; i < n; i++) {
            const float x = raw[(size_t)i * vs];
            line[(size_t)i * vs] = x;
            mean += (double)x;
        }
        mean /= (double)n;
        for (uint32_t i = 0; i < n; i++) line[(size_t)i * vs] -= (float)mean;
        pos = 0;
        last_freq = frequency;
        initialized = 1.0f;
    }
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :287-305
        if (initialized == 0.0f) initialize_line();
        float o = line[(size_t)pos * vs] * gain + in[0];
        fv[0] = fv[1]; fv[1] = fv[2]; fv[2] = o;    // damping.filter_mono: Fir<U3>::tick fir.rs:57-70
        float acc = 0.0f;
        acc += w[0] * fv[0];
        acc += w[1] * fv[1];
        acc += w[2] * fv[2];
        const float y0 = eta * (acc - ay1) + ax1;   // tuning.filter_mono: Allpole::tick filter.rs:320-324
        ax1 = acc; ay1 = y0;
        line[(size_t)pos * vs] = y0;
        pos += 1;
        if (pos == len) pos = 0;
        out[0] = y0;
    }
    FD_STEP2_VIA_STEP
};

// Tap<U1> (cubic, ID 50, delay.rs:148-286) and TapLinear<U1> (ID 54, :386-505): variable fractional delay in seconds
// on input 1.  The f32x8 `process` writes 8 samples and then reads with per-lane offsets; because the clamped delay
// is at least one sample (Tap) / the read never runs ahead of the write (TapLinear), each lane reads exactly what
// `tick` reads, so one sample-serial implementation serves both paths.
template <bool LINEAR, int NT = 1>  // NT taps: multitap / multitap_linear (inputs 1..NT = delays, output = their sum)
struct TapT {
    static constexpr int IN = 1 + NT, OUT = 1, RINGS = 1;
    static constexpr uint64_t ID = LINEAR ? 54 : 50;
    float min_delay, max_delay;                    // params
    float srf, min_c, max_c;                       // coefs
    uint32_t mask, i;
    float* ring;
    size_t vs;
    uint32_t cap;
    Ctx owner;
    template <class V> FD_HD void visit(V& v) {
        v.f(min_delay, PARAM, "min_delay"); v.f(max_delay, PARAM, "max_delay");
        v.f(srf, COEF, "sample_rate"); v.f(min_c, COEF, "min_delay_clamped"); v.f(max_c, COEF, "max_delay_clamped");
        v.u32(mask, COEF, "mask");
        v.u32(i, STATE, "i");
    }
    FD_HD void bind(Ctx& c) { ring = c.claim_ring(); vs = c.vstride; cap = c.ring_cap; owner = c; }
    FD_HD void init() { min_delay = 0.0f; max_delay = 0.0f; srf = 0.0f; mask = 0; i = 0; min_c = max_c = 0.0f; }
    FD_HD void update(double sample_rate) {  // :198-209 / :436-445
        float sr = (float)sample_rate;
        float blen = LINEAR ? __builtin_ceilf(max_delay * sr) + 2.0f : __builtin_ceilf(max_delay * sr) + 3.0f + 8.0f;
        uint32_t n = next_pow2_u32((uint32_t)blen);
        owner.want_positions(n);  // capacity fixed at bank creation: the host fails the call that asked for more
        while (n > cap) n >>= 1;
        uint32_t m = n - 1u;
        if (srf != sr || m != mask) {
            srf = sr;
            mask = m;
            min_c = LINEAR ? min_delay : (min_delay > 1.00001f / sr ? min_delay : 1.00001f / sr);
            max_c = LINEAR ? max_delay : (max_delay > 1.00001f / sr ? max_delay : 1.00001f / sr);
            reset();
        }
    }
    FD_HD void reset() {  // :193-196
        i = 0;
        for (uint32_t k = 0; k <= mask && k < cap; k++) ring[(size_t)k * vs] = 0.0f;
    }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :212-236 / :448-463
        ring[(size_t)i * vs] = in[0];
        float o = 0.0f;
        _Pragma("unroll") for (int k = 1; k <= NT; k++) {
            float tap = rs_clamp(min_c, max_c, in[k]) * srf;
            uint32_t tap_floor = (uint32_t)tap;
            uint32_t i1 = (i - tap_floor) & mask;
            float d = tap - (float)tap_floor;
            if (LINEAR) {
                uint32_t i2 = (i1 - 1u) & mask;
                float a = ring[(size_t)i1 * vs], b = ring[(size_t)i2 * vs];
                o += a * (1.0f - d) + b * d;  // lerp math.rs:169-178
            } else {
                uint32_t i0 = (i1 + 1u) & mask, i2 = (i1 - 1u) & mask, i3 = (i1 - 2u) & mask;
                o += splinef(ring[(size_t)i0 * vs], ring[(size_t)i1 * vs], ring[(size_t)i2 * vs], ring[(size_t)i3 * vs], d);
            }
        }
        i = (i + 1u) & mask;
        out[0] = o;
    }
    FD_STEP2_VIA_STEP
};

// AllNest<N, X>  delay.rs:294-377 (ID 83): Schroeder allpass around the single-channel node X; N = 1: fixed
// coefficient (allnest_c), N = 2: coefficient on input 1 (allnest).
template <class X, int NIN = 1>
struct AllNest {
    static_assert(X::IN == 1 && X::OUT == 1, "AllNest wraps a 1-in 1-out node");
    static constexpr int IN = NIN, OUT = 1, RINGS = X::RINGS;
    static constexpr uint64_t ID = 83;
    X x;
    float eta, z;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        v.f(eta, NIN > 1 ? STATE : PARAM, "coefficient");
        v.f(z, STATE, "z");
    }
    FD_HD void bind(Ctx& c) { x.bind(c); }
    FD_HD void init() { x.init(); eta = 0.0f; z = 0.0f; }
    FD_HD void update(double sr) { x.update(sr); }
    FD_HD void reset() { z = 0.0f; x.reset(); }  // :316-319
    FD_HD uint64_t ping(bool probe, uint64_t h) { return x.ping(probe, atto(h, ID)); }  // :337-339
    FD_HD void begin_block(int n) { x.begin_block(n); }
    FD_HD bool tripped() const { return x.tripped(); }
    FD_HD void end_simd() { x.end_simd(); }
    template <int PH> FD_HD void step(const float* in, float* out) {  // :322-330 (tick everywhere: no process override)
        if (NIN > 1) eta = in[1];
        float v = in[0] - eta * z;
        float y = eta * v + z;
        float zi;
        x.template step<PH_TICK>(&v, &zi);
        z = zi;
        out[0] = y;
    }
    FD_STEP2_VIA_STEP
};

// Reverb<F>  reverb.rs:152-279 (ID 85), reverb3_stereo(time, diffusion, filter): an allpass loop.  Per voice: 4 input
// diffusers + 8 blocks of (Delay, 4 Schroeder allpasses, filter, 4 Schroeder allpasses, filter) = 76 delay rings
// (+ the filters' own), all walked serially every sample; the loop's last value feeds the next sample's first block.
// Quirks kept: `pre` is neither reset by reset() (:211-224) nor re-sized by set_sample_rate() (:226-238) -- its delays
// stay at (n - 1) samples whatever the rate; no process override (tick arithmetic everywhere); the enclosed filters are
// never pinged (default leaf ping :156-161).  `a` = pow(db_amp(-60), 0.035 / time) as f32 and `coefficient` =
// lerp(0.5, 0.9, diffusion) as f32 are computed by the host in f64 like Reverb::new (:162-207) and arrive as parameters.
template <class F>
struct Reverb3 {
    static_assert(F::IN == 1 && F::OUT == 1, "reverb3_stereo: the loop filter is a 1-in 1-out node");
    static constexpr int IN = 2, OUT = 2, RINGS = 4 + 8 * 9 + 16 * F::RINGS;
    static constexpr uint64_t ID = 85;
    using Schroeder = AllNest<Delay>;
    struct Block {
        Schroeder ap0[4], ap1[4];
        F f0, f1;
        Delay delay;
    };
    Schroeder pre[4];
    Block block[8];
    float feedback, a, coeff;
    static FD_HD int ldelay(int k) {
        constexpr int T[32] = {401, 421, 443, 463, 487, 503, 523, 547, 563, 587, 607, 619, 643, 661, 683, 701,
                               727, 743, 761, 787, 809, 823, 839, 863, 883, 907, 929, 947, 967, 983, 1009, 1021};
        return T[k];
    }
    static FD_HD int rdelay(int k) {
        constexpr int T[32] = {419, 433, 457, 479, 491, 509, 541, 557, 577, 593, 613, 631, 653, 673, 691, 719,
                               733, 757, 773, 797, 811, 829, 853, 877, 887, 911, 937, 953, 977, 997, 1013, 1033};
        return T[k];
    }
    static FD_HD int bdelay(int k) {
        constexpr int T[8] = {1087, 1091, 1093, 1097, 1103, 1109, 1117, 1123};
        return T[k];
    }
    static FD_HD int pdelay(int k) {
        constexpr int T[4] = {245, 367, 263, 349};
        return T[k];
    }
    template <class V> FD_HD void visit(V& v) {
        v.f(a, PARAM, "a");
        v.f(coeff, PARAM, "coefficient");
        v.f(feedback, STATE, "feedback");
        int k = 0;
        for (int i = 0; i < 4; i++) { v.enter(k++); pre[i].visit(v); v.leave(); }
        for (int b = 0; b < 8; b++) {
            for (int j = 0; j < 4; j++) { v.enter(k++); block[b].ap0[j].visit(v); v.leave(); }
            for (int j = 0; j < 4; j++) { v.enter(k++); block[b].ap1[j].visit(v); v.leave(); }
            v.enter(k++); block[b].f0.visit(v); v.leave();
            v.enter(k++); block[b].f1.visit(v); v.leave();
            v.enter(k++); block[b].delay.visit(v); v.leave();
        }
    }
    FD_HD void bind(Ctx& c) {
        for (int i = 0; i < 4; i++) pre[i].bind(c);
        for (int b = 0; b < 8; b++) {
            for (int j = 0; j < 4; j++) block[b].ap0[j].bind(c);
            for (int j = 0; j < 4; j++) block[b].ap1[j].bind(c);
            block[b].f0.bind(c);
            block[b].f1.bind(c);
            block[b].delay.bind(c);
        }
    }
    FD_HD void init() {
        a = 0.0f; coeff = 0.5f; feedback = 0.0f;
        for (int i = 0; i < 4; i++) pre[i].init();
        for (int b = 0; b < 8; b++) {
            for (int j = 0; j < 4; j++) { block[b].ap0[j].init(); block[b].ap1[j].init(); }
            block[b].f0.init();
            block[b].f1.init();
            block[b].delay.init();
        }
    }
    // Delay::new(time) + set_sample_rate(sr) with the f64 time of Reverb::new: length round(time * sr) + 1, a change
    // of rate or length clears the line (delay.rs:105-113)
    static FD_HD void size_delay(Delay& d, int samples_at_default_sr, double sr) {
        const double time = (double)samples_at_default_sr / 44100.0;
        uint32_t want = (uint32_t)__builtin_round(time * sr) + 1u;
        want = want > d.cap ? d.cap : want;
        if (d.last_sr != (float)sr || want != d.len) {
            d.last_sr = (float)sr;
            d.len = want;
            d.reset();
        }
    }
    FD_HD void update(double sr) {
        for (int i = 0; i < 4; i++) {
            pre[i].eta = coeff;
            size_delay(pre[i].x, pdelay(i) - 1, 44100.0);  // never re-sized (:226-238)
        }
        for (int b = 0; b < 8; b++) {
            for (int j = 0; j < 4; j++) {
                block[b].ap0[j].eta = coeff;
                block[b].ap1[j].eta = coeff;
                size_delay(block[b].ap0[j].x, ldelay(b + j * 8) - 1, sr);
                size_delay(block[b].ap1[j].x, rdelay(b + j * 8) - 1, sr);
            }
            block[b].f0.update(sr);
            block[b].f1.update(sr);
            size_delay(block[b].delay, bdelay(7 - b), sr);
        }
    }
    FD_HD void reset() {  // :211-224: not `pre`
        for (int b = 0; b < 8; b++) {
            for (int j = 0; j < 4; j++) { block[b].ap0[j].reset(); block[b].ap1[j].reset(); }
            block[b].f0.reset();
            block[b].f1.reset();
            block[b].delay.reset();
        }
        feedback = 0.0f;
    }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <class N> static FD_HD float mono(N& n, float x) {  // filter_mono
        float y;
        n.template step<PH_TICK>(&x, &y);
        return y;
    }
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :241-272
        float v0 = feedback, o0 = 0.0f, o1 = 0.0f;
        float input0 = mono(pre[0], in[0] * 0.5f);
        input0 = mono(pre[1], input0);
        float input1 = mono(pre[2], in[1] * 0.5f);
        input1 = mono(pre[3], input1);
        for (int b = 0; b < 8; b++) {
            Block& k = block[b];
            v0 = mono(k.delay, v0);
            v0 = mono(k.ap0[0], a * v0 + input0);
            v0 = mono(k.ap0[1], v0);
            v0 = mono(k.ap0[2], v0);
            v0 = mono(k.ap0[3], v0);
            v0 = mono(k.f0, v0);
            o0 = v0;
            v0 = mono(k.ap1[0], a * v0 + input1);
            v0 = mono(k.ap1[1], v0);
            v0 = mono(k.ap1[2], v0);
            v0 = mono(k.ap1[3], v0);
            v0 = mono(k.f1, v0);
            o1 = v0;
        }
        feedback = v0;
        out[0] = o0;
        out[1] = o1;
    }
    FD_STEP2_VIA_STEP
};

// Oversampler<X>  oversample.rs:66-249 (ID 51): X runs at twice the sample rate between a minimum-phase half-band
// interpolator and decimator (HALFBAND_MIN :329-373).  The reference keeps 128-sample rings per channel; only the newest
// 24 input and 48 inner-output samples ever reach a filter (START_SAMPLE_OFFSET_HALF / _FULL :523-524), so the device
// state is those histories in age order (oldest first) -- the f32x8 lane a sample lands in depends only on its age.
// wide's mul_add / reduce_add as in the oracle (unfused; low f32x4 sum + high f32x4 sum).
// Process semantics (:178-212): each half of the outer block feeds ONE inner block of `size` samples through
// X::process, so the inner node sees begin_block / SIMD items / end_simd / remainder per half; an odd `size` leaves the
// last outer sample unwritten in the reference (0.0 here, state untouched).  The reference's decimation loop runs
// over Inputs::USIZE channels (:200) -- a generator is never written at all; this node writes every output channel.
FD_HD float halfband_min(int i) {
    constexpr float H[43] = {
        4.73552339e-02f, 1.81988040e-01f, 3.49148434e-01f, 3.92748135e-01f, 2.18230867e-01f, -5.31842843e-02f,
        -1.79186566e-01f, -7.34488007e-02f, 8.94524103e-02f, 1.00868556e-01f, -2.08681451e-02f, -8.82510989e-02f,
        -2.07640777e-02f, 6.22587555e-02f, 4.07776255e-02f, -3.52258090e-02f, -4.57407870e-02f, 1.27033444e-02f,
        4.14376136e-02f, 3.30799834e-03f, -3.24608206e-02f, -1.27856355e-02f, 2.21659033e-02f, 1.67803711e-02f,
        -1.27406974e-02f, -1.68177367e-02f, 5.35518220e-03f, 1.44761581e-02f, -3.70651781e-04f, -1.11140183e-02f,
        -2.40622311e-03f, 7.71596027e-03f, 3.48227062e-03f, -4.86763558e-03f, -3.45536353e-03f, 2.79880054e-03f,
        2.86736431e-03f, -1.48746153e-03f, -2.11827989e-03f, 7.72684113e-04f, 1.44384114e-03f, -4.49807048e-04f,
        -9.41945265e-04f};
    return H[i];
}
FD_HD float wide_reduce_add8(const float* a) {
    float lo = 0.0f, hi = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) lo += a[j];
#pragma unroll
    for (int j = 4; j < 8; j++) hi += a[j];
    return lo + hi;
}
template <class X>
struct Oversampler {
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = X::RINGS;
    static constexpr uint64_t ID = 51;
    X x;
    float hin[IN > 0 ? IN : 1][24];  // newest input samples, oldest first
    float hout[OUT][48];             // newest inner output samples, oldest first
    int blk_size, blk_i;             // transients of the process walk
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
#pragma unroll
        for (int c = 0; c < IN; c++)
#pragma unroll
            for (int k = 0; k < 24; k++) v.fi(hin[c][k], STATE, "inv", c * 24 + k);
#pragma unroll
        for (int c = 0; c < OUT; c++)
#pragma unroll
            for (int k = 0; k < 48; k++) v.fi(hout[c][k], STATE, "outv", c * 48 + k);
    }
    FD_HD void bind(Ctx& c) { x.bind(c); }
    FD_HD void clear() {
#pragma unroll
        for (int c = 0; c < IN; c++)
#pragma unroll
            for (int k = 0; k < 24; k++) hin[c][k] = 0.0f;
#pragma unroll
        for (int c = 0; c < OUT; c++)
#pragma unroll
            for (int k = 0; k < 48; k++) hout[c][k] = 0.0f;
    }
    FD_HD void init() { x.init(); clear(); }
    FD_HD void update(double sr) { x.update(sr * 2.0); }                                // :137-140
    FD_HD void reset() { x.reset(); clear(); }                                          // :131-135
    FD_HD uint64_t ping(bool probe, uint64_t h) { return x.ping(probe, atto(h, ID)); }  // :218-220
    FD_HD void begin_block(int n) { blk_size = n; blk_i = 0; }
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    // push one input sample per channel, return the two inner input frames (interpolating_filter :11-41)
    FD_HD void interpolate(const float* in, float* even, float* odd) {
#pragma unroll
        for (int c = 0; c < IN; c++) {
#pragma unroll
            for (int k = 0; k < 23; k++) hin[c][k] = hin[c][k + 1];
            hin[c][23] = in[c];
            float ae[8], ao[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { ae[j] = 0.0f; ao[j] = 0.0f; }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = i * 8 + j;
                    const float ce = k < 2 ? 0.0f : halfband_min(2 * (k - 2));      // INTERPOLATING_EVEN_COEFFS :453-484
                    const float co = k < 3 ? 0.0f : halfband_min(2 * (k - 3) + 1);  // INTERPOLATING_ODD_COEFFS :488-519
                    ae[j] = hin[c][k] * ce + ae[j];
                    ao[j] = hin[c][k] * co + ao[j];
                }
            even[c] = wide_reduce_add8(ae) * 2.0f;
            odd[c] = wide_reduce_add8(ao) * 2.0f;
        }
    }
    FD_HD void push_out(const float* o) {
#pragma unroll
        for (int c = 0; c < OUT; c++) {
#pragma unroll
            for (int k = 0; k < 47; k++) hout[c][k] = hout[c][k + 1];
            hout[c][47] = o[c];
        }
    }
    FD_HD void decimate(float* out) const {  // decimating_filter :43-64
#pragma unroll
        for (int c = 0; c < OUT; c++) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = i * 8 + j;
                    const float cd = k < 5 ? 0.0f : halfband_min(k - 5);  // DECIMATING_COEFFS :381-449
                    acc[j] = hout[c][k] * cd + acc[j];
                }
            out[c] = wide_reduce_add8(acc);
        }
    }
    FD_HD void inner_process_step(int kk, const float* i, float* o) {  // inner sample kk of an inner block of blk_size
        const int ifull = blk_size & ~7;
        if (kk == 0) {
            x.begin_block(blk_size);
            if (ifull == 0) x.end_simd();
        }
        if (kk < ifull) x.template step<PH_SIMD>(i, o); else x.template step<PH_REM>(i, o);
        if (kk + 1 == ifull) x.end_simd();
    }
    template <int PH> FD_HD void step(const float* in, float* out) {
        float even[IN > 0 ? IN : 1], odd[IN > 0 ? IN : 1], o[OUT];
        if (PH == PH_TICK) {  // tick :142-176
            interpolate(in, even, odd);
            x.template step<PH_TICK>(even, o);
            push_out(o);
            x.template step<PH_TICK>(odd, o);
            push_out(o);
            decimate(out);
        } else {              // process :178-212, walked sample by sample
            // Two passes over `size / 2` outer samples each; per pass the inner node processes an inner block of `size` samples (:191-195) -- for an
            // ODD size that is one sample MORE than the 2 (size / 2) the pass interpolated: the last slot of the zero-initialised inner input
            // buffer (BufferArray::new(), :179), whose output nobody reads.  It advances the inner node all the same (an oscillator's phase, a
            // filter's state), so it is rendered here too: after the last pair of each pass, and -- size 1, where a pass is nothing else -- on
            // the untouched outer sample.
            const int half = blk_size >> 1;
            const int i = blk_i++;
            const bool tail_sample = i >= 2 * half;   // the odd last outer sample: never written by the reference
            int extra = 0;                             // zero-input inner samples to render after this outer sample
            if (tail_sample) {
#pragma unroll
                for (int c = 0; c < OUT; c++) out[c] = 0.0f;
                extra = half == 0 ? 2 : 0;             // size 1: two passes of one zero-input inner sample each
            } else {
                const int k = (i >= half ? i - half : i) * 2;
                interpolate(in, even, odd);
                inner_process_step(k, even, o);
                push_out(o);
                inner_process_step(k + 1, odd, o);
                push_out(o);
                decimate(out);
                extra = ((blk_size & 1) && (i == half - 1 || i == 2 * half - 1)) ? 1 : 0;
            }
            // the rare part in a loop that is not unrolled: one more call site of the inner node, not two (an Oversampler around an Oversampler
            // multiplies what is inlined here), and nothing of it in the way of the two steps above
#pragma unroll 1
            for (int r = 0; r < extra; r++) {
                float z[IN > 0 ? IN : 1], drop[OUT];
#pragma unroll
                for (int c = 0; c < IN; c++) z[c] = 0.0f;
                inner_process_step(tail_sample ? 0 : 2 * half, z, drop);
            }
        }
    }
    FD_STEP2_VIA_STEP
};

// Resample<X>  resample.rs:205-315 (ID 69): enclosed GENERATOR X read at a variable speed (input 0; 1 = original) with
// cubic (Catmull-Rom) interpolation.  Only the four newest inner samples are ever read (consumer_i - 1 .. + 2 and the
// producer stops at consumer_i + 3), so the 128-sample ring of the reference becomes a 4-deep history.  No process
// override in the reference: both modes tick, and X is always ticked.
template <class X>
struct Resample {
    static_assert(X::IN == 0, "Resample wraps a generator");
    static constexpr int IN = 1, OUT = X::OUT, RINGS = X::RINGS;
    static constexpr uint64_t ID = 69;
    X x;
    float h[OUT][4];       // newest four inner samples, oldest first
    uint64_t consumer_bits;  // f64 (resample.rs:218)
    uint64_t producer;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
#pragma unroll
        for (int c = 0; c < OUT; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) v.fi(h[c][k], STATE, "buffer", c * 4 + k);
        v.u64(consumer_bits, STATE, "consumer");
        v.u64(producer, STATE, "producer");
    }
    FD_HD void bind(Ctx& c) { x.bind(c); }
    FD_HD void init() {
        x.init();
        for (int c = 0; c < OUT; c++)
            for (int k = 0; k < 4; k++) h[c][k] = 0.0f;
        consumer_bits = __builtin_bit_cast(uint64_t, 1.0);
        producer = 0;
    }
    FD_HD void update(double sr) { x.update(sr); }                                     // :277-279
    FD_HD void reset() {                                                                // :270-275
        x.reset();
        consumer_bits = __builtin_bit_cast(uint64_t, 1.0);
        producer = 0;
    }
    FD_HD uint64_t ping(bool probe, uint64_t hh) { return x.ping(probe, atto(hh, ID)); }  // :308-310
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :281-303
        double consumer = __builtin_bit_cast(double, consumer_bits);
        consumer += (double)__builtin_fmaxf(0.0f, in[0]);
        const double d = consumer - __builtin_floor(consumer);
        const uint64_t ci = (uint64_t)(consumer - d);
        while (ci + 2 >= producer) {
            float inner[OUT];
            x.template step<PH_TICK>(nullptr, inner);
#pragma unroll
            for (int c = 0; c < OUT; c++) {
                h[c][0] = h[c][1]; h[c][1] = h[c][2]; h[c][2] = h[c][3]; h[c][3] = inner[c];
            }
            producer += 1;
        }
        consumer_bits = __builtin_bit_cast(uint64_t, consumer);
#pragma unroll
        for (int c = 0; c < OUT; c++) out[c] = splinef(h[c][0], h[c][1], h[c][2], h[c][3], (float)d);
    }
    FD_STEP2_VIA_STEP
};

// ---------------------------------------------------------------------------------------------------------
// one-pole family (filter.rs): Lowpole (ID 18), Highpole (ID 47), DCBlock (ID 22), Pinkpass (ID 26), Allpole (ID 46),
// and Morph (svf.rs:1040-1111, ID 62).  SURVEY 8(f) row 1: same lane-per-voice skeleton as the biquads.
// ---------------------------------------------------------------------------------------------------------
constexpr int OP_LOWPOLE = 0, OP_HIGHPOLE = 1, OP_DCBLOCK = 2, OP_ALLPOLE = 3;
template <int KIND, int NIN>
struct OnePole {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = KIND == OP_LOWPOLE ? 18 : KIND == OP_HIGHPOLE ? 47 : KIND == OP_DCBLOCK ? 22 : 46;
    float cutoff, sr, coeff, x1, y1;  // Allpole: cutoff = delay (samples), coeff = eta; Lowpole: y1 = value
    template <class V> FD_HD void visit(V& v) {
        constexpr FieldKind PK = NIN > 1 ? STATE : PARAM, CK = NIN > 1 ? STATE : COEF;
        v.f(cutoff, PK, KIND == OP_ALLPOLE ? "delay" : "cutoff");
        v.f(sr, COEF, "sample_rate");
        v.f(coeff, CK, "coeff");
        v.f(x1, STATE, "x1");
        v.f(y1, STATE, "y1");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void set_cutoff(float c) {
        cutoff = c;
        if (KIND == OP_LOWPOLE || KIND == OP_HIGHPOLE) coeff = expf_musl(-F32_TAU * c / sr);  // filter.rs:35-38, 371-374
        if (KIND == OP_DCBLOCK) coeff = 1.0f - F32_TAU / sr * c;                               // :121-124
        if (KIND == OP_ALLPOLE) coeff = (1.0f - c) / (1.0f + c);                               // :292-295
    }
    FD_HD void init() { cutoff = KIND == OP_ALLPOLE ? 1.0f : 440.0f; x1 = y1 = 0.0f; }
    FD_HD void update(double sample_rate) { sr = (float)sample_rate; set_cutoff(cutoff); }
    FD_HD void reset() { x1 = 0.0f; y1 = 0.0f; }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (NIN > 1) {
            if (KIND == OP_ALLPOLE) set_cutoff(in[1]);            // :317-319 unconditional
            else if (in[1] != cutoff) set_cutoff(in[1]);          // :57-62, :393-398
        }
        const float x = in[0];
        if (KIND == OP_LOWPOLE) {                                 // :64-66
            y1 = (1.0f - coeff) * x + coeff * y1;
            out[0] = y1;
        } else if (KIND == OP_HIGHPOLE) {                         // :399-403
            float y0 = coeff * (y1 + x - x1);
            x1 = x; y1 = y0;
            out[0] = y0;
        } else if (KIND == OP_DCBLOCK) {                          // :146-151
            float y0 = x - x1 + coeff * y1;
            x1 = x; y1 = y0;
            out[0] = y0;
        } else {                                                  // Allpole :320-324
            float y0 = coeff * (x - y1) + x1;
            x1 = x; y1 = y0;
            out[0] = y0;
        }
    }
    FD_STEP2_VIA_STEP
};

// Rez<f32, N>  rez.rs:11-105 (ID 75): Paul Kellett's resonant two-pole.  bandpass = 0 lowpass, 1 bandpass.
// NIN = 1 (fixed cutoff / q) or 3 (audio, cutoff, q).
template <int NIN>
struct Rez {
    static constexpr int IN = NIN, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 75;
    float bandpass, cutoff, q, sr, f, fb, buf0, buf1;
    template <class V> FD_HD void visit(V& v) {
        constexpr FieldKind PK = NIN > 1 ? STATE : PARAM, CK = NIN > 1 ? STATE : COEF;
        v.f(bandpass, PARAM, "bandpass");
        v.f(cutoff, PK, "cutoff");
        v.f(q, PK, "q");
        v.f(sr, COEF, "sample_rate");
        v.f(f, CK, "f");
        v.f(fb, CK, "fb");
        v.f(buf0, STATE, "buf0");
        v.f(buf1, STATE, "buf1");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void set_cutoff_q(float c, float qq) {  // :41-46
        cutoff = c;
        f = 2.0f * sinf_musl(F32_PI * c / sr);
        q = qq;
        fb = qq + qq / (1.0f - f);
    }
    FD_HD void init() { bandpass = 0.0f; cutoff = 440.0f; q = 1.0f; buf0 = buf1 = 0.0f; f = fb = 1.0f; }
    FD_HD void update(double sample_rate) { sr = (float)sample_rate; set_cutoff_q(cutoff, q); }  // :62-65
    FD_HD void reset() { buf0 = 0.0f; buf1 = 0.0f; }                                              // :57-60
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :67-82
        if (NIN > 1) {
            if (in[1] != cutoff || in[2] != q) set_cutoff_q(in[1], in[2]);
        }
        float hp = in[0] - buf0;
        float bp = buf0 - buf1;
        buf0 += f * (hp + fb * tanhf_musl(bp));
        buf1 += f * (buf0 - buf1);
        out[0] = buf1 - bandpass * buf0;
    }
    FD_STEP2_VIA_STEP
};

// Follow<f32>  follow.rs:26-131 (ID 24): three one-poles in series; the first sample is taken over as is.
struct Follow {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 24;
    float response_time, sr, coeff, coeff_now, v1, v2, v3;
    template <class V> FD_HD void visit(V& v) {
        v.f(response_time, PARAM, "response_time");
        v.f(sr, COEF, "sample_rate");
        v.f(coeff, COEF, "coeff");
        v.f(coeff_now, STATE, "coeff_now");
        v.f(v1, STATE, "v1"); v.f(v2, STATE, "v2"); v.f(v3, STATE, "v3");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { response_time = 0.1f; coeff = 0.0f; reset(); }  // Follow::new :48-56
    FD_HD void update(double sample_rate) {                              // :96-99 -> set_response_time :61-67
        sr = (float)sample_rate;
        coeff = (float)halfway_coeff((double)(response_time * sr));
        if (coeff_now < 1.0f) coeff_now = coeff;
    }
    FD_HD void reset() { v1 = v2 = v3 = 0.0f; coeff_now = 1.0f; }        // :89-94
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {     // :101-110
        const float c = coeff_now, rc = 1.0f - c;
        v1 = c * in[0] + rc * v1;
        v2 = c * v1 + rc * v2;
        v3 = c * v2 + rc * v3;
        coeff_now = coeff;
        out[0] = v3;
    }
    FD_STEP2_VIA_STEP
};

// AFollow<f32>  follow.rs:132-273 (ID 29): attack / release coefficients, ScalarOrPair for (T, T) combinator.rs:164-173
struct AFollow {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 29;
    float atime, rtime, sr, acoeff, rcoeff, acoeff_now, rcoeff_now, v1, v2, v3;
    template <class V> FD_HD void visit(V& v) {
        v.f(atime, PARAM, "attack_time");
        v.f(rtime, PARAM, "release_time");
        v.f(sr, COEF, "sample_rate");
        v.f(acoeff, COEF, "acoeff");
        v.f(rcoeff, COEF, "rcoeff");
        v.f(acoeff_now, STATE, "acoeff_now");
        v.f(rcoeff_now, STATE, "rcoeff_now");
        v.f(v1, STATE, "v1"); v.f(v2, STATE, "v2"); v.f(v3, STATE, "v3");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { atime = 0.01f; rtime = 0.1f; acoeff = rcoeff = 0.0f; reset(); }
    FD_HD void update(double sample_rate) {  // :217-221 -> set_time :178-192
        sr = (float)sample_rate;
        acoeff = (float)halfway_coeff((double)(atime * sr));
        rcoeff = (float)halfway_coeff((double)(rtime * sr));
        if (acoeff_now < 1.0f) { acoeff_now = acoeff; rcoeff_now = rcoeff; }
    }
    FD_HD void reset() { v1 = v2 = v3 = 0.0f; acoeff_now = 1.0f; rcoeff_now = 1.0f; }  // :209-215
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    FD_HD float pole(float input, float cur) const {
        return cur + __builtin_fmaxf(0.0f, input - cur) * acoeff_now - __builtin_fmaxf(0.0f, cur - input) * rcoeff_now;
    }
    template <int PH> FD_HD void step(const float* in, float* out) {  // :223-246
        v1 = pole(in[0], v1);
        v2 = pole(v1, v2);
        v3 = pole(v2, v3);
        acoeff_now = acoeff;
        rcoeff_now = rcoeff;
        out[0] = v3;
    }
    FD_STEP2_VIA_STEP
};

// MeterState  dynamics.rs:336-394: Meter::Sample (MODE 0), Meter::Peak(timescale) (1), Meter::Rms(timescale) (2).  The
// f64 timescale travels as its bit pattern in a u64 slot.
template <int MODE>
struct MeterCore {
    float smoothing, state;
    uint64_t ts_bits;
    template <class V> FD_HD void visit(V& v) {
        v.u64(ts_bits, PARAM, "timescale");
        v.f(smoothing, COEF, "smoothing");
        v.f(state, STATE, "state");
    }
    FD_HD void init() { smoothing = 0.0f; state = 0.0f; ts_bits = __builtin_bit_cast(uint64_t, 0.1); }
    FD_HD void update(double sr) {  // :355-364
        if (MODE != 0) smoothing = (float)pow_f64(0.5, 1.0 / (__builtin_bit_cast(double, ts_bits) * sr));
    }
    FD_HD void reset() { state = 0.0f; }
    FD_HD void tick(float value) {  // :367-375
        if (MODE == 0) state = value;
        else if (MODE == 1) state = __builtin_fmaxf(state * smoothing, __builtin_fabsf(value));
        else state = state * smoothing + value * value * (1.0f - smoothing);
    }
    FD_HD float level() const { return MODE == 2 ? __builtin_sqrtf(state) : state; }  // :378-384
};

// MeterNode  dynamics.rs:398-438 (ID 61): outputs the meter level; Monitor :441-508 (ID 56): passes the input through and
// keeps the level in its `state` slot (the reference publishes it through a Shared atomic; here the host reads the slot
// with fdsp_bank_get_slot).  Monitor::process ticks a Meter::Sample with the block's last sample only -- the same
// final state as ticking every sample.
template <int MODE, bool MONITOR>
struct MeterT {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = MONITOR ? 56 : 61;
    MeterCore<MODE> m;
    template <class V> FD_HD void visit(V& v) { m.visit(v); }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { m.init(); }
    FD_HD void update(double sr) { m.update(sr); }
    FD_HD void reset() { m.reset(); }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        m.tick(in[0]);
        out[0] = MONITOR ? in[0] : m.level();
    }
    FD_STEP2_VIA_STEP
};

// Var  shared.rs:85-133 (ID 68): outputs a value the host may change at any time -- a per-voice parameter here
// (fdsp_bank_set_param "..:value" plays Shared::set_value).
struct Var {
    static constexpr int IN = 0, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 68;
    float value;
    template <class V> FD_HD void visit(V& v) { v.f(value, PARAM, "value"); }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { value = 0.0f; }
    FD_HD void update(double) {}
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float*, float* out) { out[0] = value; }
    template <int PH> FD_HD void step2(const v2f*, v2f* out) { out[0] = v2f{value, value}; }
};

// Mixer<M, N>  pan.rs:95-150 (ID 84): N outputs, each the dot product of the M inputs with one row of a matrix,
// accumulated from 0.0 in input order (:125-132).  rotate(angle, gain) is the 2 x 2 case (prelude32.rs:2432).
template <int M, int N>
struct Mixer {
    static constexpr int IN = M, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 84;
    float m[N][M];
    template <class V> FD_HD void visit(V& v) {
        _Pragma("unroll") for (int i = 0; i < N; i++)
            _Pragma("unroll") for (int j = 0; j < M; j++) v.fi(m[i][j], PARAM, "matrix", i * M + j);
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { for (int i = 0; i < N; i++) for (int j = 0; j < M; j++) m[i][j] = 0.0f; }
    FD_HD void update(double) {}
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {
        float o[N];
        for (int i = 0; i < N; i++) {
            float value = 0.0f;
            for (int j = 0; j < M; j++) value += in[j] * m[i][j];
            o[i] = value;
        }
        for (int i = 0; i < N; i++) out[i] = o[i];
    }
    FD_STEP2_VIA_STEP
};

// VarFn<F, R>  shared.rs:136-184 (ID 70): f(shared value) -> NO channels; the closure is a functor
//   static void f(float value, float* out)
// and the shared value a per-voice parameter.
template <class FN, int NO>
struct VarFn {
    static constexpr int IN = 0, OUT = NO, RINGS = 0;
    static constexpr uint64_t ID = 70;
    float value;
    template <class V> FD_HD void visit(V& v) { v.f(value, PARAM, "value"); }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { value = 0.0f; }
    FD_HD void update(double) {}
    FD_HD void reset() {}
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float*, float* out) { FN::f(value, out); }
    FD_STEP2_VIA_STEP
};

// Limiter<N>  dynamics.rs:125-241 (ID 25): look-ahead limiter.  Per voice: N delay rings of `length` frames, a max
// reduction tree over the window's amplitudes (ReduceBuffer :59-121, kept in one more ring: [1] = total, leaves from
// leaf_offset), an AFollow over max(1, total * 1.10).  Every voice is at the same ring / tree index, so all accesses
// coalesce across the wave.  No process override: tick arithmetic everywhere.
// `length` = max(1, round(sample_rate * attack_time)) must fit the bank's ring capacity: leaf_offset + length + 1 slots.
template <int N>
struct Limiter {
    static constexpr int IN = N, OUT = N, RINGS = N + 1;
    static constexpr uint64_t ID = 25;
    float attack, release;  // params (Limiter::new arguments)
    AFollow follower;
    uint32_t length, leaf, index, fill;
    float last_sr;
    // windows of four frames and more: the walk is INCREMENTAL -- the values of the last update's path and of its siblings stay in registers
    // (slots between launches), see tree_set_inc
    static constexpr int MAXL = 20;
    float pv[MAXL], sv[MAXL];
    uint32_t prev;
    float* buf[N];
    float* tree;
    size_t vs;
    uint32_t cap;
    Ctx owner;
    template <class V> FD_HD void visit(V& v) {
        v.f(attack, PARAM, "attack_time");
        v.f(release, PARAM, "release_time");
        v.enter(0); follower.visit(v); v.leave();
        v.u32(length, COEF, "length");
        v.u32(leaf, COEF, "leaf_offset");
        v.f(last_sr, COEF, "sample_rate");
        v.u32(index, STATE, "index");
        v.u32(fill, STATE, "fill");
        _Pragma("unroll") for (int l = 0; l < MAXL; l++) v.fi(pv[l], STATE, "path", l);
        _Pragma("unroll") for (int l = 0; l < MAXL; l++) v.fi(sv[l], STATE, "sibling", l);
        v.u32(prev, STATE, "previous_leaf");
    }
    FD_HD void bind(Ctx& c) {
        for (int i = 0; i < N; i++) buf[i] = c.claim_ring();
        tree = c.claim_ring();
        vs = c.vstride;
        cap = c.ring_cap;
        owner = c;
    }
    FD_HD void init() {
        attack = 0.005f; release = 0.05f;
        follower.init();
        length = 0; leaf = 1; index = 0; fill = 0; last_sr = 0.0f;
        prev = 0;
        for (int l = 0; l < MAXL; l++) pv[l] = sv[l] = 0.0f;
    }
    FD_HD void clear() {  // reducer.clear(), buffer.clear(), index = 0  (:190-199)
        index = 0;
        fill = 0;
        prev = 0;
        for (int l = 0; l < MAXL; l++) pv[l] = sv[l] = 0.0f;
        for (uint32_t k = 0; k < cap; k++) tree[(size_t)k * vs] = 0.0f;
    }
    FD_HD void update(double sr) {  // Limiter::new :159-171 + set_sample_rate :188-199
        follower.atime = attack * 0.4f;
        follower.rtime = release * 0.4f;
        follower.update(sr);
        double want = __builtin_round(sr * (double)attack);
        uint32_t len = want < 1.0 ? 1u : (want < 1073741824.0 ? (uint32_t)want : 1073741824u);
        owner.want_positions(next_pow2_u32(len) + len + (len & 1u));  // capacity fixed at bank creation: the host fails the call
        while (len > 1u && next_pow2_u32(len) + len + (len & 1u) > cap) len--;
        if (len != length || last_sr != (float)sr) {
            length = len;
            leaf = next_pow2_u32(len);
            last_sr = (float)sr;
            clear();
        }
    }
    FD_HD void reset() { clear(); }  // :184-186 (the follower keeps its state, only its coefficients are refreshed)
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    // ReduceBuffer::set :104-112, returns the new total (= buffer[1]).  The walk to the root reads the SIBLING of every node on the path and
    // writes its parent; the siblings are off the path, so no load of this update sees a store of this update: CH levels' loads are issued
    // together, then their maxima and stores (one memory round trip per CH levels instead of one per level -- the 4 410-frame window of the
    // reference's own limiter bench is a 13-level tree; the loop as the reference writes it, load / max / store per level, ran 172 ms per rendered
    // second of 65 536 instances on the latency of its loads alone).  Windows up to 256 frames: 8 levels at a time; longer ones: 16.
    template <int CH> FD_HD float tree_walk(uint32_t i, float cur) {
        while (i > 1u) {
            float s[CH];
            _Pragma("unroll") for (int k = 0; k < CH; k++) {
                const uint32_t j = i >> k;
                s[k] = j > 1u ? tree[(size_t)(j ^ 1u) * vs] : 0.0f;
            }
            _Pragma("unroll") for (int k = 0; k < CH; k++) {
                const uint32_t j = i >> k;
                if (j > 1u) {
                    cur = __builtin_fmaxf(cur, s[k]);
                    tree[(size_t)(j >> 1) * vs] = cur;
                }
            }
            i >>= CH;
        }
        return cur;
    }
    // ... and not even that (windows of four frames and more).  Consecutive updates touch neighbouring leaves: the new path shares every node above level h (the
    // highest bit in which the two leaf indices differ) with the previous one, so above h the siblings are the previous update's siblings, AT h the
    // sibling is the previous path's node, and only below h -- h is the number of trailing zeros of an incremented index: one level on average -- are
    // the siblings nodes this lap has not reached yet, final since the last lap, to be loaded.  The nodes the path LEAVES (the previous path's levels
    // 1..h) are complete at that moment and are written then; a node the path is still inside is nobody's sibling until the path leaves it, so its
    // running value lives in registers only (`pv`; `sv` the sibling cache; both slots between launches), and the root is never stored -- the total
    // comes back in a register.  Every node that is ever LOADED holds exactly what ReduceBuffer::set would have left in it (max is exact in any
    // order); per update one leaf store + on average one load and one store instead of 13 + 14 for the reference bench's 4 410-frame window:
    // 119 -> ~45 ms per rendered second of 65 536 instances (the kernel was bound by the scattered 256-byte rows of that traffic, not by latency:
    // more waves per SIMD made it slower).
    template <int LV> FD_HD float tree_set_inc(uint32_t i, float value) {  // LV: levels unrolled (>= the tree's height below the root)
        const uint32_t ip = prev ? prev : i;  // the first update after a clear leaves nothing behind (all-zero tree, all-zero caches)
        const uint32_t x = i ^ ip;
        const int h = x ? 31 - __builtin_clz(x) : -1;
        float ld[LV];
        _Pragma("unroll") for (int l = 0; l < LV; l++) ld[l] = (l < h) ? tree[(size_t)((i >> l) ^ 1u) * vs] : 0.0f;
        _Pragma("unroll") for (int l = 1; l < LV; l++)
            if (l <= h) tree[(size_t)(ip >> l) * vs] = pv[l];
        tree[(size_t)i * vs] = value;
        float cur = value;
        _Pragma("unroll") for (int l = 0; l < LV; l++) {
            if ((i >> l) > 1u) {  // level l is below the root
                const float s = l < h ? ld[l] : (l == h ? pv[l] : sv[l]);
                pv[l] = cur;
                sv[l] = s;
                cur = __builtin_fmaxf(cur, s);
            }
        }
        prev = i;
        return cur;
    }
    FD_HD float tree_set(uint32_t idx, float value) {
        const uint32_t i = leaf + idx;
        if (leaf >= 4u && leaf <= (1u << MAXL))  // (branches on the tree's height: uniform unless voices differ in attack time)
            return leaf <= (1u << 7) ? tree_set_inc<7>(i, value) : leaf <= (1u << 10) ? tree_set_inc<10>(i, value) : leaf <= (1u << 13) ? tree_set_inc<13>(i, value)
                   : leaf <= (1u << 16) ? tree_set_inc<16>(i, value) : tree_set_inc<MAXL>(i, value);
        tree[(size_t)i * vs] = value;  // windows of one to three frames, and of more than 2^20: the walk level by level
        return leaf > 256u ? tree_walk<16>(i, value) : tree_walk<8>(i, value);
    }
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :202-226
        float amplitude = 0.0f;
        for (int c = 0; c < N; c++) amplitude = __builtin_fmaxf(amplitude, __builtin_fabsf(in[c]));
        float o[N];  // the delay line's oldest sample, read ahead of the tree walk (its round trip overlaps the walk's; unused while the line fills)
        for (int c = 0; c < N; c++) o[c] = buf[c][(size_t)index * vs];
        const float total = tree_set(index, amplitude);  // reducer.total() = buffer[1], the root the walk just wrote
        if (fill < length) {
            for (int c = 0; c < N; c++) buf[c][(size_t)index * vs] = in[c];
            fill++;
            if (fill == length) follower.v1 = follower.v2 = follower.v3 = total;  // set_value :196-200
            for (int c = 0; c < N; c++) out[c] = 0.0f;
        } else {
            for (int c = 0; c < N; c++) buf[c][(size_t)index * vs] = in[c];
            float x = __builtin_fmaxf(1.0f, total * 1.10f), y;
            follower.template step<PH_TICK>(&x, &y);
            const float limit = follower.v3;
            const float z = 1.0f / limit;
            for (int c = 0; c < N; c++) out[c] = o[c] * z;
        }
        index++;
        if (index >= length) index = 0;
    }
    FD_STEP2_VIA_STEP
};

// Hold  noise.rs:242-322 (ID 76): sample-and-hold of input 0 at the frequency on input 1 (Hz); each hold lasts
// lerp(1 - variability, 1 + variability, rnd.f64()) / frequency seconds.  The draws come from funutd's
// `Rnd::from_u64(hash)`, a third-party generator whose source is not under /root/reference, so -- as for Pluck's
// excitation -- the host uploads that stream (fdsp_bank_set_ring, this node's ring: each f64 draw as two consecutive
// f32 words, low word first) after reading the node's ":hash" slot; the ring wraps if a render outlasts it.  reset()
// re-seeds the generator in the reference (:281-285): the stream restarts.  Time is f64; no process override.
struct Hold {
    static constexpr int IN = 2, OUT = 1, RINGS = 1;
    static constexpr uint64_t ID = 76;
    float variability, hold;
    uint32_t pos;
    uint64_t sd_bits, t_bits, next_bits, hash;
    float* ring;
    size_t vs;
    uint32_t cap;
    template <class V> FD_HD void visit(V& v) {
        v.f(variability, PARAM, "variability");
        v.u64(sd_bits, COEF, "sample_duration");
        v.u64(t_bits, STATE, "t");
        v.u64(next_bits, STATE, "next_t");
        v.f(hold, STATE, "hold");
        v.u32(pos, STATE, "draws");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx& c) { ring = c.claim_ring(); vs = c.vstride; cap = c.ring_cap; }
    FD_HD void init() { variability = 0.0f; hold = 0.0f; hash = 0; sd_bits = 0; reset(); }
    FD_HD void update(double sr) { sd_bits = __builtin_bit_cast(uint64_t, 1.0 / sr); }  // :287-289
    FD_HD void reset() { pos = 0; t_bits = 0; next_bits = 0; }                             // :281-285
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :312-315
            hash = h;
            reset();
        }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :292-304
        double t = __builtin_bit_cast(double, t_bits);
        const double next_t = __builtin_bit_cast(double, next_bits);
        if (t >= next_t) {
            hold = in[0];
            const uint32_t draws = cap / 2u;
            double r = 0.5;  // a ring too small to hold one draw: the mean hold length
            if (draws) {
                const uint32_t k = pos % draws;
                const uint64_t lo = __builtin_bit_cast(uint32_t, ring[(size_t)(2u * k) * vs]);
                const uint64_t hi = __builtin_bit_cast(uint32_t, ring[(size_t)(2u * k + 1u) * vs]);
                r = __builtin_bit_cast(double, lo | (hi << 32));
            }
            pos += 1u;
            const double a = 1.0 - (double)variability, b = 1.0 + (double)variability;
            next_bits = __builtin_bit_cast(uint64_t, t + (a * (1.0 - r) + b * r) / (double)in[1]);  // lerp math.rs:169-178
        }
        t += __builtin_bit_cast(double, sd_bits);
        t_bits = __builtin_bit_cast(uint64_t, t);
        out[0] = hold;
    }
    FD_STEP2_VIA_STEP
};

// Mls  noise.rs:14-151 (ID 19): maximum length sequence noise, integer-exact.  `bits` (1..31) is a parameter; a bank
// starts in MlsState::new's all-ones state of the default 29-bit sequence until reset / set_seed re-derives it.
FD_HD uint32_t mls_poly(uint32_t n) {  // MLS_POLY noise.rs:23-55
    constexpr uint32_t P[31] = {0x1u, 0x3u, 0x6u, 0xCu, 0x14u, 0x30u, 0x48u, 0xB8u, 0x110u, 0x240u, 0x500u, 0xCA0u,
                                0x1B00u, 0x3088u, 0x6000u, 0xD008u, 0x12000u, 0x20400u, 0x63000u, 0x90000u, 0x140000u,
                                0x300000u, 0x420000u, 0xE10000u, 0x1200000u, 0x2000023u, 0x4000013u, 0x9000000u,
                                0x14000000u, 0x20000029u, 0x48000000u};
    uint32_t r = P[0];
#pragma unroll
    for (int i = 1; i < 31; i++) r = (n == (uint32_t)i + 1u) ? P[i] : r;  // select chain: no per-lane table in scratch
    return r;
}
struct Mls {
    static constexpr int IN = 0, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 19;
    float bits, has_seed;
    uint32_t state, poly;
    uint64_t seed, hash;
    template <class V> FD_HD void visit(V& v) {
        v.f(bits, PARAM, "bits");
        v.u32(state, STATE, "state");
        v.u32(poly, COEF, "poly");
        v.f(has_seed, PARAM, "has_seed");
        v.u64(seed, PARAM, "seed");
        v.u64(hash, STATE, "hash");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD uint32_t n() const { return (uint32_t)bits; }
    FD_HD void init() {  // mls() = mls_bits(29): Mls::new(MlsState::new(29)) :58-62,109-117
        bits = 29.0f; has_seed = 0.0f; seed = 0; hash = 0;
        state = (1u << 29) - 1u;
        poly = mls_poly(29);
    }
    FD_HD void update(double) { poly = mls_poly(n()); }
    FD_HD void reset() {  // :124-127 -> MlsState::new_with_seed :66-72
        uint64_t h = has_seed != 0.0f ? seed : hash;
        uint32_t s32 = (uint32_t)(h ^ (h >> 32));
        state = 1u + s32 % ((1u << n()) - 1u);
    }
    FD_HD uint64_t ping(bool probe, uint64_t h) {
        if (!probe) {  // set_hash :142-145
            hash = h;
            reset();
        }
        return atto(h, ID);
    }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float*, float* out) {  // tick :129-134, MlsState::next / value :81-96
        const uint32_t nn = n();
        float value = (float)((state >> (nn - 1u)) & 1u);
        uint32_t parity = (uint32_t)__builtin_popcount(poly & state) & 1u;
        state = ((state << 1) | parity) & ((1u << nn) - 1u);
        out[0] = value * 2.0f - 1.0f;
    }
    FD_STEP2_VIA_STEP
};

// Pinkpass  filter.rs:178-262 (Paul Kellett's pinking filter)
struct Pinkpass {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 26;
    float b[7];
    template <class V> FD_HD void visit(V& v) {
        _Pragma("unroll") for (int i = 0; i < 7; i++) v.fi(b[i], STATE, "b", i);
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { reset(); }
    FD_HD void update(double) {}
    FD_HD void reset() { for (int i = 0; i < 7; i++) b[i] = 0.0f; }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // :226-246
        const float x = in[0];
        b[0] = (float)0.99886 * b[0] + x * (float)0.0555179;
        b[1] = (float)0.99332 * b[1] + x * (float)0.0750759;
        b[2] = (float)0.96900 * b[2] + x * (float)0.1538520;
        b[3] = (float)0.86650 * b[3] + x * (float)0.3104856;
        b[4] = (float)0.55000 * b[4] + x * (float)0.5329522;
        b[5] = (float)-0.7616 * b[5] - x * (float)0.0168980;
        out[0] = (b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + x * (float)0.5362) * (float)0.115830421;
        b[6] = x * (float)0.115926;
    }
    FD_STEP2_VIA_STEP
};

// Morph<f32>  svf.rs:1040-1111 (ID 62): Svf<PeakMode> + dry mix; inputs (audio, cutoff, q, morph).
struct Morph {
    static constexpr int IN = 4, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 62;
    Svf<3> filter;
    float morph;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); filter.visit(v); v.leave();
        v.f(morph, STATE, "morph");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() {  // Morph::new :1046-1061: gain = 0
        filter.init();
        filter.mode = (float)SVF_PEAK;
        filter.gain = 0.0f;
        morph = 0.0f;
    }
    FD_HD void update(double sr) { filter.update(sr); }
    FD_HD void reset() { filter.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return atto(filter.ping(probe, h), ID); }  // :1108-1110
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float* in, float* out) {  // tick :1076-1080 == process :1083-1095
        morph = in[3];
        float fo;
        filter.template step<PH>(in, &fo);
        out[0] = (fo + in[3] * in[0]) * 0.5f;
    }
    FD_STEP2_VIA_STEP
};

// ---------------------------------------------------------------------------------------------------------
// combinators
// ---------------------------------------------------------------------------------------------------------

// Pipe<X, Y>  audionode.rs:1375-1492 (ID 6)
template <class X, class Y>
struct Pipe {
    static_assert(X::OUT == Y::IN, "Pipe arity mismatch");
    static constexpr int IN = X::IN, OUT = Y::OUT, RINGS = X::RINGS + Y::RINGS;
    static constexpr uint64_t ID = 6;
    X x;
    Y y;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        v.enter(1); y.visit(v); v.leave();
    }
    FD_HD void init() { x.init(); y.init(); }
    FD_HD void update(double sr) { x.update(sr); y.update(sr); }
    FD_HD void reset() { x.reset(); y.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return y.ping(probe, x.ping(probe, atto(h, ID))); }  // :1459
    FD_HD void end_simd() { x.end_simd(); y.end_simd(); }
    FD_HD void begin_block(int n) { x.begin_block(n); y.begin_block(n); }
    FD_HD void bind(Ctx& a) { x.bind(a); y.bind(a); }
    FD_HD bool tripped() const { return x.tripped() || y.tripped(); }
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[X::OUT > 0 ? X::OUT : 1];
        x.template step<PH>(in, t);
        y.template step<PH>(t, out);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        v2f t[X::OUT > 0 ? X::OUT : 1];
        x.template step2<PH>(in, t);
        y.template step2<PH>(t, out);
    }
    // x in full (y's state advance needs its input), y skipped
    template <int PH> FD_HD void skip(const float* in) {
        float t[X::OUT > 0 ? X::OUT : 1];
        x.template step<PH>(in, t);
        y.template skip<PH>(t);
    }
    template <int PH> FD_HD void skip2(const v2f* in) {
        v2f t[X::OUT > 0 ? X::OUT : 1];
        x.template step2<PH>(in, t);
        y.template skip2<PH>(t);
    }
};

// Stack<X, Y>  audionode.rs:1496-1650 (ID 7)
template <class X, class Y>
struct Stack {
    static constexpr int IN = X::IN + Y::IN, OUT = X::OUT + Y::OUT, RINGS = X::RINGS + Y::RINGS;
    static constexpr uint64_t ID = 7;
    X x;
    Y y;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        v.enter(1); y.visit(v); v.leave();
    }
    FD_HD void init() { x.init(); y.init(); }
    FD_HD void update(double sr) { x.update(sr); y.update(sr); }
    FD_HD void reset() { x.reset(); y.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return y.ping(probe, x.ping(probe, atto(h, ID))); }
    FD_HD void end_simd() { x.end_simd(); y.end_simd(); }
    FD_HD void begin_block(int n) { x.begin_block(n); y.begin_block(n); }
    FD_HD void bind(Ctx& a) { x.bind(a); y.bind(a); }
    FD_HD bool tripped() const { return x.tripped() || y.tripped(); }
    template <int PH> FD_HD void step(const float* in, float* out) {
        x.template step<PH>(in, out);
        y.template step<PH>(in + X::IN, out + X::OUT);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        x.template step2<PH>(in, out);
        y.template step2<PH>(in + X::IN, out + X::OUT);
    }
    template <int PH> FD_HD void skip(const float* in) { x.template skip<PH>(in); y.template skip<PH>(in + X::IN); }
    template <int PH> FD_HD void skip2(const v2f* in) { x.template skip2<PH>(in); y.template skip2<PH>(in + X::IN); }
};

struct OpAdd { template <class T> static FD_HD T f(T a, T b) { return a + b; } };
struct OpSub { template <class T> static FD_HD T f(T a, T b) { return a - b; } };
struct OpMul { template <class T> static FD_HD T f(T a, T b) { return a * b; } };

// Binop<B, X, Y>  audionode.rs:850-1027 (ID 3)
template <class OP, class X, class Y>
struct Binop {
    static_assert(X::OUT == Y::OUT, "Binop arity mismatch");
    static constexpr int IN = X::IN + Y::IN, OUT = X::OUT, RINGS = X::RINGS + Y::RINGS;
    static constexpr uint64_t ID = 3;
    X x;
    Y y;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        v.enter(1); y.visit(v); v.leave();
    }
    FD_HD void init() { x.init(); y.init(); }
    FD_HD void update(double sr) { x.update(sr); y.update(sr); }
    FD_HD void reset() { x.reset(); y.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return y.ping(probe, x.ping(probe, atto(h, ID))); }  // :966
    FD_HD void end_simd() { x.end_simd(); y.end_simd(); }
    FD_HD void begin_block(int n) { x.begin_block(n); y.begin_block(n); }
    FD_HD void bind(Ctx& a) { x.bind(a); y.bind(a); }
    FD_HD bool tripped() const { return x.tripped() || y.tripped(); }
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT];
        x.template step<PH>(in, t);
        y.template step<PH>(in + X::IN, out);
        for (int i = 0; i < OUT; i++) out[i] = OP::f(t[i], out[i]);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        v2f t[OUT];
        x.template step2<PH>(in, t);
        y.template step2<PH>(in + X::IN, out);
        for (int i = 0; i < OUT; i++) out[i] = OP::f(t[i], out[i]);
    }
};

// FrameUnop variants audionode.rs:1030-1228; the scalar is a per-voice parameter
struct UNeg { static constexpr bool HAS_SCALAR = false; template <class T> static FD_HD T f(T x, float) { return -x; } };
struct UAddScalar { static constexpr bool HAS_SCALAR = true; template <class T> static FD_HD T f(T x, float s) { return x + s; } };
struct UNegAddScalar { static constexpr bool HAS_SCALAR = true; template <class T> static FD_HD T f(T x, float s) { return -x + s; } };
struct UMulScalar { static constexpr bool HAS_SCALAR = true; template <class T> static FD_HD T f(T x, float s) { return x * s; } };

// Unop<X, U>  audionode.rs:1232-1326 (ID 4)
template <class X, class U>
struct Unop {
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = X::RINGS;
    static constexpr uint64_t ID = 4;
    X x;
    float scalar;
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        if (U::HAS_SCALAR) v.f(scalar, PARAM, "scalar");
    }
    FD_HD void init() { x.init(); scalar = 0.0f; }
    FD_HD void update(double sr) { x.update(sr); }
    FD_HD void reset() { x.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return x.ping(probe, atto(h, ID)); }  // :1286
    FD_HD void end_simd() { x.end_simd(); }
    FD_HD void begin_block(int n) { x.begin_block(n); }
    FD_HD void bind(Ctx& a) { x.bind(a); }
    FD_HD bool tripped() const { return x.tripped(); }
    template <int PH> FD_HD void step(const float* in, float* out) {
        x.template step<PH>(in, out);
        for (int i = 0; i < OUT; i++) out[i] = U::f(out[i], scalar);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        x.template step2<PH>(in, out);
        for (int i = 0; i < OUT; i++) out[i] = U::f(out[i], scalar);
    }
    template <int PH> FD_HD void skip(const float* in) { x.template skip<PH>(in); }
    template <int PH> FD_HD void skip2(const v2f* in) { x.template skip2<PH>(in); }
};

// ---- type-level variants of a graph -------------------------------------------------------------------------------
// A variant replaces leaf types by layout-identical derived types (same fields, same visit order, same ID) that differ
// only in the arithmetic of the packed path; the combinators Pipe / Stack / Binop / Unop carry the replacement through.
// LpOf<G>:   FixedSvf -> FixedSvfLp.  Chosen per WAVE on the device when lp_ok(g) holds in all lanes; exact.
// FastOf<G>: Sine -> SineFast, Moog<N> -> MoogFast<N>.  Chosen per LAUNCH by the host (fdsp_set_option("math", 1)); tolerance mode.
template <class G> struct LpOf { using type = G; };
template <> struct LpOf<FixedSvf> { using type = FixedSvfLp; };
template <class G> struct PlainOf { using type = G; };     // Sine -> SinePlain (layout-identical; producer stages only)
template <> struct PlainOf<Sine> { using type = SinePlain; };
template <class G> struct FastOf { using type = G; };
template <> struct FastOf<Sine> { using type = SineFast; };
template <int N> struct FastOf<Moog<N>> { using type = MoogFast<N>; };
#define FD_VARIANT_THROUGH(TRAIT)                                                                                       \
    template <class X, class Y> struct TRAIT<Pipe<X, Y>> { using type = Pipe<typename TRAIT<X>::type, typename TRAIT<Y>::type>; };    \
    template <class X, class Y> struct TRAIT<Stack<X, Y>> { using type = Stack<typename TRAIT<X>::type, typename TRAIT<Y>::type>; };  \
    template <class O, class X, class Y> struct TRAIT<Binop<O, X, Y>> { using type = Binop<O, typename TRAIT<X>::type, typename TRAIT<Y>::type>; }; \
    template <class X, class U> struct TRAIT<Unop<X, U>> { using type = Unop<typename TRAIT<X>::type, U>; };
FD_VARIANT_THROUGH(LpOf)
FD_VARIANT_THROUGH(FastOf)
FD_VARIANT_THROUGH(PlainOf)
template <class A, class B> struct SameType { static constexpr bool v = false; };
template <class A> struct SameType<A, A> { static constexpr bool v = true; };
template <class T> struct Pointee;
template <class T> struct Pointee<T*> { using type = T; };

// lp_ok(g): this lane's FixedSvf nodes (those LpOf reaches) all satisfy FixedSvfLp's preconditions
template <class G> FD_HD bool lp_ok(const G&) { return true; }
FD_HD bool lp_ok(const FixedSvf& f) { return svf_is_plain_lowpass(f); }
template <class X, class Y> FD_HD bool lp_ok(const Pipe<X, Y>& g);
template <class X, class Y> FD_HD bool lp_ok(const Stack<X, Y>& g);
template <class O, class X, class Y> FD_HD bool lp_ok(const Binop<O, X, Y>& g);
template <class X, class U> FD_HD bool lp_ok(const Unop<X, U>& g);
template <class X, class Y> FD_HD bool lp_ok(const Pipe<X, Y>& g) { return lp_ok(g.x) && lp_ok(g.y); }
template <class X, class Y> FD_HD bool lp_ok(const Stack<X, Y>& g) { return lp_ok(g.x) && lp_ok(g.y); }
template <class O, class X, class Y> FD_HD bool lp_ok(const Binop<O, X, Y>& g) { return lp_ok(g.x) && lp_ok(g.y); }
template <class X, class U> FD_HD bool lp_ok(const Unop<X, U>& g) { return lp_ok(g.x); }

// item_begin(g): told by the packed loops that an 8-sample SIMD item starts with the next frame.  Nodes that do something
// once per item (WaveSynth picks its table pair from the item's first frequency) keep a counter for that; at an item
// start the counter is 0 mod 8 anyway, so writing the constant is no change of state -- but it lets the compiler fold
// the `counter & 7` tests of the item's four frame pairs, and with the branches gone the table gathers of all four
// pairs are issued back to back instead of one L2 round trip per pair.  Reaches through the plain combinators only.
template <class G> FD_HD void item_begin(G&) {}
template <int SET, int NOUT> FD_HD void item_begin(WaveSynth<SET, NOUT>& w) { w.item_pos = 0; }
template <class X, class Y> FD_HD void item_begin(Pipe<X, Y>& g);
template <class X, class Y> FD_HD void item_begin(Stack<X, Y>& g);
template <class O, class X, class Y> FD_HD void item_begin(Binop<O, X, Y>& g);
template <class X, class U> FD_HD void item_begin(Unop<X, U>& g);
template <class X, class Y> FD_HD void item_begin(Pipe<X, Y>& g) { item_begin(g.x); item_begin(g.y); }
template <class X, class Y> FD_HD void item_begin(Stack<X, Y>& g) { item_begin(g.x); item_begin(g.y); }
template <class O, class X, class Y> FD_HD void item_begin(Binop<O, X, Y>& g) { item_begin(g.x); item_begin(g.y); }
template <class X, class U> FD_HD void item_begin(Unop<X, U>& g) { item_begin(g.x); }

// ---------------------------------------------------------------------------------------------------------
// routing leaves and the remaining combinators (audionode.rs).  All of them are arithmetic-free or a handful of
// adds; they exist so that any graph the reference's operators can spell fuses into the one per-voice kernel.
// ---------------------------------------------------------------------------------------------------------

#define FD_STATELESS_LEAF                                            \
    template <class V> FD_HD void visit(V&) {}                       \
    FD_HD void init() {}                                             \
    FD_HD void update(double) {}                                     \
    FD_HD void reset() {}                                            \
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }    \
    FD_HD void end_simd() {}                                         \
    FD_HD void begin_block(int) {}                                   \
    FD_HD bool tripped() const { return false; }                     \
    FD_HD void bind(Ctx&) {}

// MultiPass<N>  audionode.rs:373-403 (ID 0)
template <int N>
struct MultiPass {
    static constexpr int IN = N, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 0;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) { for (int i = 0; i < N; i++) out[i] = in[i]; }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) { for (int i = 0; i < N; i++) out[i] = in[i]; }
};

// Sink<N>  audionode.rs:437-462 (ID 1)
template <int N>
struct Sink {
    static constexpr int IN = N, OUT = 0, RINGS = 0;
    static constexpr uint64_t ID = 1;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float*, float*) {}
    template <int PH> FD_HD void step2(const v2f*, v2f*) {}
};

// MultiSplit<M, N>: M inputs copied to N branches, output j + i*M = input j  audionode.rs:571-613 (ID 38);
// Split<N> = the M == 1 case with its own ID  audionode.rs:527-568 (ID 40)
template <int M, int N>
struct MultiSplit {
    static constexpr int IN = M, OUT = M * N, RINGS = 0;
    static constexpr uint64_t ID = 38;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) { for (int i = 0; i < M * N; i++) out[i] = in[i % M]; }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) { for (int i = 0; i < M * N; i++) out[i] = in[i % M]; }
};
template <int N>
struct Split {
    static constexpr int IN = 1, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 40;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) { for (int i = 0; i < N; i++) out[i] = in[0]; }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) { for (int i = 0; i < N; i++) out[i] = in[0]; }
};

// MultiJoin<M, N>  audionode.rs:668-730 (ID 39) and Join<N> :617-660 (ID 41): average N branches.
// tick sums then DIVIDES by N (:643-648, :700-708); process scales every term by z = 1/N and sums (:649-659, :710-724).
template <int M, int N, uint64_t JID>
struct JoinT {
    static constexpr int IN = M * N, OUT = M, RINGS = 0;
    static constexpr uint64_t ID = JID;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_TICK) {
            for (int j = 0; j < M; j++) {
                float o = in[j];
                for (int i = 1; i < N; i++) o += in[j + i * M];
                out[j] = o / (float)N;
            }
        } else {
            const float z = 1.0f / (float)N;
            for (int j = 0; j < M; j++) {
                float o = in[j] * z;
                for (int i = 1; i < N; i++) o += in[j + i * M] * z;
                out[j] = o;
            }
        }
    }
    FD_STEP2_VIA_STEP
};
template <int N> using Join = JoinT<1, N, 41>;
template <int M, int N> using MultiJoin = JoinT<M, N, 39>;

// Reverse<N>  audionode.rs:2808-2837 (ID 45)
template <int N>
struct Reverse {
    static constexpr int IN = N, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 45;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) { for (int i = 0; i < N; i++) out[i] = in[N - 1 - i]; }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) { for (int i = 0; i < N; i++) out[i] = in[N - 1 - i]; }
};

// Impulse<N>  audionode.rs:2841-2875 (ID 81): one on the first sample after reset, zero afterwards
template <int N>
struct Impulse {
    static constexpr int IN = 0, OUT = N, RINGS = 0;
    static constexpr uint64_t ID = 81;
    float value;
    template <class V> FD_HD void visit(V& v) { v.f(value, STATE, "value"); }
    FD_HD void init() { value = 1.0f; }
    FD_HD void update(double) {}
    FD_HD void reset() { value = 1.0f; }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void bind(Ctx&) {}
    template <int PH> FD_HD void step(const float*, float* out) {
        for (int i = 0; i < N; i++) out[i] = value;
        value = 0.0f;
    }
    FD_STEP2_VIA_STEP
};

// Map<M, I, O>  audionode.rs:1330-1371 (ID 5): the closure is a functor type `FN` with
//   static void f(const float* in, float* out)     (one frame: NI inputs -> NO outputs)
// supplied as source to fdsp_graph_compile_src.  Map has no process override, so every phase is the tick arithmetic.
template <class FN, int NI, int NO>
struct Map {
    static constexpr int IN = NI, OUT = NO, RINGS = 0;
    static constexpr uint64_t ID = 5;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) { FN::f(in, out); }
    FD_STEP2_VIA_STEP
};

// WavePlayer  wave.rs:739-797 (ID 65): plays channel `channel` of a shared Wave from start_point to end_point, jumping to
// loop_point when it reaches the end (0xFFFFFFFF = no loop: silence afterwards).  The Wave lives in one of the
// process-wide sample slots (fdsp_wave_upload); positions are per voice, so a bank is a sampler with independent heads
// (resample(playwave(..)) gives each voice its own playback speed).  No process override.
template <int SLOT>
struct WavePlayer {
    static constexpr int IN = 0, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 65;
    uint32_t channel, index, start_point, end_point, loop_point;
    const WaveBuf* wb;
    template <class V> FD_HD void visit(V& v) {
        v.u32(channel, PARAM, "channel");
        v.u32(start_point, PARAM, "start_point");
        v.u32(end_point, PARAM, "end_point");
        v.u32(loop_point, PARAM, "loop_point");
        v.u32(index, STATE, "index");
    }
    FD_HD void bind(Ctx& a) { wb = &a.aux->wave[SLOT]; }
    FD_HD void init() { channel = 0; start_point = 0; end_point = 0; loop_point = 0xFFFFFFFFu; index = 0; }
    // WavePlayer::new starts at start_point (:761); parameters arrive after construction here, so a head that has not
    // moved yet follows its start point
    FD_HD void update(double) { if (index < start_point || index > end_point) index = start_point; }
    FD_HD void reset() { index = start_point; }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD void begin_block(int) {}
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    template <int PH> FD_HD void step(const float*, float* out) {  // tick :779-792
        float value = 0.0f;
        const uint32_t end = end_point < wb->length ? end_point : wb->length;  // new() asserts end_point <= length
        if (index < end && channel < wb->channels) {
            value = gload(wb->data + ((size_t)channel * wb->length + index));
            index += 1;
            if (index == end_point && loop_point != 0xFFFFFFFFu) index = loop_point;
        }
        out[0] = value;
    }
    FD_STEP2_VIA_STEP
};

// PulseWave  wavetable.rs:437-491 (ID 44): a pulse wave as the difference of two saw waves half a pulse width apart,
//   (WaveSynth<U2>(saw) | pass()) >> (pass() | (pass() + pass()) >> PhaseSynth(saw)) >> pass() - pass()
// input 0 = frequency, input 1 = pulse width in 0...1.  ping() pings the inner graph FIRST and hashes its own ID last
// (:484-486); PulseWave::new does not ping, the inner Pipe's constructor did (CtorPing<PulseWave> in fd_device.hpp).
struct PulseWave {
    using Inner = Pipe<Pipe<Stack<WaveSynth<0, 2>, Pass>, Stack<Pass, Pipe<Binop<OpAdd, Pass, Pass>, PhaseSynth<0>>>>,
                       Binop<OpSub, Pass, Pass>>;
    static constexpr int IN = 2, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 44;
    Inner pulse;
    template <class V> FD_HD void visit(V& v) { v.enter(0); pulse.visit(v); v.leave(); }
    FD_HD void init() { pulse.init(); }
    FD_HD void update(double sr) { pulse.update(sr); }
    FD_HD void reset() { pulse.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return atto(pulse.ping(probe, h), ID); }
    FD_HD void end_simd() { pulse.end_simd(); }
    FD_HD void begin_block(int n) { pulse.begin_block(n); }
    FD_HD void bind(Ctx& a) { pulse.bind(a); }
    FD_HD bool tripped() const { return pulse.tripped(); }
    template <int PH> FD_HD void step(const float* in, float* out) { pulse.template step<PH>(in, out); }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) { pulse.template step2<PH>(in, out); }
};

// Shaper<ShapeFn<S>>  shape.rs:35-42, 205-247 (ID 42): shape_fn(|x| ..) with the closure as a functor type
//   static float f(float x)
// The Shape trait's default `simd` applies `shape` lane by lane (shape.rs:16-18), so every phase is the same arithmetic.
template <class FN>
struct ShaperFn {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 42;
    FD_STATELESS_LEAF
    template <int PH> FD_HD void step(const float* in, float* out) { out[0] = FN::f(in[0]); }
    FD_STEP2_VIA_STEP
};

// Declick<f32>  dynamics.rs:245-313 (ID 23): smooth5 fade-in over `duration` seconds from time zero.
// tick re-derives the fade phase from t every sample (:278-287); process accumulates phase_d = sample_duration /
// duration inside the block and advances t by size * sample_duration in one multiplication (:289-307).
struct Declick {
    static constexpr int IN = 1, OUT = 1, RINGS = 0;
    static constexpr uint64_t ID = 23;
    float t, duration, sd;
    float phase, phase_d, end_time;  // block transients (process path)
    int blk_i, blk_size, end_index;
    bool fading;
    template <class V> FD_HD void visit(V& v) {
        v.f(duration, PARAM, "duration");
        v.f(sd, COEF, "sample_duration");
        v.f(t, STATE, "t");
    }
    FD_HD void bind(Ctx&) {}
    FD_HD void init() { t = 0.0f; duration = 0.010f; sd = 0.0f; fading = false; blk_i = blk_size = end_index = 0; }
    FD_HD void update(double sr) { sd = (float)(1.0 / sr); }
    FD_HD void reset() { t = 0.0f; }
    FD_HD uint64_t ping(bool, uint64_t h) { return atto(h, ID); }
    FD_HD bool tripped() const { return false; }
    FD_HD void end_simd() {}
    static FD_HD float smooth5(float x) { return ((x * 6.0f - 15.0f) * x + 10.0f) * x * x * x; }  // math.rs:418-420
    FD_HD void begin_block(int size) {
        blk_i = 0;
        blk_size = size;
        fading = t < duration;
        if (fading) {
            phase = (t - 0.0f) / (duration - 0.0f);  // delerp math.rs:218-220
            phase_d = sd / duration;
            end_time = t + (float)(long long)size * sd;
            end_index = duration < end_time ? (int)(long long)__builtin_ceilf((duration - t) / sd) : size;
        }
    }
    template <int PH> FD_HD void step(const float* in, float* out) {
        if (PH == PH_TICK) {
            if (t < duration) {
                const float ph = (t - 0.0f) / (duration - 0.0f);
                const float value = smooth5(ph);
                t += sd;
                out[0] = in[0] * value;
            } else {
                out[0] = in[0];
            }
        } else {
            float x = in[0];
            if (fading) {
                if (blk_i < end_index) {
                    x *= smooth5(phase);
                    phase += phase_d;
                }
                blk_i++;
                if (blk_i == blk_size) t = end_time;  // :305, once per block
            }
            out[0] = x;
        }
    }
    FD_STEP2_VIA_STEP
};

#define FD_PAIR_COMBINATOR                                                                                  \
    X x;                                                                                                    \
    Y y;                                                                                                    \
    template <class V> FD_HD void visit(V& v) {                                                             \
        v.enter(0); x.visit(v); v.leave();                                                                  \
        v.enter(1); y.visit(v); v.leave();                                                                  \
    }                                                                                                       \
    FD_HD void init() { x.init(); y.init(); }                                                               \
    FD_HD void update(double sr) { x.update(sr); y.update(sr); }                                            \
    FD_HD void reset() { x.reset(); y.reset(); }                                                            \
    FD_HD uint64_t ping(bool probe, uint64_t h) { return y.ping(probe, x.ping(probe, atto(h, ID))); }       \
    FD_HD void end_simd() { x.end_simd(); y.end_simd(); }                                                   \
    FD_HD void begin_block(int n) { x.begin_block(n); y.begin_block(n); }                                   \
    FD_HD void bind(Ctx& a) { x.bind(a); y.bind(a); }                                                       \
    FD_HD bool tripped() const { return x.tripped() || y.tripped(); }

// Branch<X, Y>  audionode.rs:1653-1792 (ID 8): both read the same input, outputs side by side
template <class X, class Y>
struct Branch {
    static_assert(X::IN == Y::IN, "Branch arity mismatch");
    static constexpr int IN = X::IN, OUT = X::OUT + Y::OUT, RINGS = X::RINGS + Y::RINGS;
    static constexpr uint64_t ID = 8;
    FD_PAIR_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        x.template step<PH>(in, out);
        y.template step<PH>(in, out + X::OUT);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        x.template step2<PH>(in, out);
        y.template step2<PH>(in, out + X::OUT);
    }
};

// Bus<X, Y>  audionode.rs:1796-1947 (ID 10): both read the same input, outputs added (x + y in tick :1861-1865 and
// in process :1867-1876)
template <class X, class Y>
struct Bus {
    static_assert(X::IN == Y::IN && X::OUT == Y::OUT, "Bus arity mismatch");
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = X::RINGS + Y::RINGS;
    static constexpr uint64_t ID = 10;
    FD_PAIR_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT > 0 ? OUT : 1];
        x.template step<PH>(in, out);
        y.template step<PH>(in, t);
        for (int i = 0; i < OUT; i++) out[i] = out[i] + t[i];
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        v2f t[OUT > 0 ? OUT : 1];
        x.template step2<PH>(in, out);
        y.template step2<PH>(in, t);
        for (int i = 0; i < OUT; i++) out[i] = out[i] + t[i];
    }
};

// Thru<X>  audionode.rs:1951-2060 (ID 12): X's outputs, then the inputs X has no output for; when X has MORE outputs
// than inputs the surplus outputs are cut (:1990-1999)
template <class X>
struct Thru {
    static constexpr int IN = X::IN, OUT = X::IN, RINGS = X::RINGS;
    static constexpr uint64_t ID = 12;
    X x;
    template <class V> FD_HD void visit(V& v) { v.enter(0); x.visit(v); v.leave(); }
    FD_HD void init() { x.init(); }
    FD_HD void update(double sr) { x.update(sr); }
    FD_HD void reset() { x.reset(); }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return x.ping(probe, atto(h, ID)); }  // :2026-2028
    FD_HD void end_simd() { x.end_simd(); }
    FD_HD void begin_block(int n) { x.begin_block(n); }
    FD_HD void bind(Ctx& a) { x.bind(a); }
    FD_HD bool tripped() const { return x.tripped(); }
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[X::OUT > 0 ? X::OUT : 1];
        x.template step<PH>(in, t);
        for (int i = 0; i < OUT; i++) out[i] = i < X::OUT ? t[i] : in[i];
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        v2f t[X::OUT > 0 ? X::OUT : 1];
        x.template step2<PH>(in, t);
        for (int i = 0; i < OUT; i++) out[i] = i < X::OUT ? t[i] : in[i];
    }
};

// Feedback<N, X, U>  feedback.rs:71-190 (ID 11) and Feedback2<N, X, Y, U> :193-330 (ID 66): the enclosed node's output
// (through Y for Feedback2), passed through the feedback operator U, is added to the next sample's input.
// U = FrameId (feedback(), feedback2()) or FrameHadamard (fdn(), fdn2(); :35-57: in-place butterflies h = 1, 2, 4, ..
// then a scale by (1 / sqrt(N)) as f32).  process is the tick loop (:136-146, :277-287): the enclosed nodes always run
// their tick arithmetic.  Feedback::new calls prevent_denormals() (:96, denormal.rs:18), so a graph with a Feedback
// node is compiled with f32 denormals flushed (fd_jit.hip), like the FDN reverb kernel.
struct FbId {
    template <int N> static FD_HD void f(float*) {}
};
struct FbHadamard {
    template <int N> static FD_HD void f(float* o) {
        static_assert((N & (N - 1)) == 0, "fdn: the channel count must be a power of two");
        _Pragma("unroll") for (int h = 1; h < N; h *= 2)
            _Pragma("unroll") for (int i = 0; i < N; i += h * 2)
                _Pragma("unroll") for (int j = i; j < i + h; j++) {
                    const float a = o[j], b = o[j + h];
                    o[j] = a + b;
                    o[j + h] = a - b;
                }
        const float z = (float)(1.0 / __builtin_sqrt((double)N));
        _Pragma("unroll") for (int i = 0; i < N; i++) o[i] = o[i] * z;
    }
};

template <class X, class U>
struct Feedback {
    static_assert(X::IN == X::OUT, "feedback: the enclosed node needs as many outputs as inputs");
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = X::RINGS;
    static constexpr uint64_t ID = 11;
    static constexpr bool FLUSHES_DENORMALS = true;
    X x;
    float value[OUT];
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(value[i], STATE, "value", i);
    }
    FD_HD void init() { x.init(); for (int i = 0; i < OUT; i++) value[i] = 0.0f; }
    FD_HD void update(double sr) { x.update(sr); }
    FD_HD void reset() { x.reset(); for (int i = 0; i < OUT; i++) value[i] = 0.0f; }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return x.ping(probe, atto(h, ID)); }  // :152-154
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD void bind(Ctx& a) { x.bind(a); }
    FD_HD bool tripped() const { return false; }
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT];
        for (int i = 0; i < OUT; i++) t[i] = in[i] + value[i];
        x.template step<PH_TICK>(t, out);
        for (int i = 0; i < OUT; i++) value[i] = out[i];
        U::template f<OUT>(value);
    }
    FD_STEP2_VIA_STEP
};

template <class X, class Y, class U>
struct Feedback2 {
    static_assert(X::IN == X::OUT && Y::IN == X::OUT && Y::OUT == X::OUT, "feedback2: arity mismatch");
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = X::RINGS + Y::RINGS;
    static constexpr uint64_t ID = 66;
    static constexpr bool FLUSHES_DENORMALS = true;
    X x;
    Y y;
    float value[OUT];
    template <class V> FD_HD void visit(V& v) {
        v.enter(0); x.visit(v); v.leave();
        v.enter(1); y.visit(v); v.leave();
        _Pragma("unroll") for (int i = 0; i < OUT; i++) v.fi(value[i], STATE, "value", i);
    }
    FD_HD void init() { x.init(); y.init(); for (int i = 0; i < OUT; i++) value[i] = 0.0f; }
    FD_HD void update(double sr) { x.update(sr); y.update(sr); }
    FD_HD void reset() { x.reset(); y.reset(); for (int i = 0; i < OUT; i++) value[i] = 0.0f; }
    FD_HD uint64_t ping(bool probe, uint64_t h) { return y.ping(probe, x.ping(probe, atto(h, ID))); }  // :293-295
    FD_HD void end_simd() {}
    FD_HD void begin_block(int) {}
    FD_HD void bind(Ctx& a) { x.bind(a); y.bind(a); }
    FD_HD bool tripped() const { return false; }
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT];
        for (int i = 0; i < OUT; i++) t[i] = in[i] + value[i];
        x.template step<PH_TICK>(t, out);
        y.template step<PH_TICK>(out, value);
        U::template f<OUT>(value);
    }
    FD_STEP2_VIA_STEP
};

// N-fold combinators over N nodes of ONE type (the closure forms busi/stacki/branchi/sumi/pipei of the prelude).
#define FD_MULTI_COMBINATOR                                                                                 \
    X x[N];                                                                                                 \
    template <class V> FD_HD void visit(V& v) {                                                             \
        for (int i = 0; i < N; i++) { v.enter(i); x[i].visit(v); v.leave(); }                               \
    }                                                                                                       \
    FD_HD void init() { for (int i = 0; i < N; i++) x[i].init(); }                                          \
    FD_HD void update(double sr) { for (int i = 0; i < N; i++) x[i].update(sr); }                           \
    FD_HD void reset() { for (int i = 0; i < N; i++) x[i].reset(); }                                        \
    FD_HD uint64_t ping(bool probe, uint64_t h) {                                                           \
        h = atto(h, ID);                                                                                    \
        for (int i = 0; i < N; i++) h = x[i].ping(probe, h);                                                \
        return h;                                                                                           \
    }                                                                                                       \
    FD_HD void end_simd() { for (int i = 0; i < N; i++) x[i].end_simd(); }                                  \
    FD_HD void begin_block(int n) { for (int i = 0; i < N; i++) x[i].begin_block(n); }                      \
    FD_HD void bind(Ctx& a) { for (int i = 0; i < N; i++) x[i].bind(a); }                                   \
    FD_HD bool tripped() const {                                                                            \
        bool t = false;                                                                                     \
        for (int i = 0; i < N; i++) t = t || x[i].tripped();                                                \
        return t;                                                                                           \
    }

// MultiBus<N, X>  audionode.rs:2065-2207 (ID 28).  tick folds from a zero frame, (0 + x0) + x1 .. (:2117-2121);
// process starts from x0's output (:2123-2134) -- they differ only in the sign of a zero.
template <int N, class X>
struct MultiBus {
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = N * X::RINGS;
    static constexpr uint64_t ID = 28;
    FD_MULTI_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT > 0 ? OUT : 1];
        x[0].template step<PH>(in, out);
        if (PH == PH_TICK)
            for (int c = 0; c < OUT; c++) out[c] = 0.0f + out[c];
        _Pragma("unroll") for (int i = 1; i < N; i++) {
            x[i].template step<PH>(in, t);
            for (int c = 0; c < OUT; c++) out[c] = out[c] + t[c];
        }
    }
    FD_STEP2_VIA_STEP
};

// MultiStack<N, X>  audionode.rs:2211-2362 (ID 30)
template <int N, class X>
struct MultiStack {
    static constexpr int IN = N * X::IN, OUT = N * X::OUT, RINGS = N * X::RINGS;
    static constexpr uint64_t ID = 30;
    FD_MULTI_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        _Pragma("unroll") for (int i = 0; i < N; i++) x[i].template step<PH>(in + i * X::IN, out + i * X::OUT);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        _Pragma("unroll") for (int i = 0; i < N; i++) x[i].template step2<PH>(in + i * X::IN, out + i * X::OUT);
    }
};

// MultiBranch<N, X>  audionode.rs:2532-2669 (ID 33)
template <int N, class X>
struct MultiBranch {
    static constexpr int IN = X::IN, OUT = N * X::OUT, RINGS = N * X::RINGS;
    static constexpr uint64_t ID = 33;
    FD_MULTI_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        _Pragma("unroll") for (int i = 0; i < N; i++) x[i].template step<PH>(in, out + i * X::OUT);
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        _Pragma("unroll") for (int i = 0; i < N; i++) x[i].template step2<PH>(in, out + i * X::OUT);
    }
};

// Reduce<N, X, B>  audionode.rs:2366-2528 (ID 31): own inputs per node, outputs folded left with the binary op
template <int N, class X, class OP>
struct Reduce {
    static constexpr int IN = N * X::IN, OUT = X::OUT, RINGS = N * X::RINGS;
    static constexpr uint64_t ID = 31;
    FD_MULTI_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT > 0 ? OUT : 1];
        x[0].template step<PH>(in, out);
        _Pragma("unroll") for (int i = 1; i < N; i++) {
            x[i].template step<PH>(in + i * X::IN, t);
            for (int c = 0; c < OUT; c++) out[c] = OP::f(out[c], t[c]);
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        v2f t[OUT > 0 ? OUT : 1];
        x[0].template step2<PH>(in, out);
        _Pragma("unroll") for (int i = 1; i < N; i++) {
            x[i].template step2<PH>(in + i * X::IN, t);
            for (int c = 0; c < OUT; c++) out[c] = OP::f(out[c], t[c]);
        }
    }
};

// Chain<N, X> of the reference (audionode.rs:2673-2804, ID 32): N nodes of one type in series (pipei)
template <int N, class X>
struct PipeN {
    static_assert(X::IN == X::OUT, "pipei needs as many outputs as inputs");
    static constexpr int IN = X::IN, OUT = X::OUT, RINGS = N * X::RINGS;
    static constexpr uint64_t ID = 32;
    FD_MULTI_COMBINATOR
    template <int PH> FD_HD void step(const float* in, float* out) {
        float t[OUT > 0 ? OUT : 1];
        x[0].template step<PH>(in, out);
        _Pragma("unroll") for (int i = 1; i < N; i++) {
            for (int c = 0; c < OUT; c++) t[c] = out[c];
            x[i].template step<PH>(t, out);
        }
    }
    template <int PH> FD_HD void step2(const v2f* in, v2f* out) {
        v2f t[OUT > 0 ? OUT : 1];
        x[0].template step2<PH>(in, out);
        _Pragma("unroll") for (int i = 1; i < N; i++) {
            for (int c = 0; c < OUT; c++) t[c] = out[c];
            x[i].template step2<PH>(t, out);
        }
    }
};

}  // namespace fd
