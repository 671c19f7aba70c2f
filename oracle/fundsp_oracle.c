/*
 * oracle/fundsp_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the FunDSP hot path:
 * AudioNode::tick / AudioNode::process of the leaf DSP nodes plus the minimal combinator / executor glue the
 * BASELINE configs need.  It is the checker for the HIP engine; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product library (fundsp_amd/csrc) never includes or links it.
 *
 * Reference: SamiPerttu/fundsp 0.23.0 mounted at /root/reference.  Every function cites the reference
 * file:line it restates.  The reference cannot be compiled in this image (no rustc/cargo), so parity is
 * pinned through the reference's own test invariants re-run against this file (tests/test_oracle_*.py):
 * closed-form frequency responses (tests/test_flow.rs:18-80), tick == process to 1e-4
 * (tests/test_basic.rs:21-92), exact delay/constant identities (tests/test_basic.rs:365-378,520-529).
 * Transcendentals from the un-vendored `libm`/`wide` crates: see o_math.h ("parity unpinned" at bit level).
 *
 * Numeric contract: prelude32 (F = f32); no FMA contraction, no reassociation (build: -ffp-contract=off).
 * Buffers are planar [channel][64] f32 exactly like BufferRef/BufferMut (src/buffer.rs:8-12).
 */
#include "fundsp_oracle.h"
#include "o_math.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
#endif

#define DEFAULT_SR 44100.0 /* src/lib.rs:42 */
#define MAXB 64            /* MAX_BUFFER_SIZE src/lib.rs:48 */
#define SIMD_N 8
#define O_MAX_FIR 16
#define O_MAX_CH 64
#define O_MAX_ENV 8

static const float F32_PI = 3.14159274101257324f;  /* core::f32::consts::PI */
static const float F32_TAU = 6.28318548202514648f; /* core::f32::consts::TAU */
static const float F32_SQRT_2 = 1.41421353816986084f;

typedef struct { float a1, a2, a3, m0, m1, m2; } svf_coefs;
typedef struct { float a1, a2, b0, b1, b2; } bq_coefs;

struct onode {
    int type, nin, nout;
    uint64_t id;
    onode *x, *y;
    onode *aux;   /* Limiter: its AFollow (not part of the graph recursion) */
    onode *pre[4]; /* Reverb: input diffusers, outside reset / set_sample_rate (reverb.rs:211-238) */
    float rv3_feedback, rv3_a;
    onode **kids; /* O_MULTI: the N nodes of MultiBus / MultiStack / MultiBranch / Reduce / Chain */
    int nkids, multi;
    int jm, jn;   /* Split / Join: M channels, N branches */
    int ftz;      /* this graph contains a Feedback node: rendered with MXCSR FTZ + DAZ (see o_feedback) */
    int hadamard; /* Feedback: U = FrameHadamard */
    float fb_value[O_MAX_CH];
    o_map_fn map_fn;
    void *map_ctx;
    float *tmp2;  /* second scratch of Chain::process (audionode.rs:2729) */
    float *tmp; /* [x->nout][64] scratch for Pipe / Binop (BufferArray::uninitialized, audionode.rs:1446) */
    int op;
    float scalar;
    /* leaf state (one struct for all leaves keeps the oracle small; each leaf uses its own fields) */
    struct {
        float value[O_MAX_CH]; /* Constant */
        /* Sine (oscillator.rs:21-26) */
        float phase, sample_duration;
        uint64_t hash;
        int has_initial_phase;
        float initial_phase;
        /* Noise (noise.rs:173-177) */
        uint32_t nstate;
        int has_seed;
        uint64_t seed;
        /* Svf / FixedSvf (svf.rs:748-759, 861-871) */
        int mode;
        float sr, cutoff, q, gain;
        svf_coefs sc;
        float ic1eq, ic2eq;
        /* Biquad family (biquad.rs:136-143) ; BiquadBank uses the [8] arrays (biquad_bank.rs:14-24) */
        bq_coefs bc;
        float x1, x2, y1, y2;
        double sr64;
        float center;
        bq_coefs bank_c[8];
        float bx1[8], bx2[8], by1[8], by2[8];
        /* Moog (moog.rs:17-33) */
        float rez, p, k, s0, s1, s2, s3, px, ps0, ps1, ps2;
        /* Fir (fir.rs:14-18) */
        int fir_n;
        float w[O_MAX_FIR], v[O_MAX_FIR];
        /* Tick (delay.rs:19-22) */
        float tickbuf[O_MAX_CH];
        /* Delay (delay.rs:72-78) */
        float *dbuf;
        size_t dlen, di, time_in_samples;
        double dtime, dsr;
        /* WaveSynth (wavetable.rs:249-264) */
        const owavetable *wt;
        size_t table_hint;
        float ws_sr;
        /* EnvelopeIn<f32, adsr_live closure, U1, f32> (envelope.rs:185-218, adsr.rs:21-57) */
        float et, et0, et1, ev0, ev1, ev, evd, einterval, esd;
        uint64_t et_hash;
        int attacked;
        float attack_start, release_start, adsr_a, adsr_d, adsr_s, adsr_r;
        /* Panner (pan.rs:26-30) */
        float left_weight, right_weight;
        /* Tap / TapLinear (delay.rs:148-161, 386-397), AllNest (delay.rs:294-303) */
        int tap_linear;
        float tap_min, tap_max, tap_min_c, tap_max_c, tap_sr, eta, zz;
        float *tbuf;
        size_t tlen, ti;
        /* one-pole family (filter.rs:19-431), Pinkpass, Morph */
        int op_kind;
        float op_coeff, op_x1, op_y1, pink[7], morph;
        /* Shape x2 (shape.rs), PhaseOsc kind, Chaos, nonlinear biquad (biquad.rs:494-920) */
        struct { int kind; float p0, p1, smoothing, state, ts; } sh[2];  /* ts: Adaptive's timescale */
        int osc_kind, lorenz, nl_dirty, nl_mode;
        float cx, cy, cz;
        float ns1, ns2;
        /* Resample<X> (resample.rs:210-220): 128-sample ring per output channel */
        float *rs_buf;
        double rs_consumer;
        size_t rs_producer;
        /* Oversampler<X> (oversample.rs:66-80): 128-sample input / output rings per channel */
        float *os_inv, *os_outv; /* [channel][128] */
        size_t os_in_i, os_out_i;
        /* Pluck (oscillator.rs:215-226): Fir<U3> damping = w/v above, Allpole tuning = op_* below; the excitation is the
         * stream `Rnd::from_u64(hash).f32_in(-1, 1)` of funutd (crate source absent) and is supplied by the caller */
        float pl_freq, pl_gain;
        float *pl_raw, *pl_line;
        size_t pl_raw_n, pl_len, pl_pos;
        int pl_init;
        double pl_sr;
        /* Envelope<f32, E, R> (envelope.rs:17-49): the closure is a C callback; time fields et/et0/et1/einterval/esd/et_hash shared with EnvelopeIn */
        o_env_fn env_fn;
        o_envin_fn envin_fn;
        void *env_ctx;
        float env_v0[O_MAX_ENV], env_v1[O_MAX_ENV], env_val[O_MAX_ENV], env_d[O_MAX_ENV];
        int ps_ready; /* PhaseSynth::phase_ready */
        /* WavePlayer (wave.rs:739-746) */
        const float *wp_data;
        size_t wp_length, wp_index, wp_start, wp_end;
        long wp_loop;
        int wp_channel;
        /* Hold (noise.rs:242-250) */
        double *hd_draws;
        size_t hd_n, hd_pos;
        double hd_sd, hd_t, hd_next;
        float hd_var, hd_hold;
        /* MeterState (dynamics.rs:336-339) */
        int mt_mode, mt_monitor;
        double mt_timescale;
        float mt_smoothing, mt_state;
        /* Limiter (dynamics.rs:125-139) */
        double lm_lookahead, lm_sr;
        size_t lm_length, lm_leaf, lm_index, lm_fill;
        float *lm_tree, *lm_buf;
        /* Declick (dynamics.rs:245-250) */
        float dc_t, dc_duration, dc_sd;
        /* Dsf (oscillator.rs:121-129) */
        float dsf_roughness, dsf_spacing;
        /* Rez (rez.rs:11-21), Follow / AFollow (follow.rs:31-43,137-152), Mls (noise.rs:14-20,103-107) */
        float rz_buf0, rz_buf1, rz_f, rz_fb, rz_bandpass;
        float fo_v1, fo_v2, fo_v3, fo_coeff, fo_coeff_now, fo_rcoeff, fo_rcoeff_now, fo_time, fo_rtime;
        uint32_t mls_n, mls_s;
        /* reverb_stereo: 32 x (Delay >> Fir<U3>) inside Feedback<U32,_,FrameHadamard> (prelude.rs:1732-1762) */
        double rv_room, rv_time, rv_damping, rv_sr;
        float *rv_buf[32];
        size_t rv_len[32], rv_i[32];
        float rv_v[32][3], rv_w[3], rv_value[32], rv_wl[32], rv_wr[32];
    } s;
};

struct owavetable {
    int n;
    float *pitch;
    int *len;
    float **tab;
};

/* ------------------------------------------------------------------------------------------------------ */
/* helpers                                                                                                */
/* ------------------------------------------------------------------------------------------------------ */
static void meter_set_sr(onode *n, double sr);
static void limiter_set_sr(onode *n, double sr);
static onode *o_new(int type, int nin, int nout, uint64_t id) {
    onode *n = (onode *)calloc(1, sizeof(onode));
    n->type = type;
    n->nin = nin;
    n->nout = nout;
    n->id = id;
    return n;
}

int o_inputs(const onode *n) { return n->nin; }
int o_outputs(const onode *n) { return n->nout; }

void o_free(onode *n) {
    if (!n) return;
    if (n->type == O_REVERB_STEREO)
        for (int i = 0; i < 32; i++) free(n->s.rv_buf[i]);
    o_free(n->x);
    o_free(n->y);
    o_free(n->aux);
    for (int i = 0; i < 4; i++) o_free(n->pre[i]);
    free(n->s.hd_draws);
    free(n->s.lm_tree);
    free(n->s.lm_buf);
    for (int i = 0; i < n->nkids; i++) o_free(n->kids[i]);
    free(n->kids);
    free(n->tmp2);
    free(n->tmp);
    free(n->s.dbuf);
    free(n->s.tbuf);
    free(n->s.pl_raw);
    free(n->s.pl_line);
    free(n->s.rs_buf);
    free(n->s.os_inv);
    free(n->s.os_outv);
    free(n);
}

/* exported scalar math for the math unit tests */
float o_math_sinf(float x) { return o_sinf(x); }
float o_math_cosf(float x) { return o_cosf(x); }
float o_math_tanf(float x) { return o_tanf(x); }
float o_math_tanhf(float x) { return o_tanhf(x); }
float o_math_expf(float x) { return o_expf(x); }
float o_math_powf(float x, float y) { return o_powf(x, y); }
float o_math_expm1f(float x) { return o_expm1f(x); }
float o_math_wide_sinf(float x) { return o_wide_sinf(x); }
float o_math_atanf(float x) { return o_atanf(x); }
float o_math_wide_atanf(float x) { return o_wide_atanf(x); }
/* Adaptive::set_sample_rate shape.rs:197-200: pow(0.5, 1.0 / (timescale.to_f64() * sample_rate)).to_f32() as f64 value */
double o_adaptive_smoothing(float timescale, double sample_rate) { return pow(0.5, 1.0 / ((double)timescale * sample_rate)); }
double o_math_rnd1(uint64_t x) { return o_rnd1(x); }
uint64_t o_math_hash1(uint64_t x) { return o_hash1(x); }
uint64_t o_math_atto(uint64_t state, uint64_t data) { return o_atto(state, data); }
uint32_t o_math_hash32x(uint32_t x) { return o_hash32x(x); }

/* ------------------------------------------------------------------------------------------------------ */
/* coefficient constructors                                                                               */
/* ------------------------------------------------------------------------------------------------------ */

/* SvfCoefs::{lowpass..highshelf} src/svf.rs:28-221.  g = tan((PI*cutoff)/sample_rate) in f32. */
static svf_coefs svf_make(int mode, float sr, float cutoff, float q, float gain) {
    svf_coefs c;
    float g, k, a;
    switch (mode) {
    case O_SVF_BELL: /* svf.rs:155-174 */
        a = sqrtf(gain);
        g = o_tanf(F32_PI * cutoff / sr);
        k = 1.0f / (q * a);
        break;
    case O_SVF_LOWSHELF: /* svf.rs:178-197 */
        a = sqrtf(gain);
        g = o_tanf(F32_PI * cutoff / sr) / sqrtf(a);
        k = 1.0f / q;
        break;
    case O_SVF_HIGHSHELF: /* svf.rs:201-220 */
        a = sqrtf(gain);
        g = o_tanf(F32_PI * cutoff / sr) * sqrtf(a);
        k = 1.0f / q;
        break;
    default: /* svf.rs:28-151 */
        a = 0.0f;
        g = o_tanf(F32_PI * cutoff / sr);
        k = 1.0f / q;
        break;
    }
    c.a1 = 1.0f / (1.0f + g * (g + k));
    c.a2 = g * c.a1;
    c.a3 = g * c.a2;
    switch (mode) {
    case O_SVF_LOWPASS: c.m0 = 0.0f; c.m1 = 0.0f; c.m2 = 1.0f; break;
    case O_SVF_HIGHPASS: c.m0 = 1.0f; c.m1 = -k; c.m2 = -1.0f; break;
    case O_SVF_BANDPASS: c.m0 = 0.0f; c.m1 = 1.0f; c.m2 = 0.0f; break;
    case O_SVF_NOTCH: c.m0 = 1.0f; c.m1 = -k; c.m2 = 0.0f; break;
    case O_SVF_PEAK: c.m0 = 1.0f; c.m1 = -k; c.m2 = -2.0f; break;
    case O_SVF_ALLPASS: c.m0 = 1.0f; c.m1 = -2.0f * k; c.m2 = 0.0f; break;
    case O_SVF_BELL: c.m0 = 1.0f; c.m1 = k * (a * a - 1.0f); c.m2 = 0.0f; break;
    case O_SVF_LOWSHELF: c.m0 = 1.0f; c.m1 = k * (a - 1.0f); c.m2 = a * a - 1.0f; break;
    default: /* HIGHSHELF */ c.m0 = a * a; c.m1 = k * (1.0f - a) * a; c.m2 = 1.0f - a * a; break;
    }
    return c;
}

/* BiquadCoefs::butter_lowpass src/biquad.rs:30-41 */
static bq_coefs bq_butter_lowpass(float sr, float cutoff) {
    bq_coefs c;
    float f = o_tanf(cutoff * F32_PI / sr);
    float a0r = 1.0f / (1.0f + F32_SQRT_2 * f + f * f);
    c.a1 = (2.0f * f * f - 2.0f) * a0r;
    c.a2 = (1.0f - F32_SQRT_2 * f + f * f) * a0r;
    c.b0 = f * f * a0r;
    c.b1 = 2.0f * c.b0;
    c.b2 = c.b0;
    return c;
}
/* BiquadCoefs::resonator src/biquad.rs:45-54 */
static bq_coefs bq_resonator(float sr, float center, float q) {
    bq_coefs c;
    float r = o_expf(-F32_PI * center / (q * sr));
    c.a1 = -2.0f * r * o_cosf(F32_TAU * center / sr);
    c.a2 = r * r;
    c.b0 = sqrtf(1.0f - r * r) * 0.5f;
    c.b1 = 0.0f;
    c.b2 = -c.b0;
    return c;
}
/* BiquadCoefs::lowpass src/biquad.rs:58-71 */
static bq_coefs bq_lowpass(float sr, float cutoff, float q) {
    bq_coefs c;
    float omega = F32_TAU * cutoff / sr;
    float alpha = o_sinf(omega) / (2.0f * q);
    float beta = o_cosf(omega);
    float a0r = 1.0f / (1.0f + alpha);
    c.a1 = -2.0f * beta * a0r;
    c.a2 = (1.0f - alpha) * a0r;
    c.b1 = (1.0f - beta) * a0r;
    c.b0 = c.b1 * 0.5f;
    c.b2 = c.b0;
    return c;
}
/* BiquadCoefs::highpass src/biquad.rs:75-88 */
static bq_coefs bq_highpass(float sr, float cutoff, float q) {
    bq_coefs c;
    float omega = F32_TAU * cutoff / sr;
    float alpha = o_sinf(omega) / (2.0f * q);
    float beta = o_cosf(omega);
    float a0r = 1.0f / (1.0f + alpha);
    c.a1 = -2.0f * beta * a0r;
    c.a2 = (1.0f - alpha) * a0r;
    c.b0 = (1.0f + beta) * 0.5f * a0r;
    c.b1 = (-1.0f - beta) * a0r;
    c.b2 = c.b0;
    return c;
}
/* BiquadCoefs::bell src/biquad.rs:92-106 */
static bq_coefs bq_bell(float sr, float center, float q, float gain) {
    bq_coefs c;
    float omega = F32_TAU * center / sr;
    float alpha = o_sinf(omega) / (2.0f * q);
    float beta = o_cosf(omega);
    float a = sqrtf(gain);
    float a0r = 1.0f / (1.0f + alpha / a);
    c.a1 = -2.0f * beta * a0r;
    c.a2 = (1.0f - alpha / a) * a0r;
    c.b0 = (1.0f + alpha * a) * a0r;
    c.b1 = c.a1;
    c.b2 = (1.0f - alpha * a) * a0r;
    return c;
}

static bq_coefs bq_by_mode(int mode, float sr, float center, float q, float gain) { /* BiquadMode::update biquad.rs:404-490 */
    switch (mode) {
    case O_BQ_RESONATOR: return bq_resonator(sr, center, q);
    case O_BQ_LOWPASS: return bq_lowpass(sr, center, q);
    case O_BQ_HIGHPASS: return bq_highpass(sr, center, q);
    default: return bq_bell(sr, center, q, gain);
    }
}

void o_biquad_coefs(int kind, float sr, float f, float q, float gain, float *out5) {
    bq_coefs c;
    switch (kind) {
    case O_BQ_BUTTER: c = bq_butter_lowpass(sr, f); break;
    case O_BQ_RESONATOR: c = bq_resonator(sr, f, q); break;
    case O_BQ_LOWPASS: c = bq_lowpass(sr, f, q); break;
    case O_BQ_HIGHPASS: c = bq_highpass(sr, f, q); break;
    default: c = bq_bell(sr, f, q, gain); break;
    }
    out5[0] = c.a1; out5[1] = c.a2; out5[2] = c.b0; out5[3] = c.b1; out5[4] = c.b2;
}

void o_svf_coefs(int mode, float sr, float cutoff, float q, float gain, float *out6) {
    svf_coefs c = svf_make(mode, sr, cutoff, q, gain);
    out6[0] = c.a1; out6[1] = c.a2; out6[2] = c.a3; out6[3] = c.m0; out6[4] = c.m1; out6[5] = c.m2;
}

/* Moog::set_cutoff_q src/moog.rs:48-57 */
static void moog_set_cutoff_q(onode *n, float cutoff, float q) {
    n->s.cutoff = cutoff;
    n->s.q = q;
    float c = 2.0f * cutoff / n->s.sr;
    n->s.p = c * (1.8f - 0.8f * c);
    n->s.k = 2.0f * o_sinf(c * F32_PI * 0.5f) - 1.0f;
    float t1 = (1.0f - n->s.p) * 1.386249f;
    float t2 = 12.0f + t1 * t1;
    n->s.rez = q * (t2 + 6.0f * t1) / (t2 - 6.0f * t1);
}

void o_moog_coefs(float sr, float cutoff, float q, float *out3) {
    onode tmp;
    memset(&tmp, 0, sizeof tmp);
    tmp.s.sr = sr;
    moog_set_cutoff_q(&tmp, cutoff, q);
    out3[0] = tmp.s.rez; out3[1] = tmp.s.p; out3[2] = tmp.s.k;
}

/* ---- shapes (shape.rs:35-201) ---- */
static float smooth9f(float x);
static inline float rs_clampf(float lo, float hi, float x) { /* math.rs:130-132 */
    x = x > lo ? x : lo;
    return x < hi ? x : hi;
}
static float shape_scalar(onode *n, int which, float input) { /* Shape::shape */
    float p0 = n->s.sh[which].p0, p1 = n->s.sh[which].p1;
    int kind = n->s.sh[which].kind;
    if (kind >= O_SH_ADAPTIVE_TANH) { /* Adaptive<S> shape.rs:185-192: level estimate, then the inner shape on input / sqrt(level) */
        float sm = n->s.sh[which].smoothing;
        n->s.sh[which].state = sm * n->s.sh[which].state + (1.0f - sm) * (1.0e-6f + input * input);
        input = input / sqrtf(n->s.sh[which].state);
        kind = kind == O_SH_ADAPTIVE_TANH ? O_SH_TANH : kind - O_SH_ADAPTIVE;
    }
    switch (kind) {
    case O_SH_CLIP: return rs_clampf(-1.0f, 1.0f, input * p0);
    case O_SH_CLIPTO: return rs_clampf(p0, p1, input);
    case O_SH_TANH: return o_tanhf(input * p0);
    case O_SH_ATAN: return o_atanf(input * (p0 * F32_PI * 0.5f)) * (2.0f / F32_PI);
    case O_SH_SOFTSIGN: { float x = input * p0; return x / (1.0f + fabsf(x)); }
    case O_SH_CRUSH: return roundf(input * p0) / p0;
    default: { float x = input * p0; float y = floorf(x); return (y + smooth9f(x - y)) / p0; } /* O_SH_SOFTCRUSH */
    }
}
static float shape_simd_lane(onode *n, int which, float input) { /* Shape::simd, one lane */
    float p0 = n->s.sh[which].p0;
    switch (n->s.sh[which].kind) {
    case O_SH_ATAN: return o_wide_atanf(input * (p0 * F32_PI * 0.5f)) * (2.0f / F32_PI);
    case O_SH_SOFTSIGN: return input * p0 / (1.0f + fabsf(input) * p0);
    case O_SH_CRUSH: return nearbyintf(input * p0) / p0;            /* wide round: half to even */
    case O_SH_SOFTCRUSH: {                                           /* Num::floor for f32x8: (x - 0.4999999).round() lib.rs:326 */
        float x = input * p0;
        float y = nearbyintf(x - 0.4999999f);
        return (y + smooth9f(x - y)) / p0;
    }
    default: return shape_scalar(n, which, input);
    }
}
static inline float polyblepf(float t, float dt) { /* oscillator.rs:512-523 */
    if (t < dt) {
        float z = t / dt;
        return z + z - z * z - 1.0f;
    } else if (t > 1.0f - dt) {
        float z = (t - 1.0f) / dt;
        return z + z + z * z + 1.0f;
    }
    return 0.0f;
}
static bq_coefs bq_by_mode(int mode, float sr, float center, float q, float gain);
static void onepole_set(onode *n, float c);
static void rez_set(onode *n, float cutoff, float q);
static void follow_set(onode *n, float t);
static void afollow_set(onode *n, float a, float r);

/* ------------------------------------------------------------------------------------------------------ */
/* reset / set_sample_rate / set_hash / ping                                                              */
/* ------------------------------------------------------------------------------------------------------ */

static void leaf_reset(onode *n) {
    switch (n->type) {
    case O_SINE: /* oscillator.rs:55-60 */
        n->s.phase = n->s.has_initial_phase ? n->s.initial_phase : (float)o_rnd1(n->s.hash);
        break;
    case O_NOISE: { /* noise.rs:192-195 */
        uint64_t h = n->s.has_seed ? n->s.seed : n->s.hash;
        n->s.nstate = (uint32_t)(h ^ (h >> 32));
        break;
    }
    case O_SVF:
    case O_FIXED_SVF: /* svf.rs:818-821, 984-987 */
        n->s.ic1eq = 0.0f;
        n->s.ic2eq = 0.0f;
        break;
    case O_BIQUAD:
    case O_BUTTER_LOWPASS:
    case O_RESONATOR: /* biquad.rs:172-177 */
        n->s.x1 = n->s.x2 = n->s.y1 = n->s.y2 = 0.0f;
        break;
    case O_BIQUAD_BANK: /* biquad_bank.rs:61-66 */
        for (int i = 0; i < 8; i++) n->s.bx1[i] = n->s.bx2[i] = n->s.by1[i] = n->s.by2[i] = 0.0f;
        break;
    case O_MOOG: /* moog.rs:65-74 */
        n->s.s0 = n->s.s1 = n->s.s2 = n->s.s3 = n->s.px = n->s.ps0 = n->s.ps1 = n->s.ps2 = 0.0f;
        break;
    case O_FIR: /* fir.rs:48-50 */
        for (int i = 0; i < O_MAX_FIR; i++) n->s.v[i] = 0.0f;
        break;
    case O_TICK: /* delay.rs:39-41 */
        for (int i = 0; i < O_MAX_CH; i++) n->s.tickbuf[i] = 0.0f;
        break;
    case O_DELAY: /* delay.rs:100-103 */
        n->s.di = 0;
        for (size_t i = 0; i < n->s.dlen; i++) n->s.dbuf[i] = 0.0f;
        break;
    case O_REVERB_STEREO: /* Feedback::reset feedback.rs:123-126 -> Delay / Fir reset */
        for (int i = 0; i < 32; i++) {
            n->s.rv_i[i] = 0;
            for (size_t k = 0; k < n->s.rv_len[i]; k++) n->s.rv_buf[i][k] = 0.0f;
            n->s.rv_v[i][0] = n->s.rv_v[i][1] = n->s.rv_v[i][2] = 0.0f;
            n->s.rv_value[i] = 0.0f;
        }
        break;
    case O_WAVESYNTH: /* wavetable.rs:292-297 */
    case O_DSF:       /* oscillator.rs:161-166 */
    case O_PHASE_OSC: /* oscillator.rs:449-454 etc. */
        n->s.phase = n->s.has_initial_phase ? n->s.initial_phase : (float)o_rnd1(n->s.hash);
        break;
    case O_ONEPOLE: n->s.op_x1 = n->s.op_y1 = 0.0f; break;
    case O_REZ: n->s.rz_buf0 = n->s.rz_buf1 = 0.0f; break; /* rez.rs:57-60 */
    case O_PLUCK: /* oscillator.rs:274-277 */
        for (int i = 0; i < O_MAX_FIR; i++) n->s.v[i] = 0.0f;
        n->s.pl_init = 0;
        break;
    case O_RESAMPLE: /* resample.rs:270-275 (child reset by o_reset; the ring keeps its contents) */
        n->s.rs_consumer = 1.0;
        n->s.rs_producer = 0;
        break;
    case O_OVERSAMPLE: /* oversample.rs:131-135: rings cleared, ring indices kept (child reset by o_reset) */
        memset(n->s.os_inv, 0, (size_t)(n->nin ? n->nin : 1) * 128 * sizeof(float));
        memset(n->s.os_outv, 0, (size_t)n->nout * 128 * sizeof(float));
        break;
    case O_FOLLOW: /* follow.rs:89-94 */
        n->s.fo_v1 = n->s.fo_v2 = n->s.fo_v3 = 0.0f;
        n->s.fo_coeff_now = 1.0f;
        break;
    case O_AFOLLOW: /* follow.rs:209-215 */
        n->s.fo_v1 = n->s.fo_v2 = n->s.fo_v3 = 0.0f;
        n->s.fo_coeff_now = 1.0f;
        n->s.fo_rcoeff_now = 1.0f;
        break;
    case O_MLS: { /* noise.rs:124-127, MlsState::new_with_seed :66-72 */
        uint64_t h = n->s.has_seed ? n->s.seed : n->s.hash;
        uint32_t seed = (uint32_t)(h ^ (h >> 32));
        n->s.mls_s = 1u + seed % ((1u << n->s.mls_n) - 1u);
        break;
    }
    case O_PINKPASS: for (int i = 0; i < 7; i++) n->s.pink[i] = 0.0f; break;
    case O_MORPH: n->s.ic1eq = n->s.ic2eq = 0.0f; break;
    case O_TAP: /* delay.rs:193-196 */
        n->s.ti = 0;
        for (size_t i = 0; i < n->s.tlen; i++) n->s.tbuf[i] = 0.0f;
        break;
    case O_ALLNEST: /* delay.rs:316-319 (child reset by o_reset's recursion) */
        n->s.zz = 0.0f;
        break;
    case O_CHAOS: { /* oscillator.rs:337-341, 396-400: lerp(0.0, 1.0, rnd1(hash) as f32) */
        float t = (float)o_rnd1(n->s.hash);
        n->s.cx = 0.0f * (1.0f - t) + 1.0f * t;
        n->s.cy = 1.0f;
        n->s.cz = 1.0f;
        break;
    }
    case O_SHAPER: /* Adaptive::reset shape.rs:193-196 */
        if (n->s.sh[0].kind >= O_SH_ADAPTIVE_TANH) n->s.sh[0].state = 1.0e-3f;
        break;
    case O_NLBIQUAD: /* biquad.rs:529-533, 750-755 */
        n->s.ns1 = n->s.ns2 = 0.0f;
        for (int i = 0; i < (n->s.nl_dirty ? 2 : 1); i++)
            if (n->s.sh[i].kind >= O_SH_ADAPTIVE_TANH) n->s.sh[i].state = 1.0e-3f;
        break;
    case O_ENVELOPE: { /* envelope.rs:114-122 */
        n->s.et = 0.0f; n->s.et0 = 0.0f; n->s.et1 = 0.0f;
        n->s.et_hash = n->s.hash;
        n->s.env_fn(n->s.et0, n->s.env_v0, n->s.env_ctx);
        for (int i = 0; i < n->nout; i++) n->s.env_v1[i] = n->s.env_v0[i];
        break;
    }
    case O_ENVELOPE_IN: /* envelope.rs:293-298 */
        n->s.et = 0.0f; n->s.et0 = 0.0f; n->s.et1 = 0.0f;
        n->s.et_hash = n->s.hash;
        break;
    case O_ADSR_LIVE: /* envelope.rs:293-298: the closure state (attacked, start times) is NOT reset */
        n->s.et = 0.0f;
        n->s.et0 = 0.0f;
        n->s.et1 = 0.0f;
        n->s.et_hash = n->s.hash;
        break;
    default: break;
    }
}

void o_reset(onode *n) {
    if (n->x) o_reset(n->x); /* Pipe/Stack/Binop/Unop::reset audionode.rs:1430-1433 etc. */
    if (n->y) o_reset(n->y);
    for (int i = 0; i < n->nkids; i++) o_reset(n->kids[i]);
    if (n->type == O_IMPULSE) n->s.value[0] = 1.0f; /* audionode.rs:2860-2862 */
    if (n->type == O_DECLICK) n->s.dc_t = 0.0f;     /* dynamics.rs:268-270 */
    if (n->type == O_PHASESYNTH) n->s.ps_ready = 0; /* wavetable.rs:387-389 */
    if (n->type == O_METER) n->s.mt_state = 0.0f;   /* dynamics.rs:351-353 */
    if (n->type == O_WAVEPLAYER) n->s.wp_index = n->s.wp_start; /* wave.rs:774-776 */
    if (n->type == O_HOLD) { n->s.hd_pos = 0; n->s.hd_t = 0.0; n->s.hd_next = 0.0; } /* noise.rs:281-285: rnd re-seeded */
    if (n->type == O_REVERB3) n->rv3_feedback = 0.0f; /* reverb.rs:223 (the blocks are the kids: reset by the recursion) */
    if (n->type == O_LIMITER) limiter_set_sr(n, n->s.lm_sr); /* dynamics.rs:184-186 */
    if (n->type == O_FEEDBACK) memset(n->fb_value, 0, sizeof n->fb_value); /* feedback.rs:118-121 */
    leaf_reset(n);
}

static void leaf_set_sample_rate(onode *n, double sr) {
    switch (n->type) {
    case O_SINE: /* oscillator.rs:62-64: convert(1.0 / sample_rate) -- f64 divide then cast */
        n->s.sample_duration = (float)(1.0 / sr);
        break;
    case O_SVF:
    case O_FIXED_SVF: /* svf.rs:823-826, 989-992 -> SvfMode::update_frequency -> update (svf.rs:236-239) */
        n->s.sr = (float)sr;
        n->s.sc = svf_make(n->s.mode, n->s.sr, n->s.cutoff, n->s.q, n->s.gain);
        break;
    case O_BIQUAD: /* biquad.rs:179-181: coefficients are NOT recomputed */
    case O_BIQUAD_BANK:
    case O_FIR:
    case O_TICK:
        n->s.sr64 = sr;
        break;
    case O_BUTTER_LOWPASS: /* biquad.rs:263-267 */
        n->s.sr = (float)sr;
        n->s.sr64 = sr;
        n->s.bc = bq_butter_lowpass(n->s.sr, n->s.cutoff);
        break;
    case O_RESONATOR: /* biquad.rs:349-352 */
        n->s.sr = (float)sr;
        n->s.bc = bq_resonator(n->s.sr, n->s.center, n->s.q);
        break;
    case O_MOOG: /* moog.rs:76-79 */
        n->s.sr = (float)sr;
        moog_set_cutoff_q(n, n->s.cutoff, n->s.q);
        break;
    case O_REVERB_STEREO:
        if (n->s.rv_sr != sr) { /* Delay::set_sample_rate delay.rs:105-113 (resize + reset only on change) */
            int d[32];
            n->s.rv_sr = sr;
            o_reverb_stereo_params(n->s.rv_room, n->s.rv_time, n->s.rv_damping, sr, n->s.rv_w, d, n->s.rv_wl, n->s.rv_wr);
            for (int i = 0; i < 32; i++) {
                n->s.rv_len[i] = (size_t)d[i] + 1;
                n->s.rv_buf[i] = (float *)realloc(n->s.rv_buf[i], n->s.rv_len[i] * sizeof(float));
                n->s.rv_i[i] = 0;
                for (size_t k = 0; k < n->s.rv_len[i]; k++) n->s.rv_buf[i][k] = 0.0f;
            }
        }
        break;
    case O_WAVESYNTH: /* wavetable.rs:299-302: f32 reciprocal of the f32-cast rate */
        n->s.ws_sr = (float)sr;
        n->s.sample_duration = 1.0f / (float)sr;
        break;
    case O_ENVELOPE:  /* envelope.rs:124-126 */
    case O_ENVELOPE_IN:
    case O_ADSR_LIVE: /* envelope.rs:300-302 */
        n->s.esd = (float)(1.0 / sr);
        break;
    case O_PHASE_OSC: /* oscillator.rs:456-458 */
    case O_DSF:       /* oscillator.rs:168-170 */
        n->s.sample_duration = (float)(1.0 / sr);
        break;
    case O_ONEPOLE: /* filter.rs:47-50 etc. */
        n->s.sr = (float)sr;
        onepole_set(n, n->s.cutoff);
        break;
    case O_PLUCK: /* oscillator.rs:279-285 */
        if (n->s.pl_sr != sr) {
            n->s.pl_sr = sr;
            n->s.pl_init = 0;
        }
        break;
    case O_REZ: /* rez.rs:62-65 */
        n->s.sr = (float)sr;
        rez_set(n, n->s.cutoff, n->s.q);
        break;
    case O_FOLLOW: /* follow.rs:96-99 */
        n->s.sr = (float)sr;
        follow_set(n, n->s.fo_time);
        break;
    case O_AFOLLOW: /* follow.rs:217-221 */
        n->s.sr = (float)sr;
        afollow_set(n, n->s.fo_time, n->s.fo_rtime);
        break;
    case O_MORPH: /* svf.rs:1072-1074 -> Svf::set_sample_rate */
        n->s.sr = (float)sr;
        n->s.sc = svf_make(O_SVF_PEAK, n->s.sr, n->s.cutoff, n->s.q, n->s.gain);
        break;
    case O_TAP: { /* delay.rs:198-209 / :436-445 */
        float srf = (float)sr;
        if (n->s.tap_sr != srf) {
            n->s.tap_sr = srf;
            float blen;
            if (n->s.tap_linear) {
                n->s.tap_min_c = n->s.tap_min;
                n->s.tap_max_c = n->s.tap_max;
                blen = ceilf(n->s.tap_max * srf) + 2.0f;
            } else {
                n->s.tap_min_c = n->s.tap_min > 1.00001f / srf ? n->s.tap_min : 1.00001f / srf;
                n->s.tap_max_c = n->s.tap_max > 1.00001f / srf ? n->s.tap_max : 1.00001f / srf;
                blen = ceilf(n->s.tap_max * srf) + 3.0f + (float)SIMD_N;
            }
            size_t len = 1;
            while (len < (size_t)blen) len <<= 1;
            n->s.tbuf = (float *)realloc(n->s.tbuf, len * sizeof(float));
            n->s.tlen = len;
            leaf_reset(n);
        }
        break;
    }
    case O_CHAOS:
        n->s.sr = (float)sr;
        break;
    case O_SHAPER:
        if (n->s.sh[0].kind >= O_SH_ADAPTIVE_TANH) n->s.sh[0].smoothing = (float)o_adaptive_smoothing(n->s.sh[0].ts, sr);
        break;
    case O_NLBIQUAD: /* biquad.rs:535-538 */
        n->s.sr = (float)sr;
        n->s.bc = bq_by_mode(n->s.nl_mode, n->s.sr, n->s.center, n->s.q, n->s.gain);
        for (int i = 0; i < 2; i++)
            if (n->s.sh[i].kind >= O_SH_ADAPTIVE_TANH) n->s.sh[i].smoothing = (float)o_adaptive_smoothing(n->s.sh[i].ts, sr);
        break;
    case O_DELAY: /* delay.rs:105-113 */
        if (n->s.dsr != sr) {
            n->s.dsr = sr;
            n->s.time_in_samples = (size_t)round(n->s.dtime * sr);
            size_t len = n->s.time_in_samples + 1;
            n->s.dbuf = (float *)realloc(n->s.dbuf, len * sizeof(float));
            n->s.dlen = len;
            leaf_reset(n);
        }
        break;
    default: break;
    }
}

void o_set_sample_rate(onode *n, double sr) {
    if (n->type == O_OVERSAMPLE) { /* oversample.rs:137-140 */
        o_set_sample_rate(n->x, sr * 2.0);
        return;
    }
    if (n->x) o_set_sample_rate(n->x, sr);
    if (n->y) o_set_sample_rate(n->y, sr);
    for (int i = 0; i < n->nkids; i++) o_set_sample_rate(n->kids[i], sr);
    if (n->type == O_DECLICK) n->s.dc_sd = (float)(1.0 / sr); /* dynamics.rs:272-275 */
    if (n->type == O_PHASESYNTH) n->s.ws_sr = (float)sr;      /* wavetable.rs:391-393 */
    if (n->type == O_METER) meter_set_sr(n, sr);
    if (n->type == O_HOLD) n->s.hd_sd = 1.0 / sr; /* noise.rs:287-289 */
    if (n->type == O_LIMITER) limiter_set_sr(n, sr);
    leaf_set_sample_rate(n, sr);
}

/* AudioNode::set_hash (oscillator.rs:94-97, noise.rs:226-229); default is a no-op (audionode.rs:136-139) */
static void leaf_set_hash(onode *n, uint64_t hash) {
    if (n->type == O_SINE || n->type == O_NOISE || n->type == O_WAVESYNTH || n->type == O_PHASE_OSC || n->type == O_CHAOS ||
        n->type == O_MLS || n->type == O_DSF) { /* Mls::set_hash noise.rs:142-145 */
        n->s.hash = hash;
        leaf_reset(n);
    } else if (n->type == O_HOLD) { /* noise.rs:312-315: set_hash resets (the stream restarts) */
        n->s.hash = hash;
        n->s.hd_pos = 0; n->s.hd_t = 0.0; n->s.hd_next = 0.0;
    } else if (n->type == O_PLUCK) { /* oscillator.rs:307-310 */
        n->s.hash = hash;
        n->s.pl_init = 0;
    } else if (n->type == O_ADSR_LIVE || n->type == O_ENVELOPE || n->type == O_ENVELOPE_IN) { /* envelope.rs:346-349, 165-168: no reset */
        n->s.hash = hash;
        n->s.et_hash = hash;
    }
}

/* AudioNode::ping: leaf default audionode.rs:156-161; Pipe/Stack/Binop :1459-1461,:966-968; Unop :1286-1288 */
static uint64_t o_ping(onode *n, int probe, uint64_t hash) {
    switch (n->type) {
    case O_PIPE:
    case O_STACK:
    case O_BINOP:
    case O_BRANCH: /* audionode.rs:1753-1755 */
    case O_BUS:    /* audionode.rs:1888-1890 */
        return o_ping(n->y, probe, o_ping(n->x, probe, o_atto(hash, n->id)));
    case O_MULTI: { /* audionode.rs:2142-2148 and its four siblings */
        uint64_t h = o_atto(hash, n->id);
        for (int i = 0; i < n->nkids; i++) h = o_ping(n->kids[i], probe, h);
        return h;
    }
    case O_FEEDBACK: /* feedback.rs:152-154, 293-295 */
        if (n->y) return o_ping(n->y, probe, o_ping(n->x, probe, o_atto(hash, n->id)));
        return o_ping(n->x, probe, o_atto(hash, n->id));
    case O_WRAP: /* PulseWave::ping wavetable.rs:484-486: inner first, own ID last */
        return o_atto(o_ping(n->x, probe, hash), n->id);
    case O_THRU: /* audionode.rs:2026-2028 */
    case O_UNOP:
    case O_RESAMPLE:   /* resample.rs:308-310 */
    case O_OVERSAMPLE: /* oversample.rs:218-220 */
    case O_ALLNEST: /* delay.rs:337-339 */
        return o_ping(n->x, probe, o_atto(hash, n->id));
    case O_REVERB_STEREO: {
        /* This node stands for the tree reverb_stereo builds (prelude.rs:1755-1763):
         *   multisplit::<U2, U16>() >> fdn::<U32>(stacki::<U32>(|i| delay(..) >> fir(..))) >> sumf::<U32>(|x| pan(..)) * dc((1/16, 1/16))
         * = Pipe<Pipe<MultiSplit, Feedback<MultiStack<U32, Pipe<Delay, Fir>>>>, Binop<Mul, Reduce<U32, Pan>, Constant>>, and its ping is the walk
         * of that tree (Pipe / Binop: y.ping(x.ping(hash.hash(ID))); Feedback :152-154; MultiStack / Reduce: own ID, then the nodes in order;
         * leaves: hash.hash(ID)).  None of those nodes keeps hashed state, so only the hash handed on matters -- to whatever hashed node follows
         * the reverb in a Pipe, and, through the probe ping of a constructor, to every hashed node of the graph (found by the construction-hash
         * test of the reverb bench, tests/test_gpu_criterion.py: the leaf default below had stood in for the walk). */
        uint64_t h = o_atto(hash, 6);  /* Pipe<Pipe<..>, Binop> */
        h = o_atto(h, 6);              /* Pipe<MultiSplit, Feedback> */
        h = o_atto(h, 38);             /* MultiSplit */
        h = o_atto(h, 11);             /* Feedback */
        h = o_atto(h, 30);             /* MultiStack<U32, _> */
        for (int i = 0; i < 32; i++) {
            h = o_atto(h, 6);          /* Pipe<Delay, Fir> */
            h = o_atto(h, 13);         /* Delay */
            h = o_atto(h, 52);         /* Fir */
        }
        h = o_atto(h, 3);              /* Binop<FrameMul, Reduce, Constant> */
        h = o_atto(h, 31);             /* Reduce<U32, Pan, FrameAdd> */
        for (int i = 0; i < 32; i++) h = o_atto(h, 49); /* Pan */
        return o_atto(h, 2);           /* Constant */
    }
    default:
        if (!probe) leaf_set_hash(n, hash);
        return o_atto(hash, n->id);
    }
}

/* what every combinator constructor does (audionode.rs:871-876, 1242-1247, 1389-1394) */
static void ctor_ping(onode *n) {
    if ((n->x && n->x->ftz) || (n->y && n->y->ftz)) n->ftz = 1;
    for (int i = 0; i < n->nkids; i++)
        if (n->kids[i]->ftz) n->ftz = 1;
    uint64_t h = o_ping(n, 1, n->id); /* AttoHash::new(Self::ID) */
    o_ping(n, 0, h);
}

/* AudioNode::set_seed audionode.rs:366-368 */
void o_set_seed(onode *n, uint64_t seed) { o_ping(n, 0, seed); }

uint64_t o_sine_hash(const onode *n) { return n->s.hash; }
float o_sine_phase(const onode *n) { return n->s.phase; }
uint32_t o_noise_state(const onode *n) { return n->s.nstate; }
/* Setting::phase (oscillator.rs:88-92): stores only; caller resets (combinator.rs:263-267 resets) */
void o_sine_set_phase(onode *n, float phase) {
    n->s.has_initial_phase = 1;
    n->s.initial_phase = phase;
    leaf_reset(n);
}
void o_noise_set_seed(onode *n, uint64_t seed) {
    n->s.has_seed = 1;
    n->s.seed = seed;
    leaf_reset(n);
}

/* ------------------------------------------------------------------------------------------------------ */
/* constructors                                                                                           */
/* ------------------------------------------------------------------------------------------------------ */

onode *o_constant(int n, const float *v) { /* audionode.rs:465-475, ID 2 */
    onode *c = o_new(O_CONSTANT, 0, n, 2);
    for (int i = 0; i < n; i++) c->s.value[i] = v[i];
    return c;
}
onode *o_pass(void) { return o_new(O_PASS, 1, 1, 48); } /* audionode.rs:408-436 */

onode *o_sine(void) { /* Sine::new oscillator.rs:30-35, ID 21 */
    onode *n = o_new(O_SINE, 1, 1, 21);
    leaf_reset(n);
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
onode *o_noise(void) { return o_new(O_NOISE, 0, 1, 20); } /* Noise::new = default, noise.rs:179-183 */

/* FixedSvf::new svf.rs:879-892 with SvfParams{sample_rate: DEFAULT_SR,..} prelude.rs:2111-2121; ID 43 */
onode *o_fixed_svf(int mode, float cutoff, float q, float gain) {
    onode *n = o_new(O_FIXED_SVF, 1, 1, 43);
    n->s.mode = mode;
    n->s.sr = (float)DEFAULT_SR;
    n->s.cutoff = cutoff;
    n->s.q = q;
    n->s.gain = gain;
    n->s.sc = svf_make(mode, n->s.sr, cutoff, q, gain);
    return n;
}
/* Svf::new svf.rs:767-779; inputs = 3 (audio,cutoff,q) or 4 (+gain) for bell/shelves; ID 36 */
onode *o_svf(int mode, float cutoff, float q, float gain) {
    int nin = (mode >= O_SVF_BELL) ? 4 : 3;
    onode *n = o_new(O_SVF, nin, 1, 36);
    n->s.mode = mode;
    n->s.sr = (float)DEFAULT_SR;
    n->s.cutoff = cutoff;
    n->s.q = q;
    n->s.gain = gain;
    n->s.sc = svf_make(mode, n->s.sr, cutoff, q, gain);
    return n;
}
onode *o_biquad(float a1, float a2, float b0, float b1, float b2) { /* biquad.rs:151-158, ID 15 */
    onode *n = o_new(O_BIQUAD, 1, 1, 15);
    n->s.bc.a1 = a1; n->s.bc.a2 = a2; n->s.bc.b0 = b0; n->s.bc.b1 = b1; n->s.bc.b2 = b2;
    n->s.sr64 = DEFAULT_SR;
    return n;
}
onode *o_butter_lowpass(int inputs, float cutoff) { /* biquad.rs:234-250, ID 16 */
    onode *n = o_new(O_BUTTER_LOWPASS, inputs, 1, 16);
    n->s.sr = (float)DEFAULT_SR;
    n->s.sr64 = DEFAULT_SR;
    n->s.bc = bq_butter_lowpass(n->s.sr, cutoff);
    n->s.cutoff = cutoff;
    return n;
}
onode *o_resonator(int inputs, float center, float q) { /* biquad.rs:318-337, ID 17 */
    onode *n = o_new(O_RESONATOR, inputs, 1, 17);
    n->s.sr = (float)DEFAULT_SR;
    n->s.center = center;
    n->s.q = q;
    n->s.bc = bq_resonator(n->s.sr, center, q);
    return n;
}
onode *o_biquad_bank(void) { /* BiquadBank::<f32x8>::new biquad_bank.rs:30-35, ID 98 */
    onode *n = o_new(O_BIQUAD_BANK, 8, 8, 98);
    n->s.sr64 = DEFAULT_SR;
    return n;
}
/* Setting::biquad(..).index(i) -> biquad_bank.rs:86-96 */
void o_biquad_bank_set(onode *n, int index, float a1, float a2, float b0, float b1, float b2) {
    n->s.bank_c[index].a1 = a1; n->s.bank_c[index].a2 = a2;
    n->s.bank_c[index].b0 = b0; n->s.bank_c[index].b1 = b1; n->s.bank_c[index].b2 = b2;
}
onode *o_moog(int inputs, float cutoff, float q) { /* Moog::new moog.rs:37-44, ID 60 */
    onode *n = o_new(O_MOOG, inputs, 1, 60);
    n->s.sr = (float)DEFAULT_SR;
    moog_set_cutoff_q(n, cutoff, q);
    return n;
}
onode *o_fir(int n_taps, const float *w) { /* Fir::new fir.rs:21-27, ID 52 */
    onode *n = o_new(O_FIR, 1, 1, 52);
    n->s.fir_n = n_taps;
    for (int i = 0; i < n_taps; i++) n->s.w[i] = w[i];
    n->s.sr64 = DEFAULT_SR;
    return n;
}
onode *o_tick_node(int channels) { /* Tick::new delay.rs:24-31, ID 9 */
    onode *n = o_new(O_TICK, channels, channels, 9);
    n->s.sr64 = DEFAULT_SR;
    return n;
}
onode *o_delay(double time) { /* Delay::new delay.rs:80-91, ID 13 */
    onode *n = o_new(O_DELAY, 1, 1, 13);
    n->s.dtime = time;
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}

/* ---- wavetable data (wavetable.rs:82-84: Vec<(f32, Vec<f32>)>) ---- */
owavetable *o_wavetable_create(int n_tables, const float *pitches, const int *lengths, const float *data) {
    owavetable *t = (owavetable *)calloc(1, sizeof(owavetable));
    t->n = n_tables;
    t->pitch = (float *)malloc(sizeof(float) * (size_t)n_tables);
    t->len = (int *)malloc(sizeof(int) * (size_t)n_tables);
    t->tab = (float **)malloc(sizeof(float *) * (size_t)n_tables);
    size_t off = 0;
    for (int i = 0; i < n_tables; i++) {
        t->pitch[i] = pitches[i];
        t->len[i] = lengths[i];
        t->tab[i] = (float *)malloc(sizeof(float) * (size_t)lengths[i]);
        memcpy(t->tab[i], data + off, sizeof(float) * (size_t)lengths[i]);
        off += (size_t)lengths[i];
    }
    return t;
}
void o_wavetable_free(owavetable *t) {
    if (!t) return;
    for (int i = 0; i < t->n; i++) free(t->tab[i]);
    free(t->tab);
    free(t->len);
    free(t->pitch);
    free(t);
}

onode *o_wavesynth(const owavetable *table, int outputs) { /* WaveSynth::new wavetable.rs:270-281, ID 34 */
    onode *n = o_new(O_WAVESYNTH, 1, outputs, 34);
    n->s.wt = table;
    n->s.phase = 0.0f; /* NOT reset in new() */
    n->s.hash = 0;
    n->s.has_initial_phase = 0;
    n->s.table_hint = 0;
    n->s.ws_sr = (float)DEFAULT_SR;
    n->s.sample_duration = 1.0f / (float)DEFAULT_SR;
    return n;
}
onode *o_waveplayer(const float *data, int channels, size_t length, int channel, size_t start_point, size_t end_point,
                   long loop_point) { /* WavePlayer::new wave.rs:749-766 */
    if (channel < 0 || channel >= channels || end_point > length) return NULL; /* the asserts :756-757 */
    onode *n = o_new(O_WAVEPLAYER, 0, 1, 65);
    n->s.wp_data = data; n->s.wp_length = length; n->s.wp_channel = channel;
    n->s.wp_index = start_point; n->s.wp_start = start_point; n->s.wp_end = end_point; n->s.wp_loop = loop_point;
    return n;
}
onode *o_phasesynth(const owavetable *table) { /* PhaseSynth::new wavetable.rs:367-377 */
    onode *n = o_new(O_PHASESYNTH, 1, 1, 35);
    n->s.wt = table;
    n->s.phase = 0.0f;
    n->s.ps_ready = 0;
    n->s.table_hint = 0;
    n->s.ws_sr = (float)DEFAULT_SR;
    return n;
}
onode *o_wrap(onode *x, uint64_t id) {
    onode *n = o_new(O_WRAP, x->nin, x->nout, id);
    n->x = x; n->ftz = x->ftz;
    return n;
}
void o_wavesynth_set_phase(onode *n, float phase) { /* Setting::phase wavetable.rs:350-354 + reset */
    n->s.has_initial_phase = 1;
    n->s.initial_phase = phase;
    n->s.phase = phase;
}

/* adsr_live(attack, decay, sustain, release) = envelope2(closure) = EnvelopeIn::new(0.002, ..)
 * adsr.rs:21-57, prelude.rs:626-639, envelope.rs:228-250; ID 53 */
onode *o_adsr_live(float attack, float decay, float sustain, float release) {
    onode *n = o_new(O_ADSR_LIVE, 1, 1, 53);
    n->s.adsr_a = attack; n->s.adsr_d = decay; n->s.adsr_s = sustain; n->s.adsr_r = release;
    n->s.attacked = 0;
    n->s.attack_start = 0.0f;
    n->s.release_start = -1.0f;
    n->s.einterval = (float)0.002;
    n->s.hash = 0;
    leaf_set_sample_rate(n, DEFAULT_SR);
    leaf_reset(n);
    return n;
}

static void pan_weights(float value, float *l, float *r) { /* pan.rs:13-17 */
    float c = value;
    c = c > -1.0f ? c : -1.0f; /* clamp11: x.max(-1).min(1) */
    c = c < 1.0f ? c : 1.0f;
    float angle = (c + 1.0f) * (F32_PI * 0.25f);
    *l = o_cosf(angle);
    *r = o_sinf(angle);
}
onode *o_panner(int inputs, float pan) { /* Panner::new pan.rs:33-40, ID 49 */
    onode *n = o_new(O_PANNER, inputs, 2, 49);
    pan_weights(pan, &n->s.left_weight, &n->s.right_weight);
    return n;
}

onode *o_tap(int linear, float min_delay, float max_delay) { return o_multitap(linear, 1, min_delay, max_delay); }
onode *o_allnest2(onode *x) { /* allnest(x) = AllNest::<U2, _>::new(0.0, x) prelude32.rs:1112: coefficient on input 1 */
    onode *n = o_allnest(0.0f, x);
    n->nin = 2;
    return n;
}
onode *o_multitap(int linear, int taps, float min_delay, float max_delay) { /* Tap::<N>::new :164-176 (ID 50) / TapLinear::<N>::new :404-418 (ID 54) */
    if (taps < 1 || taps + 1 > O_MAX_CH) return NULL;
    onode *n = o_new(O_TAP, 1 + taps, 1, linear ? 54 : 50);
    n->s.tap_linear = linear;
    n->s.tap_min = min_delay;
    n->s.tap_max = max_delay;
    n->s.tap_sr = 0.0f;
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
onode *o_allnest(float coefficient, onode *x) { /* AllNest::new :313-326, ID 83 (no constructor ping) */
    onode *n = o_new(O_ALLNEST, 1, 1, 83);
    n->x = x; n->ftz = x->ftz;
    n->s.eta = coefficient;
    n->s.zz = 0.0f;
    return n;
}
static inline float splinef(float y0, float y1, float y2, float y3, float x) { /* math.rs:360-366 */
    return y1 + x * 0.5f * (y2 - y0 + x * (2.0f * y0 - 5.0f * y1 + 4.0f * y2 - y3 + x * (3.0f * (y1 - y2) + y3 - y0)));
}

static void onepole_set(onode *n, float c) {
    n->s.cutoff = c;
    switch (n->s.op_kind) {
    case O_OP_LOWPOLE:
    case O_OP_HIGHPOLE: n->s.op_coeff = o_expf(-F32_TAU * c / n->s.sr); break; /* filter.rs:35-38, 371-374 */
    case O_OP_DCBLOCK: n->s.op_coeff = 1.0f - F32_TAU / n->s.sr * c; break;     /* :121-124 */
    default: n->s.op_coeff = (1.0f - c) / (1.0f + c); break;                    /* Allpole :292-295 */
    }
}
onode *o_onepole(int kind, int inputs, float cutoff_or_delay) { /* ::new filter.rs:28-38,110-119,281-290,364-374 */
    static const uint64_t ids[4] = {18, 47, 22, 46};
    onode *n = o_new(O_ONEPOLE, inputs, 1, ids[kind]);
    n->s.op_kind = kind;
    n->s.sr = (float)DEFAULT_SR;
    onepole_set(n, cutoff_or_delay);
    return n;
}
onode *o_pinkpass(void) { return o_new(O_PINKPASS, 1, 1, 26); } /* filter.rs:190-197 */

/* Rez<f32, N>  rez.rs:23-47 (ID 75): Paul Kellett's resonant two-pole; bandpass = 0 lowpass / 1 bandpass */
static void rez_set(onode *n, float cutoff, float q) { /* set_cutoff_q :41-46 */
    n->s.cutoff = cutoff;
    n->s.rz_f = 2.0f * o_sinf(F32_PI * cutoff / n->s.sr);
    n->s.q = q;
    n->s.rz_fb = q + q / (1.0f - n->s.rz_f);
}
onode *o_rez(int inputs, float bandpass, float cutoff, float q) {
    onode *n = o_new(O_REZ, inputs, 1, 75);
    n->s.rz_buf0 = n->s.rz_buf1 = 0.0f;
    n->s.rz_f = 1.0f; n->s.rz_fb = 1.0f;
    n->s.sr = (float)DEFAULT_SR;
    n->s.rz_bandpass = bandpass;
    rez_set(n, cutoff, q);
    return n;
}
/* follow.rs:12-24 in f64; `log`/`exp` here are the C library's, the reference's are libm 0.2.15's: both are
 * accurate to < 1 ulp of f64 and the result is rounded to f32 (F::from_f64), so the f32 coefficient agrees except
 * when the f64 value sits within ~1e-16 relative of an f32 rounding boundary (parity unpinned at that level). */
static double halfway_coeff(double samples) {
    double r0 = log(samples > 1.0 ? samples : 1.0) - 0.861624594696583;
    double r1 = 1.0 / (1.0 + exp(0.0 - r0));
    double r2 = r1 * 1.13228543863477 - 0.1322853859;
    return 1.0 - (r2 < 0.9999999 ? r2 : 0.9999999);
}
static void follow_set(onode *n, float t) { /* set_response_time :61-67 */
    n->s.fo_time = t;
    n->s.fo_coeff = (float)halfway_coeff((double)(t * n->s.sr));
    if (n->s.fo_coeff_now < 1.0f) n->s.fo_coeff_now = n->s.fo_coeff;
}
onode *o_follow(float response_time) { /* Follow::new :48-56 (ID 24) */
    onode *n = o_new(O_FOLLOW, 1, 1, 24);
    n->s.fo_time = response_time;
    n->s.fo_coeff = 0.0f;
    leaf_reset(n);
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
static void afollow_set(onode *n, float a, float r) { /* set_time :178-192 */
    n->s.fo_time = a;
    n->s.fo_rtime = r;
    n->s.fo_coeff = (float)halfway_coeff((double)(a * n->s.sr));
    n->s.fo_rcoeff = (float)halfway_coeff((double)(r * n->s.sr));
    if (n->s.fo_coeff_now < 1.0f) {
        n->s.fo_coeff_now = n->s.fo_coeff;
        n->s.fo_rcoeff_now = n->s.fo_rcoeff;
    }
}
static void meter_set_sr(onode *n, double sr) { /* dynamics.rs:355-364 */
    if (n->s.mt_mode != O_METER_SAMPLE) n->s.mt_smoothing = (float)pow(0.5, 1.0 / (n->s.mt_timescale * sr));
}
onode *o_meter(int mode, double timescale, int monitor) {
    onode *n = o_new(O_METER, 1, 1, monitor ? 56 : 61);
    n->s.mt_mode = mode; n->s.mt_monitor = monitor; n->s.mt_timescale = timescale;
    n->s.mt_smoothing = 0.0f; n->s.mt_state = 0.0f;
    meter_set_sr(n, DEFAULT_SR);
    return n;
}
float o_meter_level(const onode *n) { return n->s.mt_mode == O_METER_RMS ? sqrtf(n->s.mt_state) : n->s.mt_state; }
onode *o_hold(float variability, const double *draws, size_t n_draws) { /* Hold::new noise.rs:255-263 */
    if (!n_draws) return NULL;
    onode *n = o_new(O_HOLD, 2, 1, 76);
    n->s.hd_var = variability;
    n->s.hd_draws = (double *)malloc(n_draws * sizeof(double));
    memcpy(n->s.hd_draws, draws, n_draws * sizeof(double));
    n->s.hd_n = n_draws;
    n->s.hd_pos = 0; n->s.hd_t = 0.0; n->s.hd_next = 0.0; n->s.hd_hold = 0.0f;
    n->s.hd_sd = 1.0 / DEFAULT_SR;
    return n;
}
onode *o_mixer(int inputs, int outputs, const float *matrix) {
    if (inputs < 1 || outputs < 1 || inputs > 8 || outputs > 8) return NULL;
    onode *n = o_new(O_MIXER, inputs, outputs, 84);
    for (int i = 0; i < inputs * outputs; i++) n->s.value[i] = matrix[i];
    return n;
}
onode *o_var_fn(float value, int outputs, o_map_fn fn, void *ctx) {
    onode *n = o_new(O_VAR, 0, outputs, 70);
    n->s.value[0] = value;
    n->map_fn = fn; n->map_ctx = ctx;
    return n;
}
onode *o_var(float value) {
    onode *n = o_new(O_VAR, 0, 1, 68);
    n->s.value[0] = value;
    return n;
}
void o_var_set(onode *n, float value) { n->s.value[0] = value; }

static void limiter_set_sr(onode *n, double sr) { /* dynamics.rs:188-199 */
    n->s.lm_index = 0;
    n->s.lm_sr = sr;
    double r = round(sr * n->s.lm_lookahead);
    size_t length = r < 1.0 ? 1 : (size_t)r;
    if (length != n->s.lm_length) { /* new_buffer :154-156, ReduceBuffer::new :75-88 */
        size_t leaf = 1;
        while (leaf < length) leaf <<= 1;
        n->s.lm_length = length;
        n->s.lm_leaf = leaf;
        n->s.lm_tree = (float *)realloc(n->s.lm_tree, (leaf + length + (length & 1)) * sizeof(float));
        n->s.lm_buf = (float *)realloc(n->s.lm_buf, length * (size_t)n->nin * sizeof(float));
    }
    o_set_sample_rate(n->aux, sr);
    memset(n->s.lm_tree, 0, (n->s.lm_leaf + n->s.lm_length + (n->s.lm_length & 1)) * sizeof(float));
    n->s.lm_fill = 0; /* buffer.clear() */
}
onode *o_afollow(float attack_time, float release_time);
onode *o_limiter(int channels, float attack_time, float release_time) { /* Limiter::new :159-171 with DEFAULT_SR (prelude32.rs:1275) */
    if (channels < 1 || channels > O_MAX_CH) return NULL;
    onode *n = o_new(O_LIMITER, channels, channels, 25);
    n->aux = o_afollow(attack_time * 0.4f, release_time * 0.4f);
    n->s.lm_lookahead = (double)attack_time;
    n->s.lm_length = 0;
    limiter_set_sr(n, DEFAULT_SR);
    return n;
}
onode *o_afollow(float attack_time, float release_time) { /* AFollow::new :157-166 (ID 29) */
    onode *n = o_new(O_AFOLLOW, 1, 1, 29);
    n->s.fo_time = attack_time;
    n->s.fo_rtime = release_time;
    n->s.fo_coeff = n->s.fo_rcoeff = 0.0f;
    leaf_reset(n);
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
/* Mls::new(MlsState::new(n))  noise.rs:58-62,109-117 (ID 19): state (1 << n) - 1 until the first reset / ping */
static const uint32_t MLS_POLY[31] = { /* noise.rs:23-55 */
    0x1u, 0x3u, 0x6u, 0xCu, 0x14u, 0x30u, 0x48u, 0xB8u, 0x110u, 0x240u, 0x500u, 0xCA0u, 0x1B00u, 0x3088u, 0x6000u, 0xD008u,
    0x12000u, 0x20400u, 0x63000u, 0x90000u, 0x140000u, 0x300000u, 0x420000u, 0xE10000u, 0x1200000u, 0x2000023u, 0x4000013u,
    0x9000000u, 0x14000000u, 0x20000029u, 0x48000000u};
onode *o_mls(unsigned bits) {
    onode *n = o_new(O_MLS, 0, 1, 19);
    n->s.mls_n = bits;
    n->s.mls_s = (1u << bits) - 1u;
    n->s.has_seed = 0; n->s.seed = 0; n->s.hash = 0;
    return n;
}
/* test helper: length of the cycle of MlsState::next from the all-ones state (a maximum length sequence has 2^n - 1) */
uint64_t o_mls_period(unsigned bits) {
    const uint32_t mask = (1u << bits) - 1u, start = mask;
    uint32_t s = start;
    uint64_t k = 0;
    do {
        uint32_t parity = (uint32_t)__builtin_popcount(MLS_POLY[bits - 1] & s) & 1u;
        s = ((s << 1) | parity) & mask;
        k++;
    } while (s != start && k <= (uint64_t)mask + 1u);
    return k;
}
void o_mls_set_seed(onode *n, uint64_t seed) { n->s.has_seed = 1; n->s.seed = seed; } /* Setting::seed :136-140 */
onode *o_morph(float cutoff, float q, float morph) { /* Morph::new svf.rs:1046-1061, ID 62 */
    onode *n = o_new(O_MORPH, 4, 1, 62);
    n->s.mode = O_SVF_PEAK;
    n->s.sr = (float)DEFAULT_SR;
    n->s.cutoff = cutoff; n->s.q = q; n->s.gain = 0.0f;
    n->s.sc = svf_make(O_SVF_PEAK, n->s.sr, cutoff, q, 0.0f);
    n->s.morph = morph;
    return n;
}

onode *o_shaper(int shape, float p0, float p1) { /* Shaper::new shape.rs:209-215, ID 42 */
    onode *n = o_new(O_SHAPER, 1, 1, 42);
    n->s.sh[0].kind = shape; n->s.sh[0].p0 = p0; n->s.sh[0].p1 = p1; n->s.sh[0].state = 0.0f;
    n->s.sh[0].ts = p1;  /* O_SH_ADAPTIVE_TANH: p1 = timescale */
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
onode *o_shaper_adaptive(int inner, float p0, float p1, float timescale) { /* Shaper::new(Adaptive::new(timescale, S)) shape.rs:173-183 */
    onode *n = o_new(O_SHAPER, 1, 1, 42);
    n->s.sh[0].kind = O_SH_ADAPTIVE + inner; n->s.sh[0].p0 = p0; n->s.sh[0].p1 = p1; n->s.sh[0].state = 0.0f;
    n->s.sh[0].ts = timescale;
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
onode *o_phase_osc(int kind) { /* Ramp::new :453 / PolySaw::new :537 / PolySquare::new :613 / PolyPulse::new :696 */
    static const uint64_t ids[4] = {94, 95, 96, 97};
    onode *n = o_new(O_PHASE_OSC, kind == O_OSC_POLYPULSE ? 2 : 1, 1, ids[kind]);
    n->s.osc_kind = kind;
    leaf_reset(n);
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
void o_osc_set_phase(onode *n, float phase) {
    n->s.has_initial_phase = 1;
    n->s.initial_phase = phase;
    leaf_reset(n);
}
/* Pluck::new oscillator.rs:229-242 (ID 58).  `excitation`: at least as many samples as the loop delay will need. */
onode *o_pluck(float frequency, float gain_per_second, float high_frequency_damping, const float *excitation, size_t n_exc) {
    onode *n = o_new(O_PLUCK, 1, 1, 58);
    float g = 1.0f - high_frequency_damping; /* fir3(1.0 - damping) prelude.rs:863-867 */
    float alpha = (g + 1.0f) / 2.0f, beta = (1.0f - alpha) / 2.0f;
    n->s.fir_n = 3;
    n->s.w[0] = beta; n->s.w[1] = alpha; n->s.w[2] = beta;
    n->s.op_kind = O_OP_ALLPOLE; /* Allpole::new(1.0) */
    n->s.sr = (float)DEFAULT_SR;
    onepole_set(n, 1.0f);
    n->s.pl_gain = (float)pow((double)gain_per_second, 1.0 / (double)frequency);
    n->s.pl_freq = frequency;
    n->s.pl_sr = DEFAULT_SR;
    n->s.pl_raw = (float *)malloc((n_exc ? n_exc : 1) * sizeof(float));
    memcpy(n->s.pl_raw, excitation, n_exc * sizeof(float));
    n->s.pl_raw_n = n_exc;
    return n;
}
static void pluck_initialize_line(onode *n) { /* oscillator.rs:244-268 */
    const double epsilon = 0.2;
    double total_delay = n->s.pl_sr / (double)n->s.pl_freq - 1.0;
    double loop_delay = floor(total_delay - epsilon);
    double allpass_delay = total_delay - loop_delay;
    n->s.op_x1 = n->s.op_y1 = 0.0f;      /* tuning.reset() */
    n->s.sr = (float)n->s.pl_sr;         /* tuning.set_sample_rate */
    onepole_set(n, (float)allpass_delay); /* tuning.set_delay */
    size_t len = loop_delay > 0.0 ? (size_t)loop_delay : 0;
    if (len > n->s.pl_raw_n) len = n->s.pl_raw_n; /* the caller supplied too little excitation */
    n->s.pl_line = (float *)realloc(n->s.pl_line, (len ? len : 1) * sizeof(float));
    n->s.pl_len = len;
    double mean = 0.0;
    for (size_t i = 0; i < len; i++) {
        n->s.pl_line[i] = n->s.pl_raw[i];
        mean += (double)n->s.pl_line[i];
    }
    mean /= (double)len;
    for (size_t i = 0; i < len; i++) n->s.pl_line[i] -= (float)mean;
    n->s.pl_pos = 0;
    n->s.pl_init = 1;
}
/* Dsf<U1/U2>  oscillator.rs:131-158 (ID 55) */
static inline float dsf_clamp_roughness(float r) { /* set_roughness :154-157: clamp(0.0001, 0.9999, r) = r.max(lo).min(hi) */
    r = fmaxf(r, 0.0001f);
    return fminf(r, 0.9999f);
}
onode *o_dsf(int inputs, float harmonic_spacing, float roughness) {
    onode *n = o_new(O_DSF, inputs, 1, 55);
    n->s.dsf_spacing = harmonic_spacing;
    n->s.dsf_roughness = roughness;
    leaf_reset(n);
    leaf_set_sample_rate(n, DEFAULT_SR);
    n->s.dsf_roughness = dsf_clamp_roughness(roughness);
    return n;
}
onode *o_chaos(int lorenz) { /* Rossler::new :331 (ID 73) / Lorenz::new :390 (ID 74) */
    onode *n = o_new(O_CHAOS, 1, 1, lorenz ? 74 : 73);
    n->s.lorenz = lorenz;
    leaf_reset(n);
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}
/* FbBiquad::new :505 (ID 88) / FixedFbBiquad::new :603 (ID 90) / DirtyBiquad::new :712 (ID 89) /
 * FixedDirtyBiquad::new :819 (ID 91), then set_center_q(_gain) like the prelude constructors (prelude.rs:2912-3100) */
onode *o_nlbiquad(int dirty, int inputs, int mode, int shape, float p0, float p1, float center, float q, float gain) {
    uint64_t id = dirty ? (inputs == 1 ? 91 : 89) : (inputs == 1 ? 90 : 88);
    onode *n = o_new(O_NLBIQUAD, inputs, 1, id);
    n->s.nl_dirty = dirty;
    n->s.nl_mode = mode;
    for (int i = 0; i < 2; i++) { n->s.sh[i].kind = shape; n->s.sh[i].p0 = p0; n->s.sh[i].p1 = p1; n->s.sh[i].state = 0.0f; n->s.sh[i].ts = p1; }
    n->s.center = center; n->s.q = q; n->s.gain = gain;
    n->s.ns1 = n->s.ns2 = 0.0f;
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}

/* ---- reverb_stereo (prelude.rs:1732-1762) ---- */
static const double RV_DELAYS[32] = {
    0.073904, 0.052918, 0.066238, 0.066387, 0.037783, 0.080073, 0.050961, 0.075900, 0.043646,
    0.072095, 0.056194, 0.045961, 0.058934, 0.068016, 0.047529, 0.058156, 0.072972, 0.036084,
    0.062715, 0.076377, 0.044339, 0.076725, 0.077884, 0.046126, 0.067741, 0.049800, 0.051709,
    0.082923, 0.070121, 0.079315, 0.055039, 0.081859,
};
static float smooth9f(float x) { /* math.rs:431-437 */
    float x2 = x * x;
    return ((((70.0f * x - 315.0f) * x + 540.0f) * x - 420.0f) * x + 126.0f) * x2 * x2 * x;
}
void o_reverb_stereo_params(double room_size, double time, double damping, double sample_rate, float *w3, int *delays32,
                            float *wl32, float *wr32) {
    /* a = pow(db_amp(-60.0), 0.03 * room_size / 10.0 / time) as f32; db_amp(x) = exp10(x/20) = exp((x/20)*LN_10)
     * (math.rs:76-78,294-296) -- f64 libm; the host C library's exp/pow stand in for the `libm` crate here */
    double db_amp = exp((-60.0 / 20.0) * 2.302585092994046);
    float a = (float)pow(db_amp, 0.03 * room_size / 10.0 / time);
    /* fir3(gain).weights() * a : prelude.rs:863-867 */
    float gain = 1.0f - (float)damping;
    float alpha = (gain + 1.0f) / 2.0f;
    float beta = (1.0f - alpha) / 2.0f;
    w3[0] = beta * a; w3[1] = alpha * a; w3[2] = beta * a;
    for (int i = 0; i < 32; i++) {
        delays32[i] = (int)round(RV_DELAYS[i] * room_size / 10.0 * sample_rate); /* Delay::set_sample_rate delay.rs:108 */
        /* sumf::<U32>(|x| pan(lerp(-1.0, 1.0, smooth9(x)))), x = (i / 31) as f32  (prelude.rs:1603-1622,1759) */
        float x = (float)((double)i / 31.0);
        float t = smooth9f(x);
        float p = -1.0f * (1.0f - t) + 1.0f * t;
        pan_weights(p, &wl32[i], &wr32[i]);
    }
}
onode *o_reverb_stereo(double room_size, double time, double damping) {
    onode *n = o_new(O_REVERB_STEREO, 2, 2, 6);
    n->s.rv_room = room_size; n->s.rv_time = time; n->s.rv_damping = damping;
    n->s.rv_sr = 0.0;
    n->ftz = 1; /* the node stands for a tree with a Feedback node in it: constructing it runs prevent_denormals() (feedback.rs:96), after which the
                 * reference renders the WHOLE graph on that thread under FTZ + DAZ -- the nodes in front of the reverb and a dry bus around it
                 * included, not only the reverb's own arithmetic (see o_feedback; combinators inherit the flag in ctor_ping) */
    leaf_set_sample_rate(n, DEFAULT_SR);
    return n;
}

/* ---- Oversampler<X>  oversample.rs (ID 51) --------------------------------------------------------------------
 * HALFBAND_MIN :329-373, grouped into f32x8 slices :381-541.  wide's f32x8 mul_add / reduce_add are restated for the
 * default x86-64 build (no `fma`, no `avx` target feature): mul_add = a * b + c unfused; reduce_add = sum of the low
 * f32x4 (iter().sum(): ((0 + a0) + a1) + a2) + a3) plus the same sum of the high f32x4.  Parity unpinned (wide 1.1.1
 * source is not under /root/reference). */
static const float HALFBAND_MIN[43] = {
    4.73552339e-02f, 1.81988040e-01f, 3.49148434e-01f, 3.92748135e-01f, 2.18230867e-01f, -5.31842843e-02f, -1.79186566e-01f,
    -7.34488007e-02f, 8.94524103e-02f, 1.00868556e-01f, -2.08681451e-02f, -8.82510989e-02f, -2.07640777e-02f, 6.22587555e-02f,
    4.07776255e-02f, -3.52258090e-02f, -4.57407870e-02f, 1.27033444e-02f, 4.14376136e-02f, 3.30799834e-03f, -3.24608206e-02f,
    -1.27856355e-02f, 2.21659033e-02f, 1.67803711e-02f, -1.27406974e-02f, -1.68177367e-02f, 5.35518220e-03f, 1.44761581e-02f,
    -3.70651781e-04f, -1.11140183e-02f, -2.40622311e-03f, 7.71596027e-03f, 3.48227062e-03f, -4.86763558e-03f, -3.45536353e-03f,
    2.79880054e-03f, 2.86736431e-03f, -1.48746153e-03f, -2.11827989e-03f, 7.72684113e-04f, 1.44384114e-03f, -4.49807048e-04f,
    -9.41945265e-04f};
static inline float os_dec_coeff(int k) { return k < 5 ? 0.0f : HALFBAND_MIN[k - 5]; }            /* DECIMATING_COEFFS :381-449 */
static inline float os_even_coeff(int k) { return k < 2 ? 0.0f : HALFBAND_MIN[2 * (k - 2)]; }     /* INTERPOLATING_EVEN :453-484 */
static inline float os_odd_coeff(int k) { return k < 3 ? 0.0f : HALFBAND_MIN[2 * (k - 3) + 1]; }  /* INTERPOLATING_ODD :488-519 */
static inline float wide_reduce_add8(const float *a) {
    float lo = 0.0f, hi = 0.0f;
    for (int j = 0; j < 4; j++) lo += a[j];
    for (int j = 4; j < 8; j++) hi += a[j];
    return lo + hi;
}
static void os_interpolate(const float *ring, size_t new_index, float *even, float *odd) { /* :11-41 */
    size_t start = new_index + (129 - 3 * 8);
    float ae[8] = {0}, ao[8] = {0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 8; j++) {
            float smp = ring[(start + (size_t)i * 8 + j) & 0x7f];
            ae[j] = smp * os_even_coeff(i * 8 + j) + ae[j];
            ao[j] = smp * os_odd_coeff(i * 8 + j) + ao[j];
        }
    *even = wide_reduce_add8(ae) * 2.0f;
    *odd = wide_reduce_add8(ao) * 2.0f;
}
static float os_decimate(const float *ring, size_t last_index) { /* :43-64 */
    size_t start = last_index + (129 - (43 / 8 + 1) * 8);
    float acc[8] = {0};
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) acc[j] = ring[(start + (size_t)i * 8 + j) & 0x7f] * os_dec_coeff(i * 8 + j) + acc[j];
    return wide_reduce_add8(acc);
}
onode *o_resample(onode *x) { /* Resample::new resample.rs:228-238 (ID 69): x is a generator */
    if (x->nin != 0) return NULL;
    onode *n = o_new(O_RESAMPLE, 1, x->nout, 69);
    n->x = x; n->ftz = x->ftz;
    n->s.rs_buf = (float *)calloc((size_t)x->nout * 128, sizeof(float));
    n->s.rs_consumer = 1.0;
    n->s.rs_producer = 0;
    o_set_sample_rate(x, DEFAULT_SR);
    uint64_t h = o_ping(x, 1, 69);
    o_ping(x, 0, h);
    return n;
}
onode *o_oversample(onode *x) { /* Oversampler::new :90-104 */
    onode *n = o_new(O_OVERSAMPLE, x->nin, x->nout, 51);
    n->x = x; n->ftz = x->ftz;
    n->s.os_inv = (float *)calloc((size_t)(x->nin ? x->nin : 1) * 128, sizeof(float));
    n->s.os_outv = (float *)calloc((size_t)x->nout * 128, sizeof(float));
    o_set_sample_rate(x, DEFAULT_SR * 2.0);
    uint64_t h = o_ping(x, 1, 51); /* node.ping(true, AttoHash::new(Self::ID)) -- the inner node, not self */
    o_ping(x, 0, h);
    return n;
}

onode *o_multipass(int n) { return o_new(O_MULTIPASS, n, n, 0); }
onode *o_sink(int n) { return o_new(O_SINK, n, 0, 1); }
onode *o_split(int m, int n) { /* IDs 40 / 38 */
    onode *s = o_new(O_SPLIT, m, m * n, m == 1 ? 40 : 38);
    s->jm = m; s->jn = n;
    return s;
}
onode *o_join(int m, int n) { /* IDs 41 / 39 */
    onode *s = o_new(O_JOIN, m * n, m, m == 1 ? 41 : 39);
    s->jm = m; s->jn = n;
    return s;
}
onode *o_reverse(int n) { return o_new(O_REVERSE, n, n, 45); }
onode *o_impulse(int n) { /* Impulse::new audionode.rs:2847-2852 */
    onode *s = o_new(O_IMPULSE, 0, n, 81);
    s->s.value[0] = 1.0f;
    return s;
}
onode *o_map(int inputs, int outputs, o_map_fn fn, void *ctx) {
    onode *s = o_new(O_MAP, inputs, outputs, 5);
    s->map_fn = fn; s->map_ctx = ctx;
    return s;
}
onode *o_shape_fn(o_map_fn fn, void *ctx) { /* the trait's default simd() is shape() per lane (shape.rs:16-18) */
    onode *s = o_new(O_MAP, 1, 1, 42);
    s->map_fn = fn; s->map_ctx = ctx;
    return s;
}
onode *o_declick(float duration) { /* Declick::new dynamics.rs:253-260 */
    onode *s = o_new(O_DECLICK, 1, 1, 23);
    s->s.dc_duration = duration;
    s->s.dc_t = 0.0f;
    s->s.dc_sd = (float)(1.0 / 44100.0);
    return s;
}
onode *o_branch(onode *x, onode *y) { /* Branch::new audionode.rs:1668-1674, ID 8 */
    if (x->nin != y->nin) return NULL;
    onode *n = o_new(O_BRANCH, x->nin, x->nout + y->nout, 8);
    n->x = x; n->y = y;
    ctor_ping(n);
    return n;
}
onode *o_bus(onode *x, onode *y) { /* Bus::new audionode.rs:1813-1819, ID 10 */
    if (x->nin != y->nin || x->nout != y->nout) return NULL;
    onode *n = o_new(O_BUS, x->nin, x->nout, 10);
    n->x = x; n->y = y;
    n->tmp = (float *)calloc((size_t)(x->nout ? x->nout : 1) * MAXB, sizeof(float));
    ctor_ping(n);
    return n;
}
onode *o_feedback(onode *x, onode *y, int hadamard) { /* Feedback::new feedback.rs:95-105 (ID 11), Feedback2::new :222-233 (ID 66) */
    if (x->nin != x->nout || (y && (y->nin != x->nout || y->nout != x->nout))) return NULL;
    if (hadamard && (x->nout & (x->nout - 1))) return NULL; /* FrameHadamard::new asserts a power of two (:27) */
    onode *n = o_new(O_FEEDBACK, x->nin, x->nout, y ? 66 : 11);
    n->x = x; n->y = y; n->hadamard = hadamard;
    n->ftz = 1; /* prevent_denormals() :96: the reference sets MXCSR FTZ + DAZ on the constructing thread for good; the
                   oracle applies that mode to every render of a graph that contains the node */
    ctor_ping(n);
    return n;
}
onode *o_reverb3(double time, double diffusion, onode **filters) { /* Reverb::new reverb.rs:162-207 */
    static const int ldelays[32] = {401, 421, 443, 463, 487, 503, 523, 547, 563, 587, 607, 619, 643, 661, 683, 701,
                                    727, 743, 761, 787, 809, 823, 839, 863, 883, 907, 929, 947, 967, 983, 1009, 1021};
    static const int rdelays[32] = {419, 433, 457, 479, 491, 509, 541, 557, 577, 593, 613, 631, 653, 673, 691, 719,
                                    733, 757, 773, 797, 811, 829, 853, 877, 887, 911, 937, 953, 977, 997, 1013, 1033};
    static const int delays[8] = {1087, 1091, 1093, 1097, 1103, 1109, 1117, 1123};
    static const int predelay[4] = {245, 367, 263, 349};
    for (int i = 0; i < 16; i++)
        if (filters[i]->nin != 1 || filters[i]->nout != 1) return NULL;
    onode *n = o_new(O_REVERB3, 2, 2, 85);
    float coeff = (float)(0.5 * (1.0 - diffusion) + 0.9 * diffusion); /* lerp(0.5, 0.9, diffusion) as f32, math.rs:169-178 */
    /* kids per block: allpass0[0..3], allpass1[0..3], filter0, filter1, delay */
    n->nkids = 8 * 11;
    n->kids = (onode **)calloc((size_t)n->nkids, sizeof(onode *));
    for (int i = 0; i < 8; i++) {
        onode **k = n->kids + i * 11;
        for (int j = 0; j < 4; j++) {
            k[j] = o_allnest(coeff, o_delay((double)(ldelays[i + j * 8] - 1) / DEFAULT_SR));
            k[4 + j] = o_allnest(coeff, o_delay((double)(rdelays[i + j * 8] - 1) / DEFAULT_SR));
        }
        k[8] = filters[2 * i];
        k[9] = filters[2 * i + 1];
        k[10] = o_delay((double)delays[7 - i] / DEFAULT_SR);
        if (k[8]->ftz || k[9]->ftz) n->ftz = 1;
    }
    n->rv3_a = (float)pow(exp(-60.0 / 20.0 * 2.302585092994046), 0.035 / time); /* pow(db_amp(-60.0), 0.035 / time) as f32; db_amp = exp10 = exp(x * LN_10), math.rs:76-78,294-296 */
    for (int i = 0; i < 4; i++) n->pre[i] = o_allnest(coeff, o_delay((double)(predelay[i] - 1) / DEFAULT_SR));
    n->rv3_feedback = 0.0f;
    return n;
}
onode *o_thru(onode *x) { /* Thru::new audionode.rs:1956-1961, ID 12 */
    onode *n = o_new(O_THRU, x->nin, x->nin, 12);
    n->x = x;
    n->tmp = (float *)calloc((size_t)(x->nout ? x->nout : 1) * MAXB, sizeof(float));
    ctor_ping(n);
    return n;
}
onode *o_multi(int kind, int count, onode **nodes, int op) {
    static const uint64_t ids[5] = {28, 30, 33, 31, 32};
    if (count < 1 || kind < 0 || kind > O_MULTI_CHAIN) return NULL;
    for (int i = 1; i < count; i++)
        if (nodes[i]->nin != nodes[0]->nin || nodes[i]->nout != nodes[0]->nout) return NULL;
    int xi = nodes[0]->nin, xo = nodes[0]->nout;
    if (kind == O_MULTI_CHAIN && xi != xo) return NULL;
    int nin = (kind == O_MULTI_STACK || kind == O_MULTI_REDUCE) ? xi * count : xi;
    int nout = (kind == O_MULTI_STACK || kind == O_MULTI_BRANCH) ? xo * count : xo;
    onode *n = o_new(O_MULTI, nin, nout, ids[kind]);
    n->multi = kind; n->op = op; n->nkids = count;
    n->kids = (onode **)calloc((size_t)count, sizeof(onode *));
    for (int i = 0; i < count; i++) n->kids[i] = nodes[i];
    n->tmp = (float *)calloc((size_t)(xo ? xo : 1) * MAXB, sizeof(float));
    n->tmp2 = (float *)calloc((size_t)(xo ? xo : 1) * MAXB, sizeof(float));
    ctor_ping(n);
    return n;
}

onode *o_pipe(onode *x, onode *y) { /* Pipe::new audionode.rs:1388-1394, ID 6 */
    if (x->nout != y->nin) return NULL;
    onode *n = o_new(O_PIPE, x->nin, y->nout, 6);
    n->x = x; n->y = y;
    n->tmp = (float *)calloc((size_t)(x->nout ? x->nout : 1) * MAXB, sizeof(float));
    ctor_ping(n);
    return n;
}
onode *o_stack(onode *x, onode *y) { /* Stack::new audionode.rs:1511-1517, ID 7 */
    onode *n = o_new(O_STACK, x->nin + y->nin, x->nout + y->nout, 7);
    n->x = x; n->y = y;
    ctor_ping(n);
    return n;
}
onode *o_binop(int op, onode *x, onode *y) { /* Binop::new audionode.rs:870-876, ID 3 */
    if (x->nout != y->nout) return NULL;
    onode *n = o_new(O_BINOP, x->nin + y->nin, x->nout, 3);
    n->x = x; n->y = y; n->op = op;
    n->tmp = (float *)calloc((size_t)x->nout * MAXB, sizeof(float));
    ctor_ping(n);
    return n;
}
onode *o_unop(int op, onode *x, float scalar) { /* Unop::new audionode.rs:1241-1247, ID 4 */
    onode *n = o_new(O_UNOP, x->nin, x->nout, 4);
    n->x = x; n->op = op; n->scalar = scalar;
    ctor_ping(n);
    return n;
}

/* ------------------------------------------------------------------------------------------------------ */
/* tick                                                                                                   */
/* ------------------------------------------------------------------------------------------------------ */

/* Svf::update_inputs for 3-input modes svf.rs:299-313, 4-input modes svf.rs:588-...: recompute on change */
static inline void svf_update_inputs(onode *n, const float *in) {
    float cutoff = in[1], q = in[2];
    if (n->nin == 4) {
        float gain = in[3];
        if (cutoff != n->s.cutoff || q != n->s.q || gain != n->s.gain) {
            n->s.cutoff = cutoff; n->s.q = q; n->s.gain = gain;
            n->s.sc = svf_make(n->s.mode, n->s.sr, cutoff, q, gain);
        }
    } else {
        if (cutoff != n->s.cutoff || q != n->s.q) {
            n->s.cutoff = cutoff; n->s.q = q;
            n->s.sc = svf_make(n->s.mode, n->s.sr, cutoff, q, n->s.gain);
        }
    }
}

/* FixedSvf::tick svf.rs:995-1006 == Svf::tick svf.rs:829-843 after update_inputs */
static inline float svf_tick(onode *n, float v0) {
    const svf_coefs *c = &n->s.sc;
    float v3 = v0 - n->s.ic2eq;
    float v1 = c->a1 * n->s.ic1eq + c->a2 * v3;
    float v2 = n->s.ic2eq + c->a2 * n->s.ic1eq + c->a3 * v3;
    n->s.ic1eq = 2.0f * v1 - n->s.ic1eq;
    n->s.ic2eq = 2.0f * v2 - n->s.ic2eq;
    return c->m0 * v0 + c->m1 * v1 + c->m2 * v2;
}

/* Biquad::tick biquad.rs:184-194 (DF1, left-to-right) */
static inline float biquad_tick(onode *n, float x0) {
    const bq_coefs *c = &n->s.bc;
    float y0 = c->b0 * x0 + c->b1 * n->s.x1 + c->b2 * n->s.x2 - c->a1 * n->s.y1 - c->a2 * n->s.y2;
    n->s.x2 = n->s.x1;
    n->s.x1 = x0;
    n->s.y2 = n->s.y1;
    n->s.y1 = y0;
    return y0;
}

/* Moog::tick moog.rs:82-100 */
static inline float moog_tick(onode *n, const float *in) {
    if (n->nin > 1) moog_set_cutoff_q(n, in[1], in[2]); /* unconditional, every sample (moog.rs:83-85) */
    float x = -n->s.rez * n->s.s3 + in[0];
    n->s.s0 = (x + n->s.px) * n->s.p - n->s.k * n->s.s0;
    n->s.s1 = (n->s.s0 + n->s.ps0) * n->s.p - n->s.k * n->s.s1;
    n->s.s2 = (n->s.s1 + n->s.ps1) * n->s.p - n->s.k * n->s.s2;
    n->s.s3 = o_tanhf((n->s.s2 + n->s.ps2) * n->s.p - n->s.k * n->s.s3);
    n->s.px = x;
    n->s.ps0 = n->s.s0;
    n->s.ps1 = n->s.s1;
    n->s.ps2 = n->s.s2;
    return n->s.s3;
}

/* Fir::tick fir.rs:57-70 */
static inline float fir_tick(onode *n, float x) {
    int N = n->s.fir_n;
    for (int i = 0; i + 1 < N; i++) n->s.v[i] = n->s.v[i + 1];
    n->s.v[N - 1] = x;
    float output = 0.0f;
    for (int i = 0; i < N; i++) output += n->s.w[i] * n->s.v[i];
    return output;
}

static inline float binop_apply(int op, float x, float y) { /* FrameAdd/Sub/Mul audionode.rs:725-847 */
    switch (op) {
    case O_ADD: return x + y;
    case O_SUB: return x - y;
    default: return x * y;
    }
}
static inline float unop_apply(int op, float x, float s) { /* FrameNeg/Id/AddScalar/NegAddScalar/MulScalar :1030-1228 */
    switch (op) {
    case O_NEG: return -x;
    case O_ID: return x;
    case O_ADD_SCALAR: return x + s;
    case O_NEG_ADD_SCALAR: return -x + s;
    default: return x * s;
    }
}

/* ---- Wavetable (wavetable.rs:24-38, 154-241) ---- */
static inline float optimal4x44(float a0, float a1, float a2, float a3, float x) {
    float z = x - (float)0.5;
    float even1 = a2 + a1, odd1 = a2 - a1, even2 = a3 + a0, odd2 = a3 - a0;
    float c0 = even1 * (float)0.4656725512077848 + even2 * (float)0.03432729708429672;
    float c1 = odd1 * (float)0.5374383075356016 + odd2 * (float)0.1542946255730746;
    float c2 = even1 * (float)-0.25194210134021744 + even2 * (float)0.2519474493593906;
    float c3 = odd1 * (float)-0.46896069955075126 + odd2 * (float)0.15578800670302476;
    float c4 = even1 * (float)0.00986988334359864 + even2 * (float)-0.00989340017126506;
    return (((c4 * z + c3) * z + c2) * z + c1) * z + c0;
}
static inline float wt_at(const owavetable *t, size_t i, float phase) { /* at() :154-166 == at_simd lane :169-186 */
    const float *tab = t->tab[i];
    size_t len = (size_t)t->len[i];
    float p = (float)len * phase;
    size_t i1 = (size_t)p; /* to_int_unchecked / fast_trunc_int: truncation, phase in 0...1 */
    float w = p - (float)i1;
    size_t mask = len - 1;
    size_t i0 = (i1 - 1) & mask;
    i1 = i1 & mask;
    size_t i2 = (i1 + 1) & mask;
    size_t i3 = (i1 + 2) & mask;
    return optimal4x44(tab[i0], tab[i1], tab[i2], tab[i3], w);
}
static inline size_t wt_table_index(const owavetable *t, size_t hint, float frequency) { /* :189-211 */
    if (frequency >= t->pitch[hint] && frequency <= t->pitch[hint + 1]) return hint;
    size_t i0 = 0, i1 = (size_t)t->n - 3;
    while (i0 < i1) {
        size_t i = (i0 + i1) >> 1;
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
static inline float fmaxf_rs(float a, float b) { return fmaxf(a, b); } /* f32::max (lib.rs:200-206): IEEE maxNum */
static inline float clamp01f(float x) { /* math.rs:136-138: x.max(0).min(1) */
    x = x > 0.0f ? x : 0.0f;
    return x < 1.0f ? x : 1.0f;
}
static inline float wt_read(const owavetable *t, size_t *hint, float frequency, float phase) { /* read :214-226 */
    size_t table = wt_table_index(t, *hint, frequency);
    float w = clamp01f((frequency - t->pitch[table]) / (t->pitch[table + 1] - t->pitch[table]));
    *hint = table;
    return (1.0f - w) * wt_at(t, table + 1, phase) + w * wt_at(t, table + 2, phase);
}

/* ---- adsr_live closure (adsr.rs:21-70) and EnvelopeIn::next_segment (envelope.rs:252-278), F = f32 ---- */
static inline float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; } /* math.rs:169-178 */
static float adsr_closure(onode *n, float time, float control) {
    if (n->s.release_start >= 0.0f && control > 0.0f) {
        n->s.attacked = 1;
        n->s.attack_start = time;
        n->s.release_start = -1.0f;
    } else if (n->s.release_start < 0.0f && control <= 0.0f) {
        n->s.release_start = time;
    }
    if (!n->s.attacked) return 0.0f;
    float tt = time - n->s.attack_start, ads;
    if (tt < n->s.adsr_a) {
        ads = lerpf(0.0f, 1.0f, tt / n->s.adsr_a);
    } else {
        float decay_time = tt - n->s.adsr_a;
        ads = decay_time < n->s.adsr_d ? lerpf(1.0f, n->s.adsr_s, decay_time / n->s.adsr_d) : n->s.adsr_s;
    }
    if (n->s.release_start < 0.0f) return ads;
    float a = n->s.release_start + n->s.adsr_r, b = n->s.release_start;
    return ads * clamp01f((time - a) / (b - a));
}
static void envelope_next_segment(onode *n) { /* Envelope::next_segment envelope.rs:79-98 */
    n->s.et0 = n->s.et1;
    for (int i = 0; i < n->nout; i++) n->s.env_v0[i] = n->s.env_v1[i];
    float next_interval = lerpf(0.75f, 1.25f, (float)o_rnd1(n->s.et_hash)) * n->s.einterval;
    n->s.et1 = n->s.et0 + next_interval;
    n->s.env_fn(n->s.et1, n->s.env_v1, n->s.env_ctx);
    n->s.et_hash = n->s.et_hash * 6364136223846793005ULL + 1ULL;
    float u = (n->s.et - n->s.et0) / (n->s.et1 - n->s.et0);
    float samples = next_interval / n->s.esd;
    for (int i = 0; i < n->nout; i++) {
        n->s.env_val[i] = lerpf(n->s.env_v0[i], n->s.env_v1[i], u);
        n->s.env_d[i] = (n->s.env_v1[i] - n->s.env_v0[i]) / samples;
    }
}
onode *o_envelope(float interval, int outputs, o_env_fn fn, void *ctx) { /* Envelope::new :58-76 (ID 14) */
    if (outputs < 1 || outputs > O_MAX_ENV) return NULL;
    onode *n = o_new(O_ENVELOPE, 0, outputs, 14);
    n->s.env_fn = fn;
    n->s.env_ctx = ctx;
    n->s.einterval = interval;
    leaf_set_sample_rate(n, DEFAULT_SR);
    leaf_reset(n);
    return n;
}
/* The two closures of the reference's own criterion benches (benches/benchmark.rs) as C functions, so that a CPU timing of those graphs
   does not pay a Python callback per envelope sample (tests/criterion_graphs.py passes their addresses to o_envelope). */
void o_envfn_criterion_envelope(float t, float *out, void *ctx) { /* benchmark.rs:57  |t| (-t).exp() * sin_hz(1.0, t);  sin_hz math.rs:462-464 */
    (void)ctx;
    out[0] = o_expf(-t) * o_sinf(t * 1.0f * F32_TAU);
}
void o_envfn_criterion_phaser(float t, float *out, void *ctx) { /* benchmark.rs:94  |t| sin_hz(0.1, t) * 0.5 + 0.5, inside phaser's
                                                                     lfo(move |t| lerp(2.0, 20.0, clamp01(phase_f(t)))) prelude.rs:2747 */
    (void)ctx;
    float p = o_sinf(t * 0.1f * F32_TAU) * 0.5f + 0.5f;
    out[0] = lerpf(2.0f, 20.0f, clamp01f(p));
}
static void envin_next_segment(onode *n, const float *input) { /* EnvelopeIn::next_segment envelope.rs:244-278 */
    if (n->s.et0 == 0.0f && n->s.et1 == 0.0f) {
        n->s.envin_fn(n->s.et0, input, n->s.env_v0, n->s.env_ctx);
    } else {
        n->s.et0 = n->s.et1;
        for (int i = 0; i < n->nout; i++) n->s.env_v0[i] = n->s.env_v1[i];
    }
    float next_interval = lerpf(0.75f, 1.25f, (float)o_rnd1(n->s.et_hash)) * n->s.einterval;
    n->s.et1 = n->s.et0 + next_interval;
    n->s.envin_fn(n->s.et1, input, n->s.env_v1, n->s.env_ctx);
    n->s.et_hash = n->s.et_hash * 6364136223846793005ULL + 1ULL;
    float u = (n->s.et - n->s.et0) / (n->s.et1 - n->s.et0);
    float samples = next_interval / n->s.esd;
    for (int i = 0; i < n->nout; i++) {
        n->s.env_val[i] = lerpf(n->s.env_v0[i], n->s.env_v1[i], u);
        n->s.env_d[i] = (n->s.env_v1[i] - n->s.env_v0[i]) / samples;
    }
}
onode *o_envelope_in(float interval, int inputs, int outputs, o_envin_fn fn, void *ctx) { /* EnvelopeIn::new :221-241 (ID 53) */
    if (outputs < 1 || outputs > O_MAX_ENV || inputs < 0 || inputs > O_MAX_CH) return NULL;
    onode *n = o_new(O_ENVELOPE_IN, inputs, outputs, 53);
    n->s.envin_fn = fn;
    n->s.env_ctx = ctx;
    n->s.einterval = interval;
    leaf_set_sample_rate(n, DEFAULT_SR);
    leaf_reset(n);
    return n;
}
static void env_next_segment(onode *n, float input) {
    if (n->s.et0 == 0.0f && n->s.et1 == 0.0f) {
        n->s.ev0 = adsr_closure(n, n->s.et0, input);
    } else {
        n->s.et0 = n->s.et1;
        n->s.ev0 = n->s.ev1;
    }
    float next_interval = lerpf(0.75f, 1.25f, (float)o_rnd1(n->s.et_hash)) * n->s.einterval;
    n->s.et1 = n->s.et0 + next_interval;
    n->s.ev1 = adsr_closure(n, n->s.et1, input);
    n->s.et_hash = n->s.et_hash * 6364136223846793005ULL + 1ULL;
    float u = (n->s.et - n->s.et0) / (n->s.et1 - n->s.et0);
    n->s.ev = lerpf(n->s.ev0, n->s.ev1, u);
    float samples = next_interval / n->s.esd;
    n->s.evd = (n->s.ev1 - n->s.ev0) / samples;
}

/* A graph that contains a Feedback node renders under MXCSR = 0x9fc0 (FTZ + DAZ), see o_feedback. */
static __thread int g_ftz_on;
#if defined(__x86_64__) || defined(__i386__)
#define FTZ_ENTER unsigned int ftz_csr = _mm_getcsr(); _mm_setcsr(0x9fc0); g_ftz_on = 1
#define FTZ_LEAVE g_ftz_on = 0; _mm_setcsr(ftz_csr)
#else
#define FTZ_ENTER g_ftz_on = 1
#define FTZ_LEAVE g_ftz_on = 0
#endif

void o_tick(onode *n, const float *in, float *out) {
    if (n->ftz && !g_ftz_on) {
        FTZ_ENTER;
        o_tick(n, in, out);
        FTZ_LEAVE;
        return;
    }
    float t[O_MAX_CH];
    switch (n->type) {
    case O_CONSTANT: /* audionode.rs:496-499 */
        for (int i = 0; i < n->nout; i++) out[i] = n->s.value[i];
        break;
    case O_PASS: out[0] = in[0]; break;
    case O_SINE: { /* oscillator.rs:67-72 */
        float phase = n->s.phase;
        n->s.phase += in[0] * n->s.sample_duration;
        n->s.phase -= floorf(n->s.phase);
        out[0] = o_sinf(phase * F32_TAU);
        break;
    }
    case O_NOISE: /* noise.rs:197-202 */
        n->s.nstate += 1u;
        out[0] = (float)(o_hash32x(n->s.nstate) >> 8) * (2.0f / (float)((1 << 24) - 1)) - 1.0f;
        break;
    case O_SVF:
        svf_update_inputs(n, in);
        out[0] = svf_tick(n, in[0]);
        break;
    case O_FIXED_SVF: out[0] = svf_tick(n, in[0]); break;
    case O_BIQUAD: out[0] = biquad_tick(n, in[0]); break;
    case O_BUTTER_LOWPASS: /* biquad.rs:269-277 */
        if (n->nin > 1) {
            float cutoff = in[1];
            if (cutoff != n->s.cutoff) {
                n->s.bc = bq_butter_lowpass(n->s.sr, cutoff);
                n->s.cutoff = cutoff;
            }
        }
        out[0] = biquad_tick(n, in[0]);
        break;
    case O_RESONATOR: /* biquad.rs:354-366 */
        if (n->nin >= 3) {
            float center = in[1], q = in[2];
            if (center != n->s.center || q != n->s.q) {
                n->s.bc = bq_resonator(n->s.sr, center, q);
                n->s.center = center;
                n->s.q = q;
            }
        }
        out[0] = biquad_tick(n, in[0]);
        break;
    case O_BIQUAD_BANK: /* biquad_bank.rs:73-84, lane-wise f32x8 arithmetic == 8 scalar DF1s */
        for (int l = 0; l < 8; l++) {
            const bq_coefs *c = &n->s.bank_c[l];
            float x0 = in[l];
            float y0 = c->b0 * x0 + c->b1 * n->s.bx1[l] + c->b2 * n->s.bx2[l] - c->a1 * n->s.by1[l] -
                       c->a2 * n->s.by2[l];
            n->s.bx2[l] = n->s.bx1[l];
            n->s.bx1[l] = x0;
            n->s.by2[l] = n->s.by1[l];
            n->s.by1[l] = y0;
            out[l] = y0;
        }
        break;
    case O_MOOG: out[0] = moog_tick(n, in); break;
    case O_FIR: out[0] = fir_tick(n, in[0]); break;
    case O_TICK: /* delay.rs:47-52 */
        for (int i = 0; i < n->nout; i++) {
            out[i] = n->s.tickbuf[i];
            n->s.tickbuf[i] = in[i];
        }
        break;
    case O_DELAY: /* delay.rs:116-124 */
        n->s.dbuf[n->s.di] = in[0];
        n->s.di += 1;
        if (n->s.di >= n->s.dlen) n->s.di = 0;
        out[0] = n->s.dbuf[n->s.di];
        break;
    case O_WAVESYNTH: { /* wavetable.rs:310-324: increment + wrap BEFORE reading */
        float frequency = in[0];
        float delta = frequency * n->s.sample_duration;
        n->s.phase += delta;
        n->s.phase -= floorf(n->s.phase);
        out[0] = wt_read(n->s.wt, &n->s.table_hint, fabsf(frequency), n->s.phase);
        if (n->nout > 1) out[1] = n->s.phase;
        break;
    }
    case O_PHASESYNTH: { /* wavetable.rs:399-425 */
        float phase = in[0];
        phase = phase - floorf(phase);
        float delta;
        if (n->s.ps_ready) {
            float a = fabsf(phase - n->s.phase), b = fabsf(phase - 1.0f - n->s.phase), c = fabsf(phase + 1.0f - n->s.phase);
            float bc = b < c ? b : c;
            delta = a < bc ? a : bc;
        } else {
            n->s.ps_ready = 1;
            delta = 0.5f;
        }
        out[0] = wt_read(n->s.wt, &n->s.table_hint, delta * n->s.ws_sr, phase);
        n->s.phase = phase;
        break;
    }
    case O_WRAP: o_tick(n->x, in, out); break; /* wavetable.rs:472-474 */
    case O_METER: { /* MeterState::tick dynamics.rs:367-375; MeterNode::tick :427-430; Monitor::tick :482-486 */
        float v = in[0];
        if (n->s.mt_mode == O_METER_SAMPLE) n->s.mt_state = v;
        else if (n->s.mt_mode == O_METER_PEAK) n->s.mt_state = fmaxf_rs(n->s.mt_state * n->s.mt_smoothing, fabsf(v));
        else n->s.mt_state = n->s.mt_state * n->s.mt_smoothing + v * v * (1.0f - n->s.mt_smoothing);
        out[0] = n->s.mt_monitor ? v : o_meter_level(n);
        break;
    }
    case O_VAR: /* shared.rs:117-120; VarFn :171-173 */
        if (n->map_fn) n->map_fn(&n->s.value[0], out, n->map_ctx);
        else out[0] = n->s.value[0];
        break;
    case O_HOLD: /* noise.rs:292-304 */
        if (n->s.hd_t >= n->s.hd_next) {
            n->s.hd_hold = in[0];
            double r = n->s.hd_draws[n->s.hd_pos % n->s.hd_n];
            n->s.hd_pos++;
            double a = 1.0 - (double)n->s.hd_var, b = 1.0 + (double)n->s.hd_var;
            n->s.hd_next = n->s.hd_t + (a * (1.0 - r) + b * r) / (double)in[1];
        }
        n->s.hd_t += n->s.hd_sd;
        out[0] = n->s.hd_hold;
        break;
    case O_WAVEPLAYER: /* wave.rs:779-792 */
        if (n->s.wp_index < n->s.wp_end) {
            out[0] = n->s.wp_data[(size_t)n->s.wp_channel * n->s.wp_length + n->s.wp_index];
            n->s.wp_index += 1;
            if (n->s.wp_index == n->s.wp_end && n->s.wp_loop >= 0) n->s.wp_index = (size_t)n->s.wp_loop;
        } else {
            out[0] = 0.0f;
        }
        break;
    case O_MIXER: /* pan.rs:124-133 */
        for (int i = 0; i < n->nout; i++) {
            float value = 0.0f;
            for (int j = 0; j < n->nin; j++) value += in[j] * n->s.value[i * n->nin + j];
            t[i] = value;
        }
        for (int i = 0; i < n->nout; i++) out[i] = t[i];
        break;
    case O_REVERB3: { /* reverb.rs:241-272 */
#define MONO(node, xin) (mono_x = (xin), o_tick((node), &mono_x, &mono_y), mono_y)
        float mono_x, mono_y;
        float v0 = n->rv3_feedback, output0 = 0.0f, output1 = 0.0f;
        float input0 = MONO(n->pre[0], in[0] * 0.5f);
        input0 = MONO(n->pre[1], input0);
        float input1 = MONO(n->pre[2], in[1] * 0.5f);
        input1 = MONO(n->pre[3], input1);
        for (int b = 0; b < 8; b++) {
            onode **k = n->kids + b * 11;
            v0 = MONO(k[10], v0);
            v0 = MONO(k[0], n->rv3_a * v0 + input0);
            v0 = MONO(k[1], v0);
            v0 = MONO(k[2], v0);
            v0 = MONO(k[3], v0);
            v0 = MONO(k[8], v0);
            output0 = v0;
            v0 = MONO(k[4], n->rv3_a * v0 + input1);
            v0 = MONO(k[5], v0);
            v0 = MONO(k[6], v0);
            v0 = MONO(k[7], v0);
            v0 = MONO(k[9], v0);
            output1 = v0;
        }
#undef MONO
        n->rv3_feedback = v0;
        out[0] = output0;
        out[1] = output1;
        break;
    }
    case O_LIMITER: { /* dynamics.rs:202-226 */
        float amplitude = 0.0f;
        for (int c = 0; c < n->nin; c++) amplitude = fmaxf_rs(amplitude, fabsf(in[c]));
        { /* ReduceBuffer::set :104-112 */
            size_t i = n->s.lm_leaf + n->s.lm_index;
            n->s.lm_tree[i] = amplitude;
            while (i > 1) {
                float reduced = fmaxf_rs(n->s.lm_tree[i], n->s.lm_tree[i ^ 1]);
                i >>= 1;
                n->s.lm_tree[i] = reduced;
            }
        }
        float total = n->s.lm_tree[1];
        float *slot = n->s.lm_buf + n->s.lm_index * (size_t)n->nin;
        if (n->s.lm_fill < n->s.lm_length) {
            for (int c = 0; c < n->nin; c++) slot[c] = in[c];
            n->s.lm_fill++;
            if (n->s.lm_fill == n->s.lm_length) /* follower.set_value(total) follow.rs:196-200 */
                n->aux->s.fo_v1 = n->aux->s.fo_v2 = n->aux->s.fo_v3 = total;
            for (int c = 0; c < n->nout; c++) out[c] = 0.0f;
        } else {
            float o[O_MAX_CH];
            for (int c = 0; c < n->nin; c++) { o[c] = slot[c]; slot[c] = in[c]; }
            float x = fmaxf_rs(1.0f, total * 1.10f), y;
            o_tick(n->aux, &x, &y); /* filter_mono */
            float limit = n->aux->s.fo_v3;
            float z = 1.0f / limit;
            for (int c = 0; c < n->nout; c++) out[c] = o[c] * z;
        }
        n->s.lm_index++; /* advance :144-149 */
        if (n->s.lm_index >= n->s.lm_length) n->s.lm_index = 0;
        break;
    }
    case O_ENVELOPE_IN: /* envelope.rs:305-313 */
        if (n->s.et >= n->s.et1) envin_next_segment(n, in);
        for (int i = 0; i < n->nout; i++) {
            out[i] = n->s.env_val[i];
            n->s.env_val[i] += n->s.env_d[i];
        }
        n->s.et += n->s.esd;
        break;
    case O_ENVELOPE: /* envelope.rs:128-136 */
        if (n->s.et >= n->s.et1) envelope_next_segment(n);
        for (int i = 0; i < n->nout; i++) {
            out[i] = n->s.env_val[i];
            n->s.env_val[i] += n->s.env_d[i];
        }
        n->s.et += n->s.esd;
        break;
    case O_ADSR_LIVE: /* envelope.rs:305-313 */
        if (n->s.et >= n->s.et1) env_next_segment(n, in[0]);
        out[0] = n->s.ev;
        n->s.ev += n->s.evd;
        n->s.et += n->s.esd;
        break;
    case O_ONEPOLE: {
        if (n->nin > 1) {
            if (n->s.op_kind == O_OP_ALLPOLE) onepole_set(n, in[1]);
            else if (in[1] != n->s.cutoff) onepole_set(n, in[1]);
        }
        float x = in[0], c = n->s.op_coeff, y0;
        switch (n->s.op_kind) {
        case O_OP_LOWPOLE: n->s.op_y1 = (1.0f - c) * x + c * n->s.op_y1; out[0] = n->s.op_y1; break;        /* :64-66 */
        case O_OP_HIGHPOLE: y0 = c * (n->s.op_y1 + x - n->s.op_x1); n->s.op_x1 = x; n->s.op_y1 = y0; out[0] = y0; break;
        case O_OP_DCBLOCK: y0 = x - n->s.op_x1 + c * n->s.op_y1; n->s.op_x1 = x; n->s.op_y1 = y0; out[0] = y0; break;
        default: y0 = c * (x - n->s.op_y1) + n->s.op_x1; n->s.op_x1 = x; n->s.op_y1 = y0; out[0] = y0; break;
        }
        break;
    }
    case O_REZ: { /* rez.rs:67-82 */
        if (n->nin > 1) {
            float cutoff = in[1], q = in[2];
            if (cutoff != n->s.cutoff || q != n->s.q) rez_set(n, cutoff, q);
        }
        float hp = in[0] - n->s.rz_buf0;
        float bp = n->s.rz_buf0 - n->s.rz_buf1;
        n->s.rz_buf0 += n->s.rz_f * (hp + n->s.rz_fb * o_tanhf(bp));
        n->s.rz_buf1 += n->s.rz_f * (n->s.rz_buf0 - n->s.rz_buf1);
        out[0] = n->s.rz_buf1 - n->s.rz_bandpass * n->s.rz_buf0;
        break;
    }
    case O_FOLLOW: { /* follow.rs:101-110 */
        float c = n->s.fo_coeff_now, rc = 1.0f - c;
        n->s.fo_v1 = c * in[0] + rc * n->s.fo_v1;
        n->s.fo_v2 = c * n->s.fo_v1 + rc * n->s.fo_v2;
        n->s.fo_v3 = c * n->s.fo_v2 + rc * n->s.fo_v3;
        n->s.fo_coeff_now = n->s.fo_coeff;
        out[0] = n->s.fo_v3;
        break;
    }
    case O_AFOLLOW: { /* follow.rs:223-246 with ScalarOrPair for (T, T), combinator.rs:164-173 */
        float a = n->s.fo_coeff_now, r = n->s.fo_rcoeff_now, x = in[0];
#define FPOLE(input, cur) ((cur) + fmaxf_rs(0.0f, (input) - (cur)) * a - fmaxf_rs(0.0f, (cur) - (input)) * r)
        n->s.fo_v1 = FPOLE(x, n->s.fo_v1);
        n->s.fo_v2 = FPOLE(n->s.fo_v1, n->s.fo_v2);
        n->s.fo_v3 = FPOLE(n->s.fo_v2, n->s.fo_v3);
#undef FPOLE
        n->s.fo_coeff_now = n->s.fo_coeff;
        n->s.fo_rcoeff_now = n->s.fo_rcoeff;
        out[0] = n->s.fo_v3;
        break;
    }
    case O_MLS: { /* noise.rs:129-134, MlsState::next / value :81-96 */
        float value = (float)((n->s.mls_s >> (n->s.mls_n - 1)) & 1u);
        uint32_t feedback = MLS_POLY[n->s.mls_n - 1] & n->s.mls_s;
        uint32_t parity = (uint32_t)__builtin_popcount(feedback) & 1u;
        n->s.mls_s = ((n->s.mls_s << 1) | parity) & ((1u << n->s.mls_n) - 1u);
        out[0] = value * 2.0f - 1.0f;
        break;
    }
    case O_PINKPASS: { /* filter.rs:226-246 */
        float x = in[0], *b = n->s.pink;
        b[0] = (float)0.99886 * b[0] + x * (float)0.0555179;
        b[1] = (float)0.99332 * b[1] + x * (float)0.0750759;
        b[2] = (float)0.96900 * b[2] + x * (float)0.1538520;
        b[3] = (float)0.86650 * b[3] + x * (float)0.3104856;
        b[4] = (float)0.55000 * b[4] + x * (float)0.5329522;
        b[5] = (float)-0.7616 * b[5] - x * (float)0.0168980;
        out[0] = (b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + x * (float)0.5362) * (float)0.115830421;
        b[6] = x * (float)0.115926;
        break;
    }
    case O_MORPH: { /* svf.rs:1076-1080: Svf<PeakMode>::tick on inputs 0..3, then dry mix */
        n->s.morph = in[3];
        if (in[1] != n->s.cutoff || in[2] != n->s.q) {
            n->s.cutoff = in[1]; n->s.q = in[2];
            n->s.sc = svf_make(O_SVF_PEAK, n->s.sr, in[1], in[2], n->s.gain);
        }
        float fo = svf_tick(n, in[0]);
        out[0] = (fo + in[3] * in[0]) * 0.5f;
        break;
    }
    case O_TAP: { /* Tap::tick delay.rs:212-236 / TapLinear::tick :448-463 (the f32x8 process path reads the same samples) */
        size_t mask = n->s.tlen - 1;
        n->s.tbuf[n->s.ti] = in[0];
        float o = 0.0f;
        for (int k = 1; k < n->nin; k++) {
            float tap = rs_clampf(n->s.tap_min_c, n->s.tap_max_c, in[k]) * n->s.tap_sr;
            size_t tap_floor = (size_t)tap;
            size_t i1 = (n->s.ti - tap_floor) & mask;
            float d = tap - (float)tap_floor;
            if (n->s.tap_linear) {
                size_t i2 = (i1 - 1) & mask;
                o += n->s.tbuf[i1] * (1.0f - d) + n->s.tbuf[i2] * d;
            } else {
                size_t i0 = (i1 + 1) & mask, i2 = (i1 - 1) & mask, i3 = (i1 - 2) & mask;
                o += splinef(n->s.tbuf[i0], n->s.tbuf[i1], n->s.tbuf[i2], n->s.tbuf[i3], d);
            }
        }
        n->s.ti = (n->s.ti + 1) & mask;
        out[0] = o;
        break;
    }
    case O_RESAMPLE: { /* resample.rs:281-303 */
        float spd = in[0] > 0.0f ? in[0] : 0.0f; /* max(0.0, input[0]) */
        n->s.rs_consumer += (double)spd;
        double d = n->s.rs_consumer - floor(n->s.rs_consumer);
        size_t ci = (size_t)(n->s.rs_consumer - d);
        float inner[O_MAX_CH];
        while (ci + 2 >= n->s.rs_producer) {
            o_tick(n->x, NULL, inner);
            for (int c = 0; c < n->nout; c++) n->s.rs_buf[c * 128 + (n->s.rs_producer & 0x7f)] = inner[c];
            n->s.rs_producer += 1;
        }
        for (int c = 0; c < n->nout; c++) {
            const float *b = n->s.rs_buf + c * 128;
            out[c] = splinef(b[(ci + 0x7f) & 0x7f], b[ci & 0x7f], b[(ci + 1) & 0x7f], b[(ci + 2) & 0x7f], (float)d);
        }
        break;
    }
    case O_OVERSAMPLE: { /* oversample.rs:142-176 */
        float oi[O_MAX_CH], oi2[O_MAX_CH], oo[O_MAX_CH];
        for (int c = 0; c < n->nin; c++) {
            n->s.os_inv[c * 128 + n->s.os_in_i] = in[c];
            os_interpolate(n->s.os_inv + c * 128, n->s.os_in_i, &oi[c], &oi2[c]);
        }
        n->s.os_in_i = (n->s.os_in_i + 1) & 0x7f;
        o_tick(n->x, oi, oo);
        for (int c = 0; c < n->nout; c++) n->s.os_outv[c * 128 + n->s.os_out_i] = oo[c];
        n->s.os_out_i = (n->s.os_out_i + 1) & 0x7f;
        o_tick(n->x, oi2, oo);
        for (int c = 0; c < n->nout; c++) n->s.os_outv[c * 128 + n->s.os_out_i] = oo[c];
        for (int c = 0; c < n->nout; c++) out[c] = os_decimate(n->s.os_outv + c * 128, n->s.os_out_i);
        n->s.os_out_i = (n->s.os_out_i + 1) & 0x7f;
        break;
    }
    case O_ALLNEST: { /* delay.rs:344-352 */
        if (n->nin > 1) n->s.eta = in[1];
        float v = in[0] - n->s.eta * n->s.zz;
        float y = n->s.eta * v + n->s.zz;
        float z;
        o_tick(n->x, &v, &z);
        n->s.zz = z;
        out[0] = y;
        break;
    }
    case O_PLUCK: { /* oscillator.rs:287-305 */
        if (!n->s.pl_init) pluck_initialize_line(n);
        float o = n->s.pl_line[n->s.pl_pos] * n->s.pl_gain + in[0];
        o = fir_tick(n, o);                                   /* damping.filter_mono */
        float y0 = n->s.op_coeff * (o - n->s.op_y1) + n->s.op_x1; /* tuning.filter_mono (Allpole :320-324) */
        n->s.op_x1 = o; n->s.op_y1 = y0;
        o = y0;
        n->s.pl_line[n->s.pl_pos] = o;
        n->s.pl_pos += 1;
        if (n->s.pl_pos == n->s.pl_len) n->s.pl_pos = 0;
        out[0] = o;
        break;
    }
    case O_DSF: { /* oscillator.rs:172-187, dsf :105-113 */
        if (n->nin > 1) n->s.dsf_roughness = dsf_clamp_roughness(in[1]);
        n->s.phase += in[0] * n->s.sample_duration;
        n->s.phase -= floorf(n->s.phase);
        float nn = floorf(22050.0f / in[0] / n->s.dsf_spacing);
        float f = n->s.phase * F32_TAU, d = n->s.phase * F32_TAU * n->s.dsf_spacing, r = n->s.dsf_roughness;
        out[0] = (o_sinf(f) - r * o_sinf(f - d) - o_powf(r, nn + 1.0f) * (o_sinf(f + (nn + 1.0f) * d) - r * o_sinf(f + nn * d))) /
                 (1.0f + r * r - 2.0f * r * o_cosf(d));
        break;
    }
    case O_SHAPER: out[0] = shape_scalar(n, 0, in[0]); break; /* shape.rs:226-229 */
    case O_PHASE_OSC: {
        float phase = n->s.phase;
        float delta = in[0] * n->s.sample_duration;
        n->s.phase += delta;
        n->s.phase -= floorf(n->s.phase);
        if (n->s.osc_kind == O_OSC_RAMP) { /* :478-483 */
            out[0] = phase;
        } else if (n->s.osc_kind == O_OSC_POLYSAW) { /* :570-577 */
            out[0] = 2.0f * phase - 1.0f - polyblepf(phase, delta);
        } else { /* :646-659 / :729-739 */
            float width = n->s.osc_kind == O_OSC_POLYPULSE ? in[1] : 0.5f;
            float square = phase < width ? 1.0f : -1.0f;
            float half = phase - width;
            out[0] = square + polyblepf(phase, delta) - polyblepf(half - floorf(half), delta);
        }
        break;
    }
    case O_CHAOS:
        if (n->s.lorenz) { /* oscillator.rs:407-417 */
            float dx = 10.0f * (n->s.cy - n->s.cx);
            float dy = n->s.cx * (28.0f - n->s.cz) - n->s.cy;
            float dz = n->s.cx * n->s.cy - (8.0f / 3.0f) * n->s.cz;
            float dt = in[0] / n->s.sr;
            n->s.cx += dx * dt; n->s.cy += dy * dt; n->s.cz += dz * dt;
            out[0] = n->s.cx * 0.05107f;
        } else { /* oscillator.rs:348-358 */
            float dx = -n->s.cy - n->s.cz;
            float dy = n->s.cx + 0.15f * n->s.cy;
            float dz = 0.2f + n->s.cz * (n->s.cx - 10.0f);
            float dt = 2.91f * in[0] / n->s.sr;
            n->s.cx += dx * dt; n->s.cy += dy * dt; n->s.cz += dz * dt;
            out[0] = n->s.cx * 0.05757f;
        }
        break;
    case O_NLBIQUAD: {
        if (n->nin == 3) { /* biquad.rs:549-557 */
            float dc = in[1] - n->s.center, dq = in[2] - n->s.q;
            if (dc * dc + dq * dq != 0.0f) {
                n->s.center = in[1]; n->s.q = in[2];
                n->s.bc = bq_by_mode(n->s.nl_mode, n->s.sr, n->s.center, n->s.q, n->s.gain);
            }
        }
        if (n->nin == 4) { /* biquad.rs:558-572 */
            float dc = in[1] - n->s.center, dq = in[2] - n->s.q, dg = in[3] - n->s.gain;
            if (dc * dc + dq * dq + dg * dg != 0.0f) {
                n->s.center = in[1]; n->s.q = in[2]; n->s.gain = in[3];
                n->s.bc = bq_by_mode(n->s.nl_mode, n->s.sr, n->s.center, n->s.q, n->s.gain);
            }
        }
        float x0 = in[0];
        float y0 = n->s.bc.b0 * x0 + n->s.ns1;
        if (n->s.nl_dirty) { /* biquad.rs:789-796 */
            float n1 = shape_scalar(n, 0, n->s.ns2 + n->s.bc.b1 * x0 - y0 * n->s.bc.a1);
            float n2 = shape_scalar(n, 1, n->s.bc.b2 * x0 - y0 * n->s.bc.a2);
            n->s.ns1 = n1;
            n->s.ns2 = n2;
        } else { /* biquad.rs:577-581 */
            float fb = shape_scalar(n, 0, y0);
            n->s.ns1 = n->s.ns2 + n->s.bc.b1 * x0 - fb * n->s.bc.a1;
            n->s.ns2 = n->s.bc.b2 * x0 - fb * n->s.bc.a2;
        }
        out[0] = y0;
        break;
    }
    case O_REVERB_STEREO: {
        /* Feedback::new calls prevent_denormals() (feedback.rs:96; denormal.rs:18: MXCSR = 0x9fc0, FTZ + DAZ) on the
         * constructing thread, so a graph rendered on that thread runs flushed: do the same for this node. */
#if defined(__x86_64__) || defined(__i386__)
        unsigned int csr = _mm_getcsr();
        _mm_setcsr(0x9fc0);
#endif
        float o[32];
        for (int i = 0; i < 32; i++) {
            float x = in[i % 2] + n->s.rv_value[i];          /* MultiSplit<U2,U16> :600-602 ; Feedback::tick :131 */
            n->s.rv_buf[i][n->s.rv_i[i]] = x;                 /* Delay::tick delay.rs:116-124 */
            n->s.rv_i[i] += 1;
            if (n->s.rv_i[i] >= n->s.rv_len[i]) n->s.rv_i[i] = 0;
            float d = n->s.rv_buf[i][n->s.rv_i[i]];
            float *v = n->s.rv_v[i];                          /* Fir<U3>::tick fir.rs:57-70 */
            v[0] = v[1]; v[1] = v[2]; v[2] = d;
            float acc = 0.0f;
            for (int k = 0; k < 3; k++) acc += n->s.rv_w[k] * v[k];
            o[i] = acc;
        }
        float h[32];                                          /* FrameHadamard::frame feedback.rs:35-57 */
        for (int i = 0; i < 32; i++) h[i] = o[i];
        for (int hh = 1; hh < 32; hh *= 2)
            for (int i = 0; i < 32; i += hh * 2)
                for (int j = i; j < i + hh; j++) {
                    float x = h[j], y = h[j + hh];
                    h[j] = x + y;
                    h[j + hh] = x - y;
                }
        float scale = (float)(1.0 / sqrt(32.0));
        for (int i = 0; i < 32; i++) n->s.rv_value[i] = h[i] * scale;
        float l = 0.0f, r = 0.0f;                             /* Reduce::tick audionode.rs:2427-2439 (left fold) */
        for (int i = 0; i < 32; i++) {
            float pl = n->s.rv_wl[i] * o[i], pr = n->s.rv_wr[i] * o[i];
            if (i > 0) { l = l + pl; r = r + pr; } else { l = pl; r = pr; }
        }
        out[0] = l * (float)(1.0 / 16.0);                     /* * dc((1/16, 1/16)) */
        out[1] = r * (float)(1.0 / 16.0);
#if defined(__x86_64__) || defined(__i386__)
        _mm_setcsr(csr);
#endif
        break;
    }
    case O_PANNER: /* pan.rs:55-62 */
        if (n->nin > 1) pan_weights(in[1], &n->s.left_weight, &n->s.right_weight);
        out[0] = n->s.left_weight * in[0];
        out[1] = n->s.right_weight * in[0];
        break;
    case O_PIPE: /* audionode.rs:1441-1443 */
        o_tick(n->x, in, t);
        o_tick(n->y, t, out);
        break;
    case O_STACK: /* audionode.rs:1564-1577 */
        o_tick(n->x, in, out);
        o_tick(n->y, in + n->x->nin, out + n->x->nout);
        break;
    case O_BINOP: /* audionode.rs:926-931 */
        o_tick(n->x, in, t);
        o_tick(n->y, in + n->x->nin, out);
        for (int i = 0; i < n->nout; i++) out[i] = binop_apply(n->op, t[i], out[i]);
        break;
    case O_MULTIPASS: for (int i = 0; i < n->nout; i++) out[i] = in[i]; break; /* audionode.rs:388-390 */
    case O_SINK: break;
    case O_SPLIT: for (int i = 0; i < n->nout; i++) out[i] = in[i % n->jm]; break; /* :551-553, :597-599 */
    case O_JOIN: /* :638-644, :700-708: sum, then divide */
        for (int j = 0; j < n->jm; j++) {
            float o = in[j];
            for (int i = 1; i < n->jn; i++) o += in[j + i * n->jm];
            out[j] = o / (float)n->jn;
        }
        break;
    case O_REVERSE: for (int i = 0; i < n->nout; i++) out[i] = in[n->nout - 1 - i]; break; /* :2823-2825 */
    case O_IMPULSE: /* :2864-2868 */
        for (int i = 0; i < n->nout; i++) out[i] = n->s.value[0];
        n->s.value[0] = 0.0f;
        break;
    case O_MAP: n->map_fn(in, out, n->map_ctx); break; /* :1363-1365 */
    case O_DECLICK: /* dynamics.rs:278-287 */
        if (n->s.dc_t < n->s.dc_duration) {
            float phase = (n->s.dc_t - 0.0f) / (n->s.dc_duration - 0.0f);
            float value = ((phase * 6.0f - 15.0f) * phase + 10.0f) * phase * phase * phase;
            n->s.dc_t += n->s.dc_sd;
            out[0] = in[0] * value;
        } else {
            out[0] = in[0];
        }
        break;
    case O_BRANCH: /* :1716-1726 */
        o_tick(n->x, in, out);
        o_tick(n->y, in, out + n->x->nout);
        break;
    case O_BUS: /* :1861-1865 */
        o_tick(n->x, in, out);
        o_tick(n->y, in, t);
        for (int i = 0; i < n->nout; i++) out[i] = out[i] + t[i];
        break;
    case O_FEEDBACK: { /* feedback.rs:129-134, 270-275 */
        float fi[O_MAX_CH];
        for (int i = 0; i < n->nin; i++) fi[i] = in[i] + n->fb_value[i];
        o_tick(n->x, fi, out);
        if (n->y) o_tick(n->y, out, n->fb_value);
        else for (int i = 0; i < n->nout; i++) n->fb_value[i] = out[i];
        if (n->hadamard) { /* FrameHadamard::frame :35-57 */
            float *o = n->fb_value;
            for (int h = 1; h < n->nout; h *= 2)
                for (int i = 0; i < n->nout; i += h * 2)
                    for (int j = i; j < i + h; j++) {
                        float a = o[j], b = o[j + h];
                        o[j] = a + b;
                        o[j + h] = a - b;
                    }
            float z = (float)(1.0 / sqrt((double)n->nout));
            for (int i = 0; i < n->nout; i++) o[i] = o[i] * z;
        }
        break;
    }
    case O_THRU: /* :1977-1986 */
        o_tick(n->x, in, t);
        for (int i = 0; i < n->nout; i++) out[i] = i < n->x->nout ? t[i] : in[i];
        break;
    case O_MULTI: {
        int xi = n->kids[0]->nin, xo = n->kids[0]->nout;
        switch (n->multi) {
        case O_MULTI_BUS: /* :2117-2121: fold from a zero frame */
            for (int c = 0; c < xo; c++) out[c] = 0.0f;
            for (int i = 0; i < n->nkids; i++) {
                o_tick(n->kids[i], in, t);
                for (int c = 0; c < xo; c++) out[c] = out[c] + t[c];
            }
            break;
        case O_MULTI_STACK: /* :2282-2291 */
            for (int i = 0; i < n->nkids; i++) o_tick(n->kids[i], in + i * xi, out + i * xo);
            break;
        case O_MULTI_BRANCH: /* :2590-2598 */
            for (int i = 0; i < n->nkids; i++) o_tick(n->kids[i], in, out + i * xo);
            break;
        case O_MULTI_REDUCE: /* :2430-2442 */
            o_tick(n->kids[0], in, out);
            for (int i = 1; i < n->nkids; i++) {
                o_tick(n->kids[i], in + i * xi, t);
                for (int c = 0; c < xo; c++) out[c] = binop_apply(n->op, out[c], t[c]);
            }
            break;
        case O_MULTI_CHAIN: /* :2728-2734 */
            o_tick(n->kids[0], in, out);
            for (int i = 1; i < n->nkids; i++) {
                for (int c = 0; c < xo; c++) t[c] = out[c];
                o_tick(n->kids[i], t, out);
            }
            break;
        }
        break;
    }
    case O_UNOP: /* audionode.rs:1268-1270 */
        o_tick(n->x, in, out);
        for (int i = 0; i < n->nout; i++) out[i] = unop_apply(n->op, out[i], n->scalar);
        break;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* process (block path).  in/out are planar [channel][64].                                                */
/* ------------------------------------------------------------------------------------------------------ */

static inline int simd_items(int samples) { return (samples + 7) >> 3; } /* lib.rs:77-79 */
static inline int full_simd_items(int samples) { return samples >> 3; }  /* lib.rs:83-85 */

/* AudioNode::process default fallback audionode.rs:85-105 */
static void process_via_tick(onode *n, int size, const float *in, float *out) {
    float fi[O_MAX_CH], fo[O_MAX_CH];
    for (int i = 0; i < size; i++) {
        for (int c = 0; c < n->nin; c++) fi[c] = in[c * MAXB + i];
        o_tick(n, fi, fo);
        for (int c = 0; c < n->nout; c++) out[c * MAXB + i] = fo[c];
    }
}
/* AudioNode::process_remainder audionode.rs:110-126 */
static void process_remainder(onode *n, int size, const float *in, float *out) {
    float fi[O_MAX_CH], fo[O_MAX_CH];
    for (int i = size & ~7; i < size; i++) {
        for (int c = 0; c < n->nin; c++) fi[c] = in[c * MAXB + i];
        o_tick(n, fi, fo);
        for (int c = 0; c < n->nout; c++) out[c * MAXB + i] = fo[c];
    }
}

/* WaveSynth::process wavetable.rs:327-348: 8 phases accumulated, vector floor wrap (wide's inherent f32x8::floor = true floor), table
 * pair chosen from LANE 0's frequency for the item.  A function of its own: the tree walk (o_process) and the monomorphised config-4
 * voice (o_c4_block, the cpu_baseline leg) run the same code. */
static void process_remainder(onode *n, int size, const float *in, float *out);
static inline void wavesynth_process(onode *n, int size, const float *in, float *out) {
    float phase = n->s.phase;
    size_t hint = n->s.table_hint;
    for (int i = 0; i < full_simd_items(size); i++) {
        float frequency = in[i << 3];
        float ph[SIMD_N];
        for (int j = 0; j < SIMD_N; j++) {
            phase += in[(i << 3) + j] * n->s.sample_duration;
            ph[j] = phase;
        }
        for (int j = 0; j < SIMD_N; j++) ph[j] = ph[j] - floorf(ph[j]);
        size_t table = wt_table_index(n->s.wt, hint, fabsf(frequency));
        float w = clamp01f((fabsf(frequency) - n->s.wt->pitch[table]) /
                           (n->s.wt->pitch[table + 1] - n->s.wt->pitch[table]));
        for (int j = 0; j < SIMD_N; j++)
            out[(i << 3) + j] = (1.0f - w) * wt_at(n->s.wt, table + 1, ph[j]) + w * wt_at(n->s.wt, table + 2, ph[j]);
        hint = table;
        if (n->nout > 1)
            for (int j = 0; j < SIMD_N; j++) out[MAXB + (i << 3) + j] = ph[j];
    }
    n->s.phase = phase - floorf(phase);
    n->s.table_hint = hint;
    process_remainder(n, size, in, out);
}
/* EnvelopeIn::process envelope.rs:315-340 for adsr_live: whole-block segment walk (no remainder path) */
static inline void adsr_live_process(onode *n, int size, const float *in, float *out) {
    if (size == 0) return;
    if (n->s.et >= n->s.et1) env_next_segment(n, in[0]);
    int i = 0;
    while (i < size) {
        int64_t left = (int64_t)ceilf((n->s.et1 - n->s.et) / n->s.esd);
        size_t segment_samples_left = (size_t)left;
        size_t loop_samples = (size_t)(size - i) < segment_samples_left ? (size_t)(size - i) : segment_samples_left;
        float value = n->s.ev, delta = n->s.evd;
        for (size_t k = 0; k < loop_samples; k++) {
            out[i + (int)k] = value;
            value += delta;
        }
        n->s.ev = value;
        i += (int)loop_samples;
        n->s.et += (float)(int64_t)loop_samples * n->s.esd;
        if (loop_samples == segment_samples_left && i < size) env_next_segment(n, in[i]);
    }
}

void o_process(onode *n, int size, const float *in, float *out) {
    if (n->ftz && !g_ftz_on) {
        FTZ_ENTER;
        o_process(n, size, in, out);
        FTZ_LEAVE;
        return;
    }
    switch (n->type) {
    case O_OVERSAMPLE: { /* oversample.rs:178-212.  Two passes of size / 2 outer samples; the inner node processes `size`
                          * inner samples per pass.  An odd `size` leaves the last outer sample untouched (the reference
                          * never writes it).  Deviation, documented in DESIGN.md: the reference's decimation loop runs
                          * over Inputs::USIZE channels (:200), so a generator (0 inputs) is never written at all and a
                          * node with more inputs than outputs indexes out of range; the loop below runs over the
                          * OUTPUT channels, which is identical whenever inputs == outputs. */
        float ii[O_MAX_CH * MAXB] = {0}, io[O_MAX_CH * MAXB]; /* BufferArray::new() :179 is zero-initialised -- and it matters: for an odd `size` the
                                                                * inner block's last sample (index 2 * (size / 2)) is never interpolated, the inner
                                                                * node reads the zero and advances by it */
        const int offs[2] = {0, size / 2};
        for (int pass = 0; pass < 2; pass++) {
            int offset = offs[pass];
            for (int i = 0; i < size / 2; i++) {
                for (int c = 0; c < n->nin; c++) {
                    n->s.os_inv[c * 128 + n->s.os_in_i] = in[c * MAXB + i + offset];
                    os_interpolate(n->s.os_inv + c * 128, n->s.os_in_i, &ii[c * MAXB + i * 2], &ii[c * MAXB + i * 2 + 1]);
                }
                n->s.os_in_i = (n->s.os_in_i + 1) & 0x7f;
            }
            o_process(n->x, size, ii, io);
            for (int i = 0; i < size / 2; i++) {
                for (int c = 0; c < n->nout; c++) {
                    n->s.os_outv[c * 128 + n->s.os_out_i] = io[c * MAXB + i * 2];
                    size_t next = (n->s.os_out_i + 1) & 0x7f;
                    n->s.os_outv[c * 128 + next] = io[c * MAXB + i * 2 + 1];
                    out[c * MAXB + i + offset] = os_decimate(n->s.os_outv + c * 128, next);
                }
                n->s.os_out_i = (n->s.os_out_i + 2) & 0x7f;
            }
        }
        break;
    }
    case O_VAR: /* Var::process shared.rs:122-125: ONE read of the shared value per block, splat over simd_items(size)
                 * (VarFn has no process override: shared.rs:136-184 -> the default per-sample walk below) */
        if (n->map_fn) { process_via_tick(n, size, in, out); break; }
        for (int j = 0; j < simd_items(size) * 8; j++) out[j] = n->s.value[0];
        break;
    case O_CONSTANT: /* audionode.rs:501-508: splat over simd_items(size) */
        for (int c = 0; c < n->nout; c++)
            for (int j = 0; j < simd_items(size) * 8; j++) out[c * MAXB + j] = n->s.value[c];
        break;
    case O_SINE: { /* oscillator.rs:74-86: phase unwrapped across the block, wide sin, one wrap at the end */
        float phase = n->s.phase;
        for (int i = 0; i < full_simd_items(size); i++) {
            float element[SIMD_N];
            for (int j = 0; j < SIMD_N; j++) {
                element[j] = phase;
                phase += in[(i << 3) + j] * n->s.sample_duration;
            }
            for (int j = 0; j < SIMD_N; j++) out[(i << 3) + j] = o_wide_sinf(element[j] * F32_TAU);
        }
        n->s.phase = phase - floorf(phase);
        process_remainder(n, size, in, out);
        break;
    }
    case O_NOISE: { /* noise.rs:204-218: writes simd_items(size)*8 samples, advances state by size */
        uint32_t state = n->s.nstate;
        for (int i = 0; i < simd_items(size) * 8; i++)
            out[i] = (float)(o_hash32x(state + (uint32_t)i + 1u) >> 8) * (2.0f / (float)((1 << 24) - 1)) - 1.0f;
        n->s.nstate = state + (uint32_t)size;
        break;
    }
    case O_PIPE: /* audionode.rs:1445-1449 */
        o_process(n->x, size, in, n->tmp);
        o_process(n->y, size, n->tmp, out);
        break;
    case O_STACK: /* audionode.rs:1580-1591 */
        o_process(n->x, size, in, out);
        o_process(n->y, size, in + n->x->nin * MAXB, out + n->x->nout * MAXB);
        break;
    case O_BINOP: /* audionode.rs:933-955 */
        o_process(n->x, size, in, n->tmp);
        o_process(n->y, size, in + n->x->nin * MAXB, out);
        for (int c = 0; c < n->nout; c++)
            for (int i = 0; i < simd_items(size) * 8; i++)
                out[c * MAXB + i] = binop_apply(n->op, n->tmp[c * MAXB + i], out[c * MAXB + i]);
        break;
    case O_MULTIPASS: /* audionode.rs:391-397 */
    case O_SPLIT:     /* :554-560, :600-606 */
    case O_REVERSE:   /* :2826-2832 */
        for (int c = 0; c < n->nout; c++) {
            int src = n->type == O_MULTIPASS ? c : n->type == O_SPLIT ? c % n->jm : n->nout - 1 - c;
            for (int i = 0; i < simd_items(size) * 8; i++) out[c * MAXB + i] = in[src * MAXB + i];
        }
        break;
    case O_SINK: break; /* :454 */
    case O_WRAP: o_process(n->x, size, in, out); break; /* wavetable.rs:475-477 */
    case O_DECLICK: { /* dynamics.rs:289-307 */
        for (int i = 0; i < simd_items(size) * 8; i++) out[i] = in[i];
        if (n->s.dc_t < n->s.dc_duration) {
            float phase = (n->s.dc_t - 0.0f) / (n->s.dc_duration - 0.0f);
            float phase_d = n->s.dc_sd / n->s.dc_duration;
            float end_time = n->s.dc_t + (float)(long long)size * n->s.dc_sd;
            int end_index = n->s.dc_duration < end_time
                                ? (int)(long long)ceilf((n->s.dc_duration - n->s.dc_t) / n->s.dc_sd) : size;
            if (end_index > MAXB) end_index = MAXB; /* the reference would panic on the slice */
            for (int i = 0; i < end_index; i++) {
                out[i] *= ((phase * 6.0f - 15.0f) * phase + 10.0f) * phase * phase * phase;
                phase += phase_d;
            }
            n->s.dc_t = end_time;
        }
        break;
    }
    case O_JOIN: { /* :649-659, :710-724: every term scaled by z = 1/N, then summed */
        float z = 1.0f / (float)n->jn;
        for (int c = 0; c < n->jm; c++)
            for (int i = 0; i < simd_items(size) * 8; i++) out[c * MAXB + i] = in[c * MAXB + i] * z;
        for (int c = n->jm; c < n->jm * n->jn; c++)
            for (int i = 0; i < simd_items(size) * 8; i++) out[(c % n->jm) * MAXB + i] += in[c * MAXB + i] * z;
        break;
    }
    case O_BRANCH: /* :1728-1736 */
        o_process(n->x, size, in, out);
        o_process(n->y, size, in, out + n->x->nout * MAXB);
        break;
    case O_BUS: /* :1867-1876 */
        o_process(n->x, size, in, out);
        o_process(n->y, size, in, n->tmp);
        for (int c = 0; c < n->nout; c++)
            for (int i = 0; i < simd_items(size) * 8; i++) out[c * MAXB + i] += n->tmp[c * MAXB + i];
        break;
    case O_THRU: /* :1988-2011 */
        if (n->nin == 0) break;
        if (n->x->nin < n->x->nout) {
            o_process(n->x, size, in, n->tmp);
            for (int c = 0; c < n->nin; c++)
                for (int i = 0; i < simd_items(size) * 8; i++) out[c * MAXB + i] = n->tmp[c * MAXB + i];
        } else {
            o_process(n->x, size, in, out);
            for (int c = n->x->nout; c < n->nin; c++)
                for (int i = 0; i < simd_items(size) * 8; i++) out[c * MAXB + i] = in[c * MAXB + i];
        }
        break;
    case O_MULTI: {
        int xi = n->kids[0]->nin, xo = n->kids[0]->nout, items = simd_items(size) * 8;
        switch (n->multi) {
        case O_MULTI_BUS: /* :2123-2134 */
            o_process(n->kids[0], size, in, out);
            for (int k = 1; k < n->nkids; k++) {
                o_process(n->kids[k], size, in, n->tmp);
                for (int c = 0; c < xo; c++)
                    for (int i = 0; i < items; i++) out[c * MAXB + i] += n->tmp[c * MAXB + i];
            }
            break;
        case O_MULTI_STACK: /* :2293-2305 */
            for (int k = 0; k < n->nkids; k++) o_process(n->kids[k], size, in + k * xi * MAXB, out + k * xo * MAXB);
            break;
        case O_MULTI_BRANCH: /* :2600-2610 */
            for (int k = 0; k < n->nkids; k++) o_process(n->kids[k], size, in, out + k * xo * MAXB);
            break;
        case O_MULTI_REDUCE: /* :2443-2464 */
            o_process(n->kids[0], size, in, out);
            for (int k = 1; k < n->nkids; k++) {
                o_process(n->kids[k], size, in + k * xi * MAXB, n->tmp);
                for (int c = 0; c < xo; c++)
                    for (int i = 0; i < items; i++)
                        out[c * MAXB + i] = binop_apply(n->op, out[c * MAXB + i], n->tmp[c * MAXB + i]);
            }
            break;
        case O_MULTI_CHAIN: { /* :2736-2750: ping-pong between the output and one scratch buffer */
            float *a = n->tmp, *b = n->tmp2;
            o_process(n->kids[0], size, in, a);
            for (int k = 1; k < n->nkids; k++) {
                o_process(n->kids[k], size, a, b);
                float *sw = a; a = b; b = sw;
            }
            for (int c = 0; c < xo; c++)
                for (int i = 0; i < items; i++) out[c * MAXB + i] = a[c * MAXB + i];
            break;
        }
        }
        break;
    }
    case O_UNOP: /* audionode.rs:1273-1280 */
        o_process(n->x, size, in, out);
        for (int c = 0; c < n->nout; c++)
            for (int i = 0; i < simd_items(size) * 8; i++)
                out[c * MAXB + i] = unop_apply(n->op, out[c * MAXB + i], n->scalar);
        break;
    case O_WAVESYNTH: wavesynth_process(n, size, in, out); break;
    case O_SHAPER: /* shape.rs:235-240: Shape::simd on full items, tick for the remainder */
        for (int i = 0; i < full_simd_items(size) * 8; i++) out[i] = shape_simd_lane(n, 0, in[i]);
        process_remainder(n, size, in, out);
        break;
    case O_ENVELOPE_IN: { /* envelope.rs:315-340 */
        if (size == 0) break;
        float fr[O_MAX_CH];
        for (int c = 0; c < n->nin; c++) fr[c] = in[c * MAXB];
        if (n->s.et >= n->s.et1) envin_next_segment(n, fr);
        int i = 0;
        while (i < size) {
            int64_t left = (int64_t)ceilf((n->s.et1 - n->s.et) / n->s.esd);
            size_t segment_samples_left = (size_t)left;
            size_t loop_samples = (size_t)(size - i) < segment_samples_left ? (size_t)(size - i) : segment_samples_left;
            for (int c = 0; c < n->nout; c++) {
                float value = n->s.env_val[c], delta = n->s.env_d[c];
                for (size_t k = 0; k < loop_samples; k++) {
                    out[c * MAXB + i + (int)k] = value;
                    value += delta;
                }
                n->s.env_val[c] = value;
            }
            i += (int)loop_samples;
            n->s.et += (float)(int64_t)loop_samples * n->s.esd;
            if (loop_samples == segment_samples_left && i < size) {
                for (int c = 0; c < n->nin; c++) fr[c] = in[c * MAXB + i];
                envin_next_segment(n, fr);
            }
        }
        break;
    }
    case O_ENVELOPE: { /* envelope.rs:138-163 */
        if (n->s.et >= n->s.et1) envelope_next_segment(n);
        int i = 0;
        while (i < size) {
            int64_t left = (int64_t)ceilf((n->s.et1 - n->s.et) / n->s.esd);
            size_t segment_samples_left = (size_t)left;
            size_t loop_samples = (size_t)(size - i) < segment_samples_left ? (size_t)(size - i) : segment_samples_left;
            for (int c = 0; c < n->nout; c++) {
                float value = n->s.env_val[c], delta = n->s.env_d[c];
                for (size_t k = 0; k < loop_samples; k++) {
                    out[c * MAXB + i + (int)k] = value;
                    value += delta;
                }
                n->s.env_val[c] = value;
            }
            i += (int)loop_samples;
            n->s.et += (float)(int64_t)loop_samples * n->s.esd;
            if (loop_samples == segment_samples_left) envelope_next_segment(n);
        }
        break;
    }
    case O_ADSR_LIVE: adsr_live_process(n, size, in, out); break;
    case O_PANNER: /* pan.rs:63-76 */
        if (n->nin == 1) {
            for (int i = 0; i < simd_items(size) * 8; i++) {
                out[i] = in[i] * n->s.left_weight;
                out[MAXB + i] = in[i] * n->s.right_weight;
            }
        } else {
            for (int i = 0; i < size; i++) {
                pan_weights(in[MAXB + i], &n->s.left_weight, &n->s.right_weight);
                out[i] = in[i] * n->s.left_weight;
                out[MAXB + i] = in[i] * n->s.right_weight;
            }
        }
        break;
    case O_FIXED_SVF: /* no override -> tick fallback; inlined here so the CPU baseline is not penalised */
        for (int i = 0; i < size; i++) out[i] = svf_tick(n, in[i]);
        break;
    case O_BIQUAD:
        for (int i = 0; i < size; i++) out[i] = biquad_tick(n, in[i]);
        break;
    default: /* every other leaf inherits the per-sample fallback */
        process_via_tick(n, size, in, out);
        break;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* executors                                                                                              */
/* ------------------------------------------------------------------------------------------------------ */

/* Wave::render src/wave.rs:441-466 for a generator (0 inputs).  out is [channels][length]. Returns length. */
size_t o_wave_render(onode *n, double sample_rate, double duration, float *out, size_t capacity) {
    if (n->nin != 0 || n->nout <= 0 || duration < 0.0) return 0;
    o_set_sample_rate(n, sample_rate);
    size_t length = (size_t)round(duration * sample_rate);
    if (length > capacity) return 0;
    float *buffer = (float *)malloc((size_t)n->nout * MAXB * sizeof(float));
    size_t i = 0;
    while (i < length) {
        int nn = (int)((length - i) < MAXB ? (length - i) : MAXB);
        o_process(n, nn, NULL, buffer);
        for (int c = 0; c < n->nout; c++)
            for (int j = 0; j < nn; j++) out[(size_t)c * length + i + j] = buffer[c * MAXB + j];
        i += (size_t)nn;
    }
    free(buffer);
    return length;
}

/* Wave::filter-style executor (wave.rs:518-565 shape): chops [channels][length] input into <=64 blocks and
 * calls process; `block` lets tests use other chunkings (the reference always uses 64). */
void o_render_blocks(onode *n, size_t length, int block, const float *in, float *out) {
    float *bi = (float *)calloc((size_t)(n->nin ? n->nin : 1) * MAXB, sizeof(float));
    float *bo = (float *)calloc((size_t)n->nout * MAXB, sizeof(float));
    size_t i = 0;
    if (block <= 0 || block > MAXB) block = MAXB;
    while (i < length) {
        int nn = (int)((length - i) < (size_t)block ? (length - i) : (size_t)block);
        for (int c = 0; c < n->nin; c++)
            for (int j = 0; j < nn; j++) bi[c * MAXB + j] = in[(size_t)c * length + i + j];
        memset(bo, 0, (size_t)n->nout * MAXB * sizeof(float)); /* samples a node leaves unwritten (Oversampler, odd size) read 0 */
        o_process(n, nn, bi, bo);
        for (int c = 0; c < n->nout; c++)
            for (int j = 0; j < nn; j++) out[(size_t)c * length + i + j] = bo[c * MAXB + j];
        i += (size_t)nn;
    }
    free(bi);
    free(bo);
}

/* per-sample executor: length ticks; in [channels][length], out [channels][length] */
void o_render_ticks(onode *n, size_t length, const float *in, float *out) {
    float fi[O_MAX_CH], fo[O_MAX_CH];
    for (size_t i = 0; i < length; i++) {
        for (int c = 0; c < n->nin; c++) fi[c] = in[(size_t)c * length + i];
        o_tick(n, fi, fo);
        for (int c = 0; c < n->nout; c++) out[(size_t)c * length + i] = fo[c];
    }
}


/* ---- state of a config-3 voice, for the monomorphic CPU baseline (o_fast.c) -------------------------------------
 * g = Pipe(Pipe(Unop(+f, Unop(*m, Unop(*f, Pipe(Constant, Sine)))), Sine), FixedSvf), as o_bank.c builds it.
 * Returns 0 and fills the registers a hand-monomorphised process() keeps, or -1 if the tree has another shape. */
int o_fm_svf_state(const onode *g, o_fm_svf_regs *r) {
    if (!g || g->type != O_PIPE || !g->x || g->x->type != O_PIPE || !g->y || g->y->type != O_FIXED_SVF) return -1;
    const onode *svf = g->y, *car = g->x->y, *e = g->x->x;
    if (!car || car->type != O_SINE || !e || e->type != O_UNOP || e->op != O_ADD_SCALAR) return -1;
    const onode *e1 = e->x;
    if (!e1 || e1->type != O_UNOP || e1->op != O_MUL_SCALAR) return -1;
    const onode *e2 = e1->x;
    if (!e2 || e2->type != O_UNOP || e2->op != O_MUL_SCALAR) return -1;
    const onode *mod = e2->x;
    if (!mod || mod->type != O_PIPE || !mod->x || mod->x->type != O_CONSTANT || !mod->y || mod->y->type != O_SINE) return -1;
    r->f_const = mod->x->s.value[0];
    r->mul_f = e2->scalar;
    r->mul_m = e1->scalar;
    r->add_f = e->scalar;
    r->mod_phase = mod->y->s.phase;
    r->mod_sd = mod->y->s.sample_duration;
    r->car_phase = car->s.phase;
    r->car_sd = car->s.sample_duration;
    r->a1 = svf->s.sc.a1; r->a2 = svf->s.sc.a2; r->a3 = svf->s.sc.a3;
    r->m0 = svf->s.sc.m0; r->m1 = svf->s.sc.m1; r->m2 = svf->s.sc.m2;
    r->ic1eq = svf->s.ic1eq; r->ic2eq = svf->s.ic2eq;
    return 0;
}

/* ---- monomorphised voices of BASELINE configs 4 and 5, for the cpu_baseline legs (o_fast.c) -------------------------------------
 * FunDSP graphs are statically typed: rustc compiles `((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * ENV >> pan(p)` into ONE process()
 * per block -- every node's process / tick inlined, buffers on the stack, no tree, no dispatch.  These functions are that form, made
 * of the SAME node functions the tree walk above calls (wavesynth_process, moog_tick, adsr_live_process, pan weights), so they are
 * bit-identical to o_process on the same graph by construction (asserted in tests/test_oracle_fast.py); built -O3 -march=native by
 * `make native`.  ENV = adsr_live(..) fed by the graph's input (the stream-gate shape) or var(gate) >> adsr_live(..) (the reference's
 * own shape, examples/live_adsr.rs:72). */
int o_c4_open(onode *g, o_c4_voice *v) {
    memset(v, 0, sizeof *v);
    if (!g || g->type != O_PIPE || !g->x || g->x->type != O_BINOP || g->x->op != O_MUL || !g->y || g->y->type != O_PANNER || g->y->nin != 1) return -1;
    onode *sm = g->x->x, *env = g->x->y;
    if (!sm || sm->type != O_PIPE || !sm->y || sm->y->type != O_MOOG || sm->y->nin != 3 || !sm->x || sm->x->type != O_STACK) return -1;
    onode *st = sm->x; /* Stack(Stack(Pipe(Constant, WaveSynth), Constant), Constant) */
    if (!st->x || st->x->type != O_STACK || !st->y || st->y->type != O_CONSTANT || st->y->nout != 1) return -1;
    onode *st2 = st->x;
    if (!st2->x || st2->x->type != O_PIPE || !st2->y || st2->y->type != O_CONSTANT || st2->y->nout != 1) return -1;
    onode *osc = st2->x;
    if (!osc->x || osc->x->type != O_CONSTANT || osc->x->nout != 1 || !osc->y || osc->y->type != O_WAVESYNTH || osc->y->nout != 1) return -1;
    if (env && env->type == O_PIPE) { /* var(gate) >> adsr_live */
        if (!env->x || env->x->type != O_VAR || env->x->map_fn || !env->y || env->y->type != O_ADSR_LIVE) return -1;
        v->var = env->x;
        v->env = env->y;
    } else if (env && env->type == O_ADSR_LIVE) {
        v->env = env;
    } else {
        return -1;
    }
    v->g = g; v->saw = osc->y; v->moog = sm->y; v->pan = g->y;
    v->f = osc->x->s.value[0]; v->fc = st2->y->s.value[0]; v->q = st->y->s.value[0];
    return 0;
}
/* one block (size <= 64): gate = the graph's input block (stream-gate shape; ignored when the voice has a Var); out = [2][64] */
void o_c4_block(o_c4_voice *v, int size, const float *gate, float *out) {
    float fr[MAXB], osc[MAXB], lad[MAXB], gt[MAXB], env[MAXB];
    const int items8 = simd_items(size) * 8;
    for (int i = 0; i < items8; i++) fr[i] = v->f;                      /* Constant::process audionode.rs:501-508 */
    wavesynth_process(v->saw, size, fr, osc);                             /* WaveSynth::process wavetable.rs:327-348 */
    for (int i = 0; i < size; i++) {                                      /* Stack -> Moog: default process = tick per sample (audionode.rs:85-105) */
        const float in3[3] = {osc[i], v->fc, v->q};
        lad[i] = moog_tick(v->moog, in3);
    }
    for (int i = size; i < items8; i++) lad[i] = 0.0f;
    if (v->var) {
        const float value = v->var->s.value[0];                           /* Var::process shared.rs:122-125: one read, splat */
        for (int i = 0; i < items8; i++) gt[i] = value;
        gate = gt;
    }
    adsr_live_process(v->env, size, gate, env);                           /* EnvelopeIn::process envelope.rs:315-340 */
    for (int i = size; i < items8; i++) env[i] = 0.0f;
    const float wl = v->pan->s.left_weight, wr = v->pan->s.right_weight;  /* Binop::process audionode.rs:933-955, Panner::process pan.rs:63-76 */
    for (int i = 0; i < items8; i++) {
        const float x = lad[i] * env[i];
        out[i] = x * wl;
        out[MAXB + i] = x * wr;
    }
}

/* reverb_stereo (prelude.rs:1739-1775: Feedback<U32, 32 x (delay >> fir3), FrameHadamard> + pan fold), one BLOCK per call: the arithmetic
 * of the O_REVERB_STEREO tick above, frame by frame, with the MXCSR switch (Feedback::new -> prevent_denormals, denormal.rs:18) once per
 * block instead of twice per sample and the 32 lines as plain arrays the compiler can keep in vector registers.  in / out = [2][64]. */
void o_reverb_stereo_block(onode *n, int size, const float *in, float *out) {
#if defined(__x86_64__) || defined(__i386__)
    const unsigned int csr = _mm_getcsr();
    _mm_setcsr(0x9fc0);
#endif
    const float scale = (float)(1.0 / sqrt(32.0)), w0 = n->s.rv_w[0], w1 = n->s.rv_w[1], w2 = n->s.rv_w[2];
    float value[32], v0[32], v1[32], v2[32];
    size_t pos[32];
    for (int i = 0; i < 32; i++) {
        value[i] = n->s.rv_value[i];
        v0[i] = n->s.rv_v[i][0]; v1[i] = n->s.rv_v[i][1]; v2[i] = n->s.rv_v[i][2];
        pos[i] = n->s.rv_i[i];
    }
    for (int t = 0; t < size; t++) {
        float o[32], h[32];
        const float x0 = in[t], x1 = in[MAXB + t];
        for (int i = 0; i < 32; i++) {
            const float x = ((i & 1) ? x1 : x0) + value[i];
            n->s.rv_buf[i][pos[i]] = x;
            pos[i] += 1;
            if (pos[i] >= n->s.rv_len[i]) pos[i] = 0;
            v0[i] = v1[i]; v1[i] = v2[i]; v2[i] = n->s.rv_buf[i][pos[i]];
        }
        for (int i = 0; i < 32; i++) {
            float acc = 0.0f;
            acc += w0 * v0[i];
            acc += w1 * v1[i];
            acc += w2 * v2[i];
            o[i] = acc;
            h[i] = acc;
        }
        for (int hh = 1; hh < 32; hh *= 2)
            for (int i = 0; i < 32; i += hh * 2)
                for (int j = i; j < i + hh; j++) {
                    const float x = h[j], y = h[j + hh];
                    h[j] = x + y;
                    h[j + hh] = x - y;
                }
        for (int i = 0; i < 32; i++) value[i] = h[i] * scale;
        float l = n->s.rv_wl[0] * o[0], r = n->s.rv_wr[0] * o[0];
        for (int i = 1; i < 32; i++) {
            l = l + n->s.rv_wl[i] * o[i];
            r = r + n->s.rv_wr[i] * o[i];
        }
        out[t] = l * (float)(1.0 / 16.0);
        out[MAXB + t] = r * (float)(1.0 / 16.0);
    }
    for (int i = 0; i < 32; i++) {
        n->s.rv_value[i] = value[i];
        n->s.rv_v[i][0] = v0[i]; n->s.rv_v[i][1] = v1[i]; n->s.rv_v[i][2] = v2[i];
        n->s.rv_i[i] = pos[i];
    }
#if defined(__x86_64__) || defined(__i386__)
    _mm_setcsr(csr);
#endif
}
int o_is_reverb_stereo(const onode *n) { return n && n->type == O_REVERB_STEREO; }

/* reverb3_stereo(time, diffusion, lowpole_hz(cutoff)) (reverb.rs:241-272), one BLOCK per call, MONOMORPHISED: the O_REVERB3 tick above with its 76
 * AllNest / Delay / Lowpole ticks written out on the nodes' own state (static dispatch, everything inlined) -- what rustc makes of Reverb<Lowpole>.
 * Bit-equal to the tree walk (tests/test_oracle_fast.py).  Only for loop filters that are fixed-cutoff lowpoles; in / out = [2][64]. */
static inline float rv3_allnest(onode *a, float x) { /* AllNest<U1, Delay>::tick delay.rs:344-352 + Delay::tick :116-124 */
    onode *d = a->x;
    const float v = x - a->s.eta * a->s.zz;
    const float y = a->s.eta * v + a->s.zz;
    d->s.dbuf[d->s.di] = v;
    d->s.di += 1;
    if (d->s.di >= d->s.dlen) d->s.di = 0;
    a->s.zz = d->s.dbuf[d->s.di];
    return y;
}
static inline float rv3_delay(onode *d, float x) {
    d->s.dbuf[d->s.di] = x;
    d->s.di += 1;
    if (d->s.di >= d->s.dlen) d->s.di = 0;
    return d->s.dbuf[d->s.di];
}
static inline float rv3_lowpole(onode *f, float x) { /* filter.rs:64-66 */
    const float c = f->s.op_coeff;
    f->s.op_y1 = (1.0f - c) * x + c * f->s.op_y1;
    return f->s.op_y1;
}
int o_reverb3_block_ok(const onode *n) {
    if (!n || n->type != O_REVERB3) return 0;
    for (int b = 0; b < 8; b++)
        for (int j = 8; j < 10; j++) {
            const onode *f = n->kids[b * 11 + j];
            if (f->type != O_ONEPOLE || f->s.op_kind != O_OP_LOWPOLE || f->nin != 1) return 0;
        }
    return 1;
}
void o_reverb3_block(onode *n, int size, const float *in, float *out) {
    const float a = n->rv3_a;
    float v0 = n->rv3_feedback;
    for (int t = 0; t < size; t++) {
        float input0 = rv3_allnest(n->pre[0], in[t] * 0.5f);
        input0 = rv3_allnest(n->pre[1], input0);
        float input1 = rv3_allnest(n->pre[2], in[MAXB + t] * 0.5f);
        input1 = rv3_allnest(n->pre[3], input1);
        float o0 = 0.0f, o1 = 0.0f;
        for (int b = 0; b < 8; b++) {
            onode **k = n->kids + b * 11;
            v0 = rv3_delay(k[10], v0);
            v0 = rv3_allnest(k[0], a * v0 + input0);
            v0 = rv3_allnest(k[1], v0);
            v0 = rv3_allnest(k[2], v0);
            v0 = rv3_allnest(k[3], v0);
            v0 = rv3_lowpole(k[8], v0);
            o0 = v0;
            v0 = rv3_allnest(k[4], a * v0 + input1);
            v0 = rv3_allnest(k[5], v0);
            v0 = rv3_allnest(k[6], v0);
            v0 = rv3_allnest(k[7], v0);
            v0 = rv3_lowpole(k[9], v0);
            o1 = v0;
        }
        out[t] = o0;
        out[MAXB + t] = o1;
    }
    n->rv3_feedback = v0;
}

/* The prelude's fdn example (prelude.rs:1334): split::<U16>() >> fdn::<U16, _>(stacki(|i| delay(t_i) >> fir((w0, w1, w2)))) >> join::<U16>(), one BLOCK
 * per call, MONOMORPHISED on plain arrays: Split, Feedback::tick (feedback.rs:130-134) with FrameHadamard (:35-57), Delay, Fir<U3> (fir.rs:57-70),
 * Join::process (audionode.rs:649-659: every term scaled by 1/16, then added) -- under the MXCSR switch of Feedback::new.  `st` = the caller's state:
 * [16] ring pointers / lengths / positions, v[16][3], value[16].  in / out = [64]. */
void o_fdn16_block(float **ring, const size_t *len, size_t *pos, float (*v)[3], float *value, const float *w, int size, const float *in, float *out) {
#if defined(__x86_64__) || defined(__i386__)
    const unsigned int csr = _mm_getcsr();
    _mm_setcsr(0x9fc0);
#endif
    const float scale = (float)(1.0 / sqrt(16.0)), z = 1.0f / 16.0f;
    for (int t = 0; t < size; t++) {
        float o[16], h[16];
        for (int i = 0; i < 16; i++) {
            const float x = in[t] + value[i];
            ring[i][pos[i]] = x;
            pos[i] += 1;
            if (pos[i] >= len[i]) pos[i] = 0;
            v[i][0] = v[i][1]; v[i][1] = v[i][2]; v[i][2] = ring[i][pos[i]];
            float acc = 0.0f;
            acc += w[0] * v[i][0];
            acc += w[1] * v[i][1];
            acc += w[2] * v[i][2];
            o[i] = acc;
            h[i] = acc;
        }
        for (int hh = 1; hh < 16; hh *= 2)
            for (int i = 0; i < 16; i += hh * 2)
                for (int j = i; j < i + hh; j++) {
                    const float x = h[j], y = h[j + hh];
                    h[j] = x + y;
                    h[j + hh] = x - y;
                }
        for (int i = 0; i < 16; i++) value[i] = h[i] * scale;
        float y = o[0] * z;
        for (int i = 1; i < 16; i++) y += o[i] * z;
        out[t] = y;
    }
#if defined(__x86_64__) || defined(__i386__)
    _mm_setcsr(csr);
#endif
}
