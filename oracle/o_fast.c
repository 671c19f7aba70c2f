/*
 * oracle/o_fast.c -- TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg; never linked into the product).
 *
 * BASELINE config 3, `sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)`, rendered the way the compiled Rust
 * reference renders it: FunDSP graphs are statically typed, so rustc monomorphises Pipe<Pipe<Unop<..>, Sine>, FixedSvf>::
 * process into one straight function per block -- no node tree, no dispatch -- and Sine::process evaluates its sine as ONE
 * `wide::f32x8` operation per 8 frames (AVX on x86-64).  o_bank.c walks a heap tree with a switch per node per block
 * and loops over the 8 "SIMD" lanes in scalar code (VERDICT r02 Weak 7: "likely several times slower than the Rust
 * path it stands for").  This file is that monomorphised form:
 *   Constant::process            audionode.rs:491-497   splat per SIMD item
 *   Sine::process                oscillator.rs:74-86    8 serial phase steps, then (F32x::new(element) * TAU).sin()
 *   Unop::process                audionode.rs:1292-1303 one vector op per item (same arithmetic per lane as the scalar form)
 *   FixedSvf (default process)   audionode.rs:85-105 -> tick svf.rs:995-1006, per sample
 *   remainders                   process_remainder audionode.rs:110-126 -> tick (libm sinf, wrapped phase)
 * The 8-lane sine is `wide`'s algorithm (o_math.h o_wide_sinf, lane for lane) on GCC vector types: with -march=native it
 * compiles to the host's 256-bit unit, exactly the operations of the scalar restatement -> bit-identical to o_bank_render
 * (asserted in tests/test_oracle_fast.py), only faster.
 */
#include "fundsp_oracle.h"
#include "o_math.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef float v8f __attribute__((vector_size(32)));
typedef int32_t v8i __attribute__((vector_size(32)));
typedef uint32_t v8u __attribute__((vector_size(32)));

#if defined(__AVX2__)
#include <immintrin.h>
const char *o_fast_simd_flavour(void) { return "AVX2 (256-bit, GCC vector types + vroundps / vcvtps2dq)"; }
static inline v8f v8_rint(v8f x) { return (v8f)_mm256_round_ps((__m256)x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
/* wide f32x8::round_int, x86 form: cvtps2dq, NaN lanes masked to 0, lanes >= 2^31 flipped to i32::MAX */
static inline v8i v8_round_int_sat(v8f y) {
    v8i q = (v8i)_mm256_cvtps_epi32((__m256)y);
    const v8f big = {2147483648.0f, 2147483648.0f, 2147483648.0f, 2147483648.0f, 2147483648.0f, 2147483648.0f, 2147483648.0f, 2147483648.0f};
    v8i is_big = (v8i)(y >= big), is_nan = (v8i)(y != y);
    const v8i imax = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
    q = (q & ~is_big) | (imax & is_big);
    return q & ~is_nan;
}
#else
const char *o_fast_simd_flavour(void) { return "portable GCC vector types (no AVX2 on the build host)"; }
static inline v8f v8_rint(v8f x) {
    v8f r;
    for (int i = 0; i < 8; i++) r[i] = o_round_half_even(x[i]);
    return r;
}
static inline v8i v8_round_int_sat(v8f y) {
    v8i q;
    for (int i = 0; i < 8; i++) q[i] = o_round_int_sat(y[i]);
    return q;
}
#endif

static inline v8f v8_splat(float x) { return (v8f){x, x, x, x, x, x, x, x}; }
static inline v8i v8_splati(int32_t x) { return (v8i){x, x, x, x, x, x, x, x}; }

/* o_wide_sinf (o_math.h), eight lanes at once: the same operations in the same order, unfused */
static inline v8f wide_sin8(v8f self) {
    const v8f DP1F = v8_splat(0.78515625f * 2.0f), DP2F = v8_splat(2.4187564849853515625E-4f * 2.0f),
              DP3F = v8_splat(3.77489497744594108E-8f * 2.0f);
    const v8f P0s = v8_splat(-1.6666654611E-1f), P1s = v8_splat(8.3321608736E-3f), P2s = v8_splat(-1.9515295891E-4f);
    const v8f P0c = v8_splat(4.166664568298827E-2f), P1c = v8_splat(-1.388731625493765E-3f), P2c = v8_splat(2.443315711809948E-5f);
    const v8f TWO_OVER_PI = v8_splat(2.0f / 3.14159274101257324f);
    const v8u ABS = (v8u)v8_splati(0x7fffffff);
    v8f xa = (v8f)((v8u)self & ABS);
    v8f y = v8_rint(xa * TWO_OVER_PI);
    v8i q = v8_round_int_sat(y);
    v8f x = xa - y * DP1F;
    x = x - y * DP2F;
    x = x - y * DP3F;
    v8f x2 = x * x;
    v8f x4 = x2 * x2;
    v8f s = (x4 * P2s + (x2 * P1s + P0s)) * (x * x2) + x;
    v8f c = (x4 * P2c + (x2 * P1c + P0c)) * (x2 * x2) + (v8_splat(1.0f) - v8_splat(0.5f) * x2);
    v8i swap = (q & v8_splati(1)) != v8_splati(0);
    const v8f INF = v8_splat(__builtin_inff());
    v8i overflow = (q > v8_splati(0x2000000)) & (v8i)(xa < INF);
    s = (v8f)(((v8i)s & ~overflow));                                         /* overflow ? 0.0 : s */
    c = (v8f)(((v8i)c & ~overflow) | ((v8i)v8_splat(1.0f) & overflow));      /* overflow ? 1.0 : c */
    v8i sin1 = ((v8i)c & swap) | ((v8i)s & ~swap);
    v8u sign_sin = ((v8u)q << 30) ^ (v8u)self;
    return (v8f)((v8u)sin1 ^ (sign_sin & (v8u)v8_splati((int32_t)0x80000000u)));
}

/* one block (size <= 64) of one voice; out[64] */
static inline void fm_svf_block(o_fm_svf_regs *r, int size, float *out) {
    const float TAU = 6.28318548202514648f;
    const int full = size & ~7;
    v8f car[8]; /* carrier samples of the full items, then filtered per sample */
    float mp = r->mod_phase, cp = r->car_phase;
    const v8f vf = v8_splat(r->mul_f), vm = v8_splat(r->mul_m), va = v8_splat(r->add_f), vtau = v8_splat(TAU);
    for (int i = 0; i < full; i += 8) {
        v8f el;
        for (int j = 0; j < 8; j++) { /* modulator Sine::process: the Constant's splat is its input */
            el[j] = mp;
            mp += r->f_const * r->mod_sd;
        }
        v8f mod = wide_sin8(el * vtau);
        v8f fr = mod * vf * vm + va; /* Unop chain: ((x * f) * m) + f, each a separate rounding (-ffp-contract=off) */
        for (int j = 0; j < 8; j++) { /* carrier Sine::process */
            el[j] = cp;
            cp += fr[j] * r->car_sd;
        }
        car[i >> 3] = wide_sin8(el * vtau);
    }
    mp = mp - floorf(mp); /* oscillator.rs:85: one wrap after the SIMD items */
    cp = cp - floorf(cp);
    float tail[8];
    for (int i = full; i < size; i++) { /* process_remainder -> tick: wrapped phase, libm sinf */
        float p = mp;
        mp += r->f_const * r->mod_sd;
        mp -= floorf(mp);
        float fr = o_sinf(p * TAU) * r->mul_f * r->mul_m + r->add_f;
        p = cp;
        cp += fr * r->car_sd;
        cp -= floorf(cp);
        tail[i - full] = o_sinf(p * TAU);
    }
    r->mod_phase = mp;
    r->car_phase = cp;
    float ic1 = r->ic1eq, ic2 = r->ic2eq;
    const float a1 = r->a1, a2 = r->a2, a3 = r->a3, m0 = r->m0, m1 = r->m1, m2 = r->m2;
    for (int i = 0; i < size; i++) { /* FixedSvf::tick svf.rs:995-1006 */
        const float v0 = i < full ? car[i >> 3][i & 7] : tail[i - full];
        const float v3 = v0 - ic2;
        const float v1 = a1 * ic1 + a2 * v3;
        const float v2 = ic2 + a2 * ic1 + a3 * v3;
        ic1 = 2.0f * v1 - ic1;
        ic2 = 2.0f * v2 - ic2;
        out[i] = m0 * v0 + m1 * v1 + m2 * v2;
    }
    r->ic1eq = ic1;
    r->ic2eq = ic2;
}

typedef struct {
    const o_bank_job *job;
    float *out;
    size_t v0, v1;
    int t;
} fslice;

static void *run_fast_slice(void *arg) {
    fslice *s = (fslice *)arg;
    const o_bank_job *job = s->job;
    o_bank_pin_self(s->t);
    const size_t T = job->frames, V = job->voices;
    float blk[64];
    for (size_t v = s->v0; v < s->v1; v++) {
        /* construction, set_sample_rate and set_seed through the generic oracle nodes (once per voice, not timed work of
         * the block path): the registers a monomorphised process() would hold */
        onode *g = o_bank_build_voice(job, v);
        o_fm_svf_regs r;
        const int ok = g && o_fm_svf_state(g, &r) == 0;
        o_free(g);
        if (!ok) abort();
        for (size_t i = 0; i < T; i += 64) {
            const int n = (int)(T - i < 64 ? T - i : 64);
            fm_svf_block(&r, n, blk);
            if (job->out_layout == 0 && s->out)
                memcpy(&s->out[v * T + i], blk, (size_t)n * sizeof(float));
            else if (job->out_layout == 1 && s->out)
                for (int j = 0; j < n; j++) s->out[(i + (size_t)j) * V + v] = blk[j];
        }
    }
    return NULL;
}

double o_bank_render_fast(const o_bank_job *job, float *out) {
    if (job->config != 3 || !job->process_mode) return -1.0;
    int nt = job->threads > 0 ? job->threads : 1;
    if ((size_t)nt > job->voices) nt = (int)job->voices;
    if (nt < 1) nt = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
    fslice *sl = (fslice *)malloc(sizeof(fslice) * (size_t)nt);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nt; t++) {
        sl[t].job = job;
        sl[t].out = out;
        sl[t].v0 = job->voices * (size_t)t / (size_t)nt;
        sl[t].v1 = job->voices * (size_t)(t + 1) / (size_t)nt;
        sl[t].t = t;
        pthread_create(&th[t], NULL, run_fast_slice, &sl[t]);
    }
    for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(sl);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- cpu_baseline legs of BASELINE configs 4 and 5 ------------------------------------------------------------------------------
 * Threaded like o_bank_render_fast: every thread renders its slice of the voices one after the other (MANY per thread, each for the whole
 * duration in 64-sample blocks), through the monomorphised block functions of fundsp_oracle.c (fast = 1) or the generic tree walk (fast = 0). */
static const owavetable *saw_table(void) { /* Wavetable::new for saw(): built once (o_wavetable.c), shared like the reference's Arc<Wavetable> */
    static const owavetable *t;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    if (!t) {
        float *pitches = (float *)malloc(64 * sizeof(float)), *data = (float *)malloc((size_t)64 * 8192 * sizeof(float));
        int *lengths = (int *)malloc(64 * sizeof(int));
        const int n = o_make_wavetable(0, 64, pitches, lengths, (size_t)64 * 8192, data);
        if (n > 0) t = o_wavetable_create(n, pitches, lengths, data);
        free(pitches); free(lengths); free(data);
    }
    pthread_mutex_unlock(&mu);
    return t;
}

/* ((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * ENV >> pan(p), set_sample_rate, set_seed(v): what tests/ and bench.py build through the
 * Python notation, here through the C constructors; *var = the Var node (Var shape) or NULL */
static onode *c4_build(const o_c4_job *job, size_t v, onode **var) {
    const owavetable *t = saw_table();
    if (!t) return NULL;
    float f = job->p0[v], fc = job->p1[v], q = job->p2[v];
    onode *osc = o_pipe(o_constant(1, &f), o_wavesynth(t, 1));
    onode *sm = o_pipe(o_stack(o_stack(osc, o_constant(1, &fc)), o_constant(1, &q)), o_moog(3, 1000.0f, 0.1f));
    onode *env = o_adsr_live(job->adsr[0], job->adsr[1], job->adsr[2], job->adsr[3]);
    *var = NULL;
    if (job->gate_var) {
        *var = o_var(0.0f);
        env = o_pipe(*var, env);
    }
    onode *g = o_pipe(o_binop(O_MUL, sm, env), o_panner(1, job->p3[v]));
    o_set_sample_rate(g, job->sample_rate);
    o_set_seed(g, job->seed[v]);
    return g;
}

typedef struct {
    const o_c4_job *job;
    float *out;
    size_t v0, v1;
    int t;
} c4slice;

static void *run_c4_slice(void *arg) {
    c4slice *s = (c4slice *)arg;
    const o_c4_job *job = s->job;
    o_bank_pin_self(s->t);
    const size_t T = job->frames;
    float in[64], blk[2 * 64];
    memset(blk, 0, sizeof blk);
    for (size_t v = s->v0; v < s->v1; v++) {
        onode *var = NULL, *g = c4_build(job, v, &var);
        o_c4_voice cv;
        if (!g || (job->fast && o_c4_open(g, &cv) != 0)) abort();
        size_t i = 0;
        /* the Var shape walks its plan: one launch per entry, a new block starts with every entry (as separate process() calls would);
         * the stream shape is one entry of T frames */
        const int np = job->gate_var ? job->n_plan : 1;
        for (int k = 0; k < np && i < T; k++) {
            size_t n_entry = T;
            if (job->gate_var) {
                o_var_set(var, job->plan[2 * k]);
                n_entry = (size_t)job->plan[2 * k + 1];
            }
            for (size_t j = 0; j < n_entry && i < T; j += 64) {
                size_t left = n_entry - j < T - i ? n_entry - j : T - i;
                const int n = (int)(left < 64 ? left : 64);
                if (!job->gate_var) memcpy(in, job->gate + i, (size_t)n * sizeof(float));
                if (job->fast) o_c4_block(&cv, n, in, blk);
                else o_process(g, n, in, blk);
                if (s->out) {
                    memcpy(&s->out[(v * 2 + 0) * T + i], blk, (size_t)n * sizeof(float));
                    memcpy(&s->out[(v * 2 + 1) * T + i], blk + 64, (size_t)n * sizeof(float));
                }
                i += (size_t)n;
            }
        }
        o_free(g);
    }
    return NULL;
}

static double run_threads(int nt, size_t units, void *(*fn)(void *), void *slices, size_t slice_size, void (*fill)(void *, size_t, size_t, int)) {
    if ((size_t)nt > units) nt = (int)units;
    if (nt < 1) nt = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
    struct timespec t0, t1;
    for (int t = 0; t < nt; t++) fill((char *)slices + (size_t)t * slice_size, units * (size_t)t / (size_t)nt, units * (size_t)(t + 1) / (size_t)nt, t);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nt; t++) pthread_create(&th[t], NULL, fn, (char *)slices + (size_t)t * slice_size);
    for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

static const o_c4_job *g_c4_job;
static float *g_c4_out;
static void c4_fill(void *p, size_t v0, size_t v1, int t) {
    c4slice *s = (c4slice *)p;
    s->job = g_c4_job; s->out = g_c4_out; s->v0 = v0; s->v1 = v1; s->t = t;
}
double o_c4_bank_render(const o_c4_job *job, float *out) {
    if (!saw_table() || (job->gate_var ? (job->n_plan < 1 || !job->plan) : !job->gate)) return -1.0;
    int nt = job->threads > 0 ? job->threads : 1;
    c4slice *sl = (c4slice *)calloc((size_t)nt, sizeof(c4slice));
    g_c4_job = job; g_c4_out = out;
    const double s = run_threads(nt, job->voices, run_c4_slice, sl, sizeof(c4slice), c4_fill);
    free(sl);
    return s;
}

typedef struct {
    int fast, t;
    double sr, room, time, damping;
    size_t i0, i1, frames;
    const float *x;
    float *out;
} rvslice;
static rvslice g_rv;
static void rv_fill(void *p, size_t i0, size_t i1, int t) {
    rvslice *s = (rvslice *)p;
    *s = g_rv;
    s->i0 = i0; s->i1 = i1; s->t = t;
}
static void *run_rv_slice(void *arg) {
    rvslice *s = (rvslice *)arg;
    o_bank_pin_self(s->t);
    const size_t T = s->frames;
    float in[2 * 64], blk[2 * 64];
    for (size_t k = s->i0; k < s->i1; k++) {
        onode *g = o_reverb_stereo(s->room, s->time, s->damping);
        o_set_sample_rate(g, s->sr);
        for (size_t i = 0; i < T; i += 64) {
            const int n = (int)(T - i < 64 ? T - i : 64);
            memcpy(in, s->x + i, (size_t)n * sizeof(float));
            memcpy(in + 64, s->x + T + i, (size_t)n * sizeof(float));
            if (s->fast) o_reverb_stereo_block(g, n, in, blk);
            else o_process(g, n, in, blk);
            if (s->out) {
                memcpy(&s->out[(k * 2 + 0) * T + i], blk, (size_t)n * sizeof(float));
                memcpy(&s->out[(k * 2 + 1) * T + i], blk + 64, (size_t)n * sizeof(float));
            }
        }
        o_free(g);
    }
    return NULL;
}
double o_reverb_bank_render(int threads, int fast, double sample_rate, size_t instances, size_t frames, double room, double time, double damping,
                            const float *x, float *out) {
    int nt = threads > 0 ? threads : 1;
    rvslice *sl = (rvslice *)calloc((size_t)nt, sizeof(rvslice));
    memset(&g_rv, 0, sizeof g_rv);
    g_rv.fast = fast; g_rv.sr = sample_rate; g_rv.room = room; g_rv.time = time; g_rv.damping = damping; g_rv.frames = frames; g_rv.x = x; g_rv.out = out;
    const double s = run_threads(nt, instances, run_rv_slice, sl, sizeof(rvslice), rv_fill);
    free(sl);
    return s;
}

/* ---- CPU legs of round 6's lane-per-frame kinds (bench.py secondary entries): fast = 1 the monomorphised block forms (fundsp_oracle.c o_reverb3_block /
 * o_fdn16_block), fast = 0 the generic tree walk of the graph; one graph object per instance, the instances split over pinned threads.  which = 0: reverb3_stereo(p[0], p[1], lowpole_hz(p[2])) (reverb.rs:152-279), 2 in / 2 out;
 * which = 1: the prelude's fdn example (prelude.rs:1334) split >> fdn::<U16>(stacki(|i| delay(p[i]) >> fir((p[16], p[17], p[18])))) >> join, 1 in / 1 out.
 * x = [inputs][frames] shared by all instances; out = [instances][outputs][frames] or NULL. */
typedef struct {
    int which, fast, t, nin, nout;
    double sr;
    const double *p;
    size_t i0, i1, frames;
    const float *x;
    float *out;
} ggslice;
static ggslice g_gg;
static void gg_fill(void *q, size_t i0, size_t i1, int t) {
    ggslice *s = (ggslice *)q;
    *s = g_gg;
    s->i0 = i0; s->i1 = i1; s->t = t;
}
static onode *gg_make(int which, const double *p) {
    if (which == 0) {
        onode *fl[16];
        for (int i = 0; i < 16; i++) fl[i] = o_onepole(0, 1, (float)p[2]);
        return o_reverb3(p[0], p[1], fl);
    }
    onode *lines[16];
    const float w[3] = {(float)p[16], (float)p[17], (float)p[18]};
    for (int i = 0; i < 16; i++) lines[i] = o_pipe(o_delay(p[i]), o_fir(3, w));
    return o_pipe(o_pipe(o_split(1, 16), o_feedback(o_multi(O_MULTI_STACK, 16, lines, 0), NULL, 1)), o_join(1, 16));
}
static void *run_gg_slice(void *arg) {
    ggslice *s = (ggslice *)arg;
    o_bank_pin_self(s->t);
    const size_t T = s->frames;
    float in[2 * 64], blk[2 * 64];
    for (size_t k = s->i0; k < s->i1; k++) {
        if (s->fast && s->which == 1) {   /* the fdn example on plain arrays (o_fdn16_block) */
            float *ring[16], v[16][3], value[16];
            size_t len[16], pos[16];
            const float w[3] = {(float)s->p[16], (float)s->p[17], (float)s->p[18]};
            for (int i = 0; i < 16; i++) {
                len[i] = (size_t)round(s->p[i] * s->sr) + 1;   /* Delay::set_sample_rate delay.rs:105-112 */
                ring[i] = (float *)calloc(len[i], sizeof(float));
                pos[i] = 0;
                v[i][0] = v[i][1] = v[i][2] = 0.0f;
                value[i] = 0.0f;
            }
            for (size_t i = 0; i < T; i += 64) {
                const int n = (int)(T - i < 64 ? T - i : 64);
                o_fdn16_block(ring, len, pos, v, value, w, n, s->x + i, blk);
                if (s->out) memcpy(&s->out[k * T + i], blk, (size_t)n * sizeof(float));
            }
            for (int i = 0; i < 16; i++) free(ring[i]);
            continue;
        }
        onode *g = gg_make(s->which, s->p);
        o_set_sample_rate(g, s->sr);
        const int block3 = s->fast && s->which == 0 && o_reverb3_block_ok(g);
        for (size_t i = 0; i < T; i += 64) {
            const int n = (int)(T - i < 64 ? T - i : 64);
            for (int c = 0; c < s->nin; c++) memcpy(in + 64 * c, s->x + (size_t)c * T + i, (size_t)n * sizeof(float));
            if (block3) o_reverb3_block(g, n, in, blk);
            else o_process(g, n, in, blk);
            if (s->out)
                for (int c = 0; c < s->nout; c++) memcpy(&s->out[(k * (size_t)s->nout + (size_t)c) * T + i], blk + 64 * c, (size_t)n * sizeof(float));
        }
        o_free(g);
    }
    return NULL;
}
double o_graph_bank_render(int threads, int which, int fast, const double *p, double sample_rate, size_t instances, size_t frames, const float *x, float *out) {
    int nt = threads > 0 ? threads : 1;
    ggslice *sl = (ggslice *)calloc((size_t)nt, sizeof(ggslice));
    memset(&g_gg, 0, sizeof g_gg);
    g_gg.which = which; g_gg.fast = fast; g_gg.p = p; g_gg.sr = sample_rate; g_gg.frames = frames; g_gg.x = x; g_gg.out = out;
    g_gg.nin = which == 0 ? 2 : 1; g_gg.nout = which == 0 ? 2 : 1;
    const double s = run_threads(nt, instances, run_gg_slice, sl, sizeof(ggslice), gg_fill);
    free(sl);
    return s;
}

/* ---- BASELINE config 2 the way the reference runs it: BiquadBank<f32x8> (biquad_bank.rs:14-130) -- EIGHT voices per SIMD instruction ----
 * `(noise() | .. | noise()) >> biquad_bank()`: 8 Noise generators (Noise::process noise.rs:204-218: 8 frames per item, time-vectorised
 * integer hash) feed one bank whose tick (biquad_bank.rs:73-84) is the DF1 expression on f32x8 -- lane k of bank j is voice 8 j + k of
 * o_bank_render's config 2 (same coefficients, same seeds), the same IEEE operations per lane in the same order -> bit-identical to the scalar
 * restatement (tests/test_oracle_fast.py), one eighth of the arithmetic instructions.  Voices in groups of 8 (a ragged tail runs with
 * silent lanes). */
typedef struct {
    const o_bank_job *job;
    float *out;
    size_t b0, b1;
    int t;
} bbslice;
static const o_bank_job *g_bb_job;
static float *g_bb_out;
static void bb_fill(void *p, size_t b0, size_t b1, int t) {
    bbslice *s = (bbslice *)p;
    s->job = g_bb_job; s->out = g_bb_out; s->b0 = b0; s->b1 = b1; s->t = t;
}
static void *run_bb_slice(void *arg) {
    bbslice *s = (bbslice *)arg;
    const o_bank_job *job = s->job;
    o_bank_pin_self(s->t);
    const size_t T = job->frames, V = job->voices;
    for (size_t bank = s->b0; bank < s->b1; bank++) {
        v8f a1, a2, b0, b1, b2, x1 = v8_splat(0.0f), x2 = x1, y1 = x1, y2 = x1;
        uint32_t nstate[8];
        for (int k = 0; k < 8; k++) {
            const size_t v = bank * 8 + (size_t)k;
            float c[5] = {0, 0, 0, 0, 0};
            uint64_t h = 0;
            if (v < V) {
                o_biquad_coefs(O_BQ_LOWPASS, (float)job->sample_rate, job->p0[v], job->p1[v], 1.0f, c);   /* Setting::biquad(..).index(k) */
                h = job->seed[v];
            }
            a1[k] = c[0]; a2[k] = c[1]; b0[k] = c[2]; b1[k] = c[3]; b2[k] = c[4];
            nstate[k] = (uint32_t)(h ^ (h >> 32));                                                        /* Noise::reset noise.rs:192-195 */
        }
        float nz[8][64], blk[8][64];
        for (size_t i = 0; i < T; i += 64) {
            const int n = (int)(T - i < 64 ? T - i : 64), items8 = ((n + 7) >> 3) * 8;
            for (int k = 0; k < 8; k++) {                                                                 /* Noise::process noise.rs:204-218 */
                const uint32_t st = nstate[k];
                for (int j = 0; j < items8; j++) nz[k][j] = (float)(o_hash32x(st + (uint32_t)j + 1u) >> 8) * (2.0f / (float)((1 << 24) - 1)) - 1.0f;
                nstate[k] = st + (uint32_t)n;
            }
            for (int j = 0; j < n; j++) {                                                                 /* default process -> BiquadBank::tick :73-84 */
                const v8f x0 = {nz[0][j], nz[1][j], nz[2][j], nz[3][j], nz[4][j], nz[5][j], nz[6][j], nz[7][j]};   /* F::from_frame */
                const v8f y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
                x2 = x1; x1 = x0; y2 = y1; y1 = y0;
                for (int k = 0; k < 8; k++) blk[k][j] = y0[k];                                            /* to_frame */
            }
            if (s->out)
                for (int k = 0; k < 8; k++) {
                    const size_t v = bank * 8 + (size_t)k;
                    if (v >= V) break;
                    if (job->out_layout == 0) memcpy(&s->out[v * T + i], blk[k], (size_t)n * sizeof(float));
                    else if (job->out_layout == 1) for (int j = 0; j < n; j++) s->out[(i + (size_t)j) * V + v] = blk[k][j];
                }
        }
    }
    return NULL;
}
/* config 2 of an o_bank_job through 8-lane banks; returns seconds, < 0 if the job is not config 2 in process mode */
double o_biquad_bank8_render(const o_bank_job *job, float *out) {
    if (job->config != 2 || !job->process_mode) return -1.0;
    int nt = job->threads > 0 ? job->threads : 1;
    bbslice *sl = (bbslice *)calloc((size_t)nt, sizeof(bbslice));
    g_bb_job = job; g_bb_out = out;
    const double s = run_threads(nt, (job->voices + 7) / 8, run_bb_slice, sl, sizeof(bbslice), bb_fill);
    free(sl);
    return s;
}
