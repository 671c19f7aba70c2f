/*
 * oracle/o_bank.c -- TEST INFRASTRUCTURE ONLY.  Renders V independent voices of a BASELINE config with the
 * generic oracle nodes (fundsp_oracle.c), optionally over several host threads.  It plays two roles:
 *   (1) bank-scale parity checker for the HIP engine (tests/), and
 *   (2) the "port" cpu_baseline leg of bench.py (the reference itself cannot be built here: no Rust).
 * Graph shapes follow BASELINE.json configs 2/3 and SURVEY.md section 8(d); reference constructors:
 * sine_hz prelude.rs:349, lowpass_hz prelude.rs:2111, noise prelude.rs (Noise::new noise.rs:179),
 * biquad prelude.rs (Biquad::with_coefs biquad.rs:151), operators combinator.rs:344-475.
 */
#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET: pinned timing threads (o_bank_pin_threads) */
#include "fundsp_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>

/* cpu_baseline honesty (bench.py): with pinning on, worker t of a bank job runs on the t-th CPU of the PROCESS'S affinity mask
 * (wrapping around when there are more workers than CPUs), so "N threads" means N distinct CPUs whenever the mask has them. */
static int g_pin_threads = 0;
void o_bank_pin_threads(int on) { g_pin_threads = on; }
int o_bank_allowed_cpus(void) {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) != 0) return -1;
    return CPU_COUNT(&set);
}
void o_bank_pin_self(int t) {
    if (!g_pin_threads) return;
    cpu_set_t set, one;
    if (sched_getaffinity(0, sizeof set, &set) != 0) return;
    int n = CPU_COUNT(&set);
    if (n <= 0) return;
    int want = t % n, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &set)) continue;
        if (seen++ == want) {
            CPU_ZERO(&one);
            CPU_SET(c, &one);
            pthread_setaffinity_np(pthread_self(), sizeof one, &one);
            return;
        }
    }
}

static onode *build_voice(const o_bank_job *job, size_t v) {
    onode *g = NULL;
    if (job->config == 3) {
        float f = job->p0[v], m = job->p1[v];
        /* sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q); Rust precedence: ((((c>>s)*f)*m)+f) >> s >> svf */
        onode *mod = o_pipe(o_constant(1, &f), o_sine());
        onode *e = o_unop(O_ADD_SCALAR, o_unop(O_MUL_SCALAR, o_unop(O_MUL_SCALAR, mod, f), m), f);
        g = o_pipe(o_pipe(e, o_sine()), o_fixed_svf(O_SVF_LOWPASS, job->p2[v], job->p3[v], 1.0f));
        o_set_sample_rate(g, job->sample_rate);
        o_set_seed(g, job->seed[v]);
    } else if (job->config == 2) {
        float c[5];
        o_biquad_coefs(O_BQ_LOWPASS, (float)job->sample_rate, job->p0[v], job->p1[v], 1.0f, c);
        onode *nz = o_noise();
        g = o_pipe(nz, o_biquad(c[0], c[1], c[2], c[3], c[4]));
        o_set_sample_rate(g, job->sample_rate);
        o_noise_set_seed(nz, job->seed[v]);
    }
    return g;
}

typedef struct {
    const o_bank_job *job;
    float *out;
    size_t v0, v1;
    int t;
} slice;

onode *o_bank_build_voice(const o_bank_job *job, size_t v) { return build_voice(job, v); }

static void *run_slice(void *arg) {
    slice *s = (slice *)arg;
    const o_bank_job *job = s->job;
    o_bank_pin_self(s->t);
    size_t T = job->frames, V = job->voices;
    float blk[64];
    for (size_t v = s->v0; v < s->v1; v++) {
        onode *g = build_voice(job, v);
        for (size_t i = 0; i < T; i += 64) {
            int n = (int)(T - i < 64 ? T - i : 64);
            if (job->process_mode) {
                o_process(g, n, NULL, blk);
            } else {
                for (int j = 0; j < n; j++) o_tick(g, NULL, &blk[j]);
            }
            if (job->out_layout == 0 && s->out)
                for (int j = 0; j < n; j++) s->out[v * T + i + (size_t)j] = blk[j];
            else if (job->out_layout == 1 && s->out)
                for (int j = 0; j < n; j++) s->out[(i + (size_t)j) * V + v] = blk[j];
        }
        o_free(g);
    }
    return NULL;
}

double o_bank_render(const o_bank_job *job, float *out) {
    int nt = job->threads > 0 ? job->threads : 1;
    if ((size_t)nt > job->voices) nt = (int)job->voices;
    if (nt < 1) nt = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nt);
    slice *sl = (slice *)malloc(sizeof(slice) * (size_t)nt);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nt; t++) {
        sl[t].job = job;
        sl[t].out = out;
        sl[t].v0 = job->voices * (size_t)t / (size_t)nt;
        sl[t].v1 = job->voices * (size_t)(t + 1) / (size_t)nt;
        sl[t].t = t;
        pthread_create(&th[t], NULL, run_slice, &sl[t]);
    }
    for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(sl);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
