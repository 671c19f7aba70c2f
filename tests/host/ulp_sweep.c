/* tests/host/ulp_sweep.c -- EXHAUSTIVE accuracy bounds of the oracle's restated libm / wide functions (test infrastructure).
 *
 * VERDICT r02 (Weak 1, Next 1b): the bounds that would catch a mistyped constant were sampled on narrow ranges.  The
 * device build of every one of these functions is bit-identical to the oracle's on ALL 2^32 f32 inputs
 * (tests/host/check_math_device.hip, profiles/r02_math_device_exhaustive.txt), so ONE exhaustive host sweep of the
 * oracle against double-precision libm bounds both sides at once.  For every function and every input bit pattern:
 *     err = |f(x) - ref(x)| / ulp_f32(ref(x))        ref = the host C library's f64 function of (double)x
 * with the maximum kept per binade (biased exponent of x, both signs), printed as a table, and the documented bounds of
 * the algorithms (musl / FreeBSD msun: sinf cosf < 0.51 ulp, tanf < 0.81, expf < 1, expm1f < 1, tanhf < 2.5,
 * atanf < 1; powf < 1; vectorclass sin within 1 ulp-of-1 absolute while its Cody-Waite reduction holds) ASSERTED.
 * NaN in <-> NaN out, and signed-zero / infinity results are compared exactly.
 *
 * build: gcc -O2 -ffp-contract=off -fno-fast-math -pthread -I oracle -o tests/host/_build/ulp_sweep tests/host/ulp_sweep.c -lm
 * run:   ulp_sweep [threads] [stride]      stride 1 = all 2^32 inputs (~10 min on 8 cores); the CPU suite runs a strided pass
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "o_math.h"

typedef struct {
    const char *name;
    float (*f)(float);
    double (*ref)(double);
    double bound;      /* asserted max ulp over the function's whole domain (0 = use abs_bound) */
    double abs_bound;  /* asserted max ABSOLUTE error where |x| <= abs_limit (vectorclass sin: its own accuracy statement) */
    double abs_limit;
    const char *note;
} Fn;

static float f_pow_diag(float x) { return o_powf(x, x); }
static double r_pow_diag(double x) { return pow(x, x); }
static float f_pow_2x(float x) { return o_powf(2.0f, x); }
static double r_pow_2x(double x) { return pow(2.0, x); }
static float f_pow_10x(float x) { return o_powf(10.0f, x); }
static double r_pow_10x(double x) { return pow(10.0, x); }
static float f_pow_x15(float x) { return o_powf(x, 1.5f); }
static double r_pow_x15(double x) { return pow(x, 1.5); }
static float f_pow_xm2(float x) { return o_powf(x, -2.0f); }
static double r_pow_xm2(double x) { return pow(x, -2.0); }

static const Fn FNS[] = {
    {"sinf", o_sinf, sin, 0.51, 0, 0, "musl sinf.c + __sindf/__cosdf + __rem_pio2f + __rem_pio2_large"},
    {"cosf", o_cosf, cos, 0.51, 0, 0, "musl cosf.c"},
    {"tanf", o_tanf, tan, 0.81, 0, 0, "musl tanf.c + __tandf"},
    {"expf", o_expf, exp, 1.0, 0, 0, "libm 0.2 expf.rs = msun e_expf.c (\"error is less than 1 ulp\")"},
    {"expm1f", o_expm1f, expm1, 1.0, 0, 0, "msun s_expm1f.c"},
    {"tanhf", o_tanhf, tanh, 2.5, 0, 0, "musl tanhf.c on expm1f"},
    {"atanf", o_atanf, atan, 1.0, 0, 0, "msun s_atanf.c"},
    {"powf(x,x)", f_pow_diag, r_pow_diag, 1.0, 0, 0, "msun e_powf.c, diagonal"},
    {"powf(2,x)", f_pow_2x, r_pow_2x, 1.0, 0, 0, "msun e_powf.c"},
    {"powf(10,x)", f_pow_10x, r_pow_10x, 1.0, 0, 0, "msun e_powf.c (db_amp)"},
    {"powf(x,1.5)", f_pow_x15, r_pow_x15, 1.0, 0, 0, "msun e_powf.c"},
    {"powf(x,-2)", f_pow_xm2, r_pow_xm2, 1.0, 0, 0, "msun e_powf.c"},
    {"wide_sin", o_wide_sinf, sin, 0, 1.2e-7, 8192.0, "vectorclass sincos_f (wide f32x8::sin), absolute error for |x| <= 8192"},
    {"wide_atan", o_wide_atanf, atan, 4.0, 0, 0, "vectorclass atan_f (wide f32x8::atan); no published bound: 4 ulp asserted, the exhaustive maximum is what this file records"},
};
#define NFN ((int)(sizeof(FNS) / sizeof(FNS[0])))

typedef struct {
    double max_ulp[256], max_abs[256];
    uint32_t arg_ulp[256];
    uint64_t mism;  /* NaN / inf / signed-zero class mismatches */
    uint32_t mism_arg;
} Stat;

typedef struct {
    int tid, nthreads;
    uint32_t stride;
    Stat st[NFN];
} Job;

static double ulp_of(double r) { /* spacing of f32 at |r| (2^-149 below the normal range) */
    double a = fabs(r);
    if (a < 1.1754943508222875e-38) return 1.401298464324817e-45;
    int e;
    frexp(a, &e); /* a = m * 2^e, m in [0.5, 1) */
    return ldexp(1.0, e - 24);
}

static void *worker(void *arg) {
    Job *j = (Job *)arg;
    /* thread t takes the bit patterns whose low bits select it: every thread sees every binade */
    for (uint64_t u64 = (uint64_t)j->tid * j->stride; u64 < (1ull << 32); u64 += (uint64_t)j->nthreads * j->stride) {
        const uint32_t u = (uint32_t)u64;
        const float x = o_u2f(u);
        const int be = (u >> 23) & 0xff;
        for (int k = 0; k < NFN; k++) {
            const float got = FNS[k].f(x);
            const double ref = FNS[k].ref((double)x);
            Stat *s = &j->st[k];
            const float reff = (float)ref;
            if (got != got || ref != ref) { /* NaN on either side: both must be NaN */
                if (!(got != got && ref != ref)) { s->mism++; s->mism_arg = u; }
                continue;
            }
            if (isinf(got) || isinf(reff) || reff == 0.0f || got == 0.0f) {
                /* overflow / underflow / exact zeros: the f32-rounded reference must agree within one step of the
                 * boundary -- measured below as ulps where finite, as a class mismatch where one side is inf */
                if (isinf(got) != isinf(reff)) {
                    /* got finite, ref overflowed (or vice versa): allowed only in the last binade before overflow */
                    const double lim = 3.4028234663852886e38;
                    if (!(fabs(ref) > lim * (1.0 - 1e-7) || fabs((double)got) > lim * (1.0 - 1e-7))) { s->mism++; s->mism_arg = u; }
                    continue;
                }
                if (isinf(got)) {
                    if ((got > 0) != (reff > 0)) { s->mism++; s->mism_arg = u; }
                    continue;
                }
            }
            const double err = fabs((double)got - ref);
            const double e_ulp = err / ulp_of(ref);
            if (e_ulp > s->max_ulp[be]) { s->max_ulp[be] = e_ulp; s->arg_ulp[be] = u; }
            if (err > s->max_abs[be]) s->max_abs[be] = err;
        }
    }
    return NULL;
}

int main(int argc, char **argv) {
    const int nthreads = argc > 1 ? atoi(argv[1]) : 8;
    const uint32_t stride = argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 0) : 1;
    pthread_t th[256];
    Job *jobs = (Job *)calloc((size_t)nthreads, sizeof(Job));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].tid = t; jobs[t].nthreads = nthreads; jobs[t].stride = stride;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    static Stat tot[NFN];
    for (int k = 0; k < NFN; k++)
        for (int t = 0; t < nthreads; t++) {
            for (int b = 0; b < 256; b++) {
                if (jobs[t].st[k].max_ulp[b] > tot[k].max_ulp[b]) { tot[k].max_ulp[b] = jobs[t].st[k].max_ulp[b]; tot[k].arg_ulp[b] = jobs[t].st[k].arg_ulp[b]; }
                if (jobs[t].st[k].max_abs[b] > tot[k].max_abs[b]) tot[k].max_abs[b] = jobs[t].st[k].max_abs[b];
            }
            if (jobs[t].st[k].mism) { tot[k].mism += jobs[t].st[k].mism; tot[k].mism_arg = jobs[t].st[k].mism_arg; }
        }
    printf("# ulp_sweep: oracle/o_math.h against the host C library's f64 functions; %s f32 bit patterns (stride %u), %d threads\n",
           stride == 1 ? "ALL 2^32" : "a strided sample of the", stride, nthreads);
    printf("# err = |f(x) - ref(x)| / ulp_f32(ref(x)); one row per binade of x (biased exponent, both signs)\n");
    printf("%-4s", "exp");
    for (int k = 0; k < NFN; k++) printf(" %11s", FNS[k].name);
    printf("\n");
    for (int b = 0; b < 256; b++) {
        printf("%-4d", b);
        for (int k = 0; k < NFN; k++) printf(" %11.4f", tot[k].max_ulp[b]);
        printf("\n");
    }
    int bad = 0;
    printf("# summary\n");
    for (int k = 0; k < NFN; k++) {
        double m = 0, mabs = 0;
        uint32_t arg = 0;
        for (int b = 0; b < 256; b++) {
            if (tot[k].max_ulp[b] > m) { m = tot[k].max_ulp[b]; arg = tot[k].arg_ulp[b]; }
            /* absolute error inside the function's stated range: binades with |x| <= abs_limit */
            if (FNS[k].abs_limit > 0 && ldexp(1.0, b - 127 + 1) <= FNS[k].abs_limit * 1.0000001 && tot[k].max_abs[b] > mabs) mabs = tot[k].max_abs[b];
        }
        int ok = tot[k].mism == 0;
        if (FNS[k].bound > 0) ok = ok && m < FNS[k].bound;
        if (FNS[k].abs_bound > 0) ok = ok && mabs < FNS[k].abs_bound;
        printf("%-12s max %.4f ulp at x = %.9g (0x%08x)", FNS[k].name, m, (double)o_u2f(arg), arg);
        if (FNS[k].bound > 0) printf("; bound %.2f ulp", FNS[k].bound);
        if (FNS[k].abs_bound > 0) printf("; max |error| %.3g for |x| <= %g (bound %.3g)", mabs, FNS[k].abs_limit, FNS[k].abs_bound);
        printf("; NaN/inf class mismatches %llu", (unsigned long long)tot[k].mism);
        if (tot[k].mism) printf(" (e.g. 0x%08x)", tot[k].mism_arg);
        printf(" -- %s: %s\n", FNS[k].note, ok ? "OK" : "FAIL");
        bad += !ok;
    }
    printf(bad ? "FAILED: %d function(s) outside their bounds\n" : "all %d functions inside their bounds\n", bad ? bad : NFN);
    return bad ? 1 : 0;
}
