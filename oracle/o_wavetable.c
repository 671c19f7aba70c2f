/*
 * oracle/o_wavetable.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
 *
 * Restatement of the reference's wavetable construction: Wavetable::new (src/wavetable.rs:91-123), make_wave
 * (:44-79) and the built-in table closures saw_table :493, square_table :510, triangle_table :523, organ_table :546,
 * soft_saw_table :574, hammond_table :598.
 *
 * make_wave ends in microfft 0.6.0's inverse FFT (src/fft.rs:51-100).  The microfft source is not under
 * /root/reference (Cargo.toml dependency, un-vendored): PARITY UNPINNED at the bit level for the table samples.  Its
 * published algorithm is restated: an in-place f32 radix-2 decimation-in-time complex FFT (bit-reversal reorder, then
 * butterfly stages of span 2, 4, .. N with twiddles exp(-2*pi*i*k/span) taken from a table of correctly rounded f32
 * sine values), the inverse obtained by reversing elements 1..N of the input, running the forward transform and
 * dividing by N.  The same restatement is written independently in the product (fundsp_amd/csrc/fd_capi.hip,
 * build_default_table_set); tests/test_wavetable_build.py and tests/test_gpu_config4.py demand identical bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "fundsp_oracle.h"
#include "o_math.h"

typedef struct { float re, im; } c32;

/* Complex32 * Complex32 (num_complex Mul): (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re) */
static inline c32 c32_mul(c32 a, c32 b) {
    c32 r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}

/* forward complex FFT, radix-2 DIT, in place */
static void o_cfft(c32 *x, size_t n) {
    for (size_t i = 1, j = 0; i < n; i++) { /* bit-reversal reorder */
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            c32 t = x[i];
            x[i] = x[j];
            x[j] = t;
        }
    }
    for (size_t span = 2; span <= n; span <<= 1) {
        const size_t half = span >> 1;
        for (size_t k = 0; k < half; k++) {
            /* twiddle exp(-2 pi i k / span): f32 roundings of the exact cosine / sine (a sine table in microfft) */
            const double a = 6.283185307179586476925286766559 * (double)k / (double)span;
            const c32 w = {(float)cos(a), (float)-sin(a)};
            for (size_t i = k; i < n; i += span) {
                const c32 y = c32_mul(w, x[i + half]);
                const c32 u = x[i];
                x[i + half].re = u.re - y.re;
                x[i + half].im = u.im - y.im;
                x[i].re = u.re + y.re;
                x[i].im = u.im + y.im;
            }
        }
    }
}

/* microfft ifft_N: reverse x[1..N], forward transform, divide by N */
static void o_ifft(c32 *x, size_t n) {
    for (size_t i = 1, j = n - 1; i < j; i++, j--) {
        c32 t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
    o_cfft(x, n);
    const float fn = (float)n;
    for (size_t i = 0; i < n; i++) {
        x[i].re = x[i].re / fn;
        x[i].im = x[i].im / fn;
    }
}

static double wt_phase(int kind, uint32_t i) {
    switch (kind) {
    case 0: return (i & 1) == 1 ? 0.0 : 0.5;                                  /* saw :502 */
    case 1: case 6: return 0.0;                                                /* square :514, hammond :602 */
    case 2: return (i & 3) == 3 ? 0.5 : 0.0;                                  /* triangle :532 */
    default: return (i & 3) == 3 ? 0.5 : ((i & 1) == 1 ? 0.0 : 0.5);          /* organ :555-563, soft saw :583-591 */
    }
}

static double wt_amplitude(int kind, uint32_t i) {
    uint32_t z = 0, j = i;
    while ((j & 1u) == 0) { /* i.trailing_zeros(), i >= 1 */
        j >>= 1;
        z++;
    }
    switch (kind) {
    case 0: return 1.0 / (double)i;                                           /* :503 */
    case 1: return (i & 1) == 1 ? 1.0 / (double)i : 0.0;                      /* :515 */
    case 2: return (i & 1) == 1 ? 1.0 / (double)(uint32_t)(i * i) : 0.0;      /* :534-538, u32 product */
    case 4: return 1.0 / (double)(uint32_t)(i + j * j * j);                   /* organ :564-568 */
    case 5: return 1.0 / (double)(uint32_t)(i * i);                           /* soft saw :592 */
    default: {                                                                 /* hammond :603-619 */
        const double f = 1.0 / (double)(uint32_t)((z + 1) * (z + 1));
        if (i <= 3) return 1.0;
        return (j == 1 || j == 3) ? f : (j == 9 ? 0.2 * f : 0.0);
    }
    }
}

/* make_wave wavetable.rs:44-79.  Returns the table length; `out` must hold 8192 floats. */
static size_t o_make_wave(int kind, double pitch, float *out) {
    const double MAX_F = 22000.0, FADE_F = 20000.0;
    const size_t harmonics = (size_t)floor(MAX_F / pitch);
    const size_t target = 4 * harmonics;
    size_t length = 1; /* usize::next_power_of_two */
    while (length < target) length <<= 1;
    length = length < 32 ? 32 : (length > 8192 ? 8192 : length); /* clamp(32, 8192, ..) */
    c32 *a = (c32 *)calloc(length, sizeof(c32));
    for (size_t i = 1; i <= harmonics; i++) {
        const double f = pitch * (double)i;
        double w = wt_amplitude(kind, (uint32_t)i);
        double x = (f - MAX_F) / (FADE_F - MAX_F);            /* delerp math.rs:218 */
        x = fmin(fmax(x, 0.0), 1.0);                          /* clamp01 math.rs:136 */
        w = w * (((x * 6.0 - 15.0) * x + 10.0) * x * x * x);  /* smooth5 math.rs:418 */
        if (w > 0.0) { /* Complex32::from_polar(w as f32, (TAU * phase) as f32) = (r cos t, r sin t) */
            const float r = (float)w, theta = (float)(6.283185307179586476925286766559 * wt_phase(kind, (uint32_t)i));
            a[i].re = r * o_cosf(theta);
            a[i].im = r * o_sinf(theta);
        }
    }
    o_ifft(a, length);
    const float z = (float)length;
    for (size_t k = 0; k < length; k++) out[k] = a[k].im * z;
    free(a);
    return length;
}

/* Wavetable::new(20.0, 20_000.0, 4.0, phase, amplitude) wavetable.rs:91-123 for the built-in table `kind`
 * (0 saw, 1 square, 2 triangle, 4 organ, 5 soft saw, 6 hammond).  Fills pitches / lengths / data (tables concatenated)
 * and returns the number of tables, or -1 if a capacity is too small. */
int o_make_wavetable(int kind, int max_tables, float *pitches, int *lengths, size_t data_cap, float *data) {
    const double p_factor = pow(2.0, 1.0 / 4.0);
    float max_amplitude = 0.0f;
    size_t total = 0;
    int n = 0;
    float *wave = (float *)malloc(8192 * sizeof(float));
    for (double pitch = 20.0; pitch <= 20000.0; pitch *= p_factor) {
        const size_t len = o_make_wave(kind, pitch, wave);
        if (n >= max_tables || total + len > data_cap) {
            free(wave);
            return -1;
        }
        for (size_t k = 0; k < len; k++) {
            max_amplitude = fmaxf(max_amplitude, fabsf(wave[k]));
            data[total + k] = wave[k];
        }
        pitches[n] = (float)pitch;
        lengths[n] = (int)len;
        total += len;
        n++;
    }
    free(wave);
    if (max_amplitude > 0.0f) {
        const float z = 1.0f / max_amplitude;
        for (size_t k = 0; k < total; k++) data[k] *= z;
    }
    return n;
}
