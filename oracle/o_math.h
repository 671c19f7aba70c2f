/*
 * oracle/o_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
 *
 * Scalar restatement of the transcendental / hashing substrate FunDSP's hot path depends on.
 *
 * FunDSP (reference @ /root/reference, crate 0.23.0) maps its `Float`/`Real` traits to two third-party
 * crates whose sources are NOT vendored under /root/reference (no Cargo.lock; semver minimums from
 * Cargo.toml:15-28):
 *   - `libm = 0.2.15`  : scalar f32 path  (src/lib.rs:180-222, 444-492, 773-798)  -> sinf cosf tanf tanhf expf floorf ...
 *   - `wide = 1.1.1`   : f32x8 SIMD path  (src/lib.rs:296-335, 596-670)           -> f32x8::sin, round ...
 *
 * `libm` 0.2.x is a line-by-line port of musl libc's math (itself FreeBSD msun): sinf.c/cosf.c/tanf.c with
 * the double-precision kernels __sindf/__cosdf/__tandf, __rem_pio2f, expm1f.c, tanhf.c, the pre-2019 expf.c.
 * `wide`'s f32x8::sin_cos is a port of Agner Fog's vectorclass `sincos_f` (vectormath_trig.h): Cody-Waite
 * 3-constant reduction + degree-2 polynomials in x^2 (Cephes single-precision coefficients).
 * Those published algorithms are restated here.  PARITY UNPINNED at the bit level for these functions: the
 * crate sources are absent and there is no Rust toolchain in the build image, so the constants below are
 * validated only (a) against their own decimal comments, (b) to < 1 ulp against double-precision libm in
 * tests/test_oracle_math.py, and (c) through the reference's 1e-4 / 2e-4 test tolerances.
 *
 * Everything is compiled with -ffp-contract=off: Rust never contracts a*b+c.
 */
#ifndef FUNDSP_ORACLE_MATH_H
#define FUNDSP_ORACLE_MATH_H

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t o_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- musl __sindf / __cosdf / __tandf (k_sinf.c, k_cosf.c, k_tanf.c) -------------------------------- */
static inline float o_k_sindf(double x) {
    static const double S1 = -0x15555554cbac77.0p-55, /* -0.166666666416265235595 */
        S2 = 0x111110896efbb2.0p-59,                  /*  0.0083333293858894631756 */
        S3 = -0x1a00f9e2cae774.0p-65,                 /* -0.000198393348360966317347 */
        S4 = 0x16cd878c3b46a7.0p-71;                  /*  0.0000027183114939898219064 */
    double r, s, w, z;
    z = x * x;
    w = z * z;
    r = S3 + z * S4;
    s = z * x;
    return (float)((x + s * (S1 + z * S2)) + s * w * r);
}

static inline float o_k_cosdf(double x) {
    static const double C0 = -0x1ffffffd0c5e81.0p-54, /* -0.499999997251031003120 */
        C1 = 0x155553e1053a42.0p-57,                  /*  0.0416666233237390631894 */
        C2 = -0x16c087e80f1e27.0p-62,                 /* -0.00138867637746099294692 */
        C3 = 0x199342e0ee5069.0p-68;                  /*  0.0000243904487962774090654 */
    double r, w, z;
    z = x * x;
    w = z * z;
    r = C2 + z * C3;
    return (float)(((1.0 + z * C0) + w * C1) + (w * z) * r);
}

static inline float o_k_tandf(double x, int odd) {
    static const double T[] = {
        0x15554d3418c99f.0p-54, /* 0.333331395030791399758 */
        0x1112fd38999f72.0p-55, /* 0.133392002712976742718 */
        0x1b54c91d865afe.0p-57, /* 0.0533812378445670393523 */
        0x191df3908c33ce.0p-58, /* 0.0245283181166547278873 */
        0x185dadfcecf44e.0p-61, /* 0.00297435743359967304927 */
        0x1362b9bf971bcd.0p-59, /* 0.00946564784943673166728 */
    };
    double z, r, w, s, t, u;
    z = x * x;
    r = T[4] + z * T[5];
    t = T[2] + z * T[3];
    w = z * z;
    s = z * x;
    u = T[0] + z * T[1];
    r = (x + s * u) + (s * w) * (t + w * r);
    return (float)(odd ? -1.0 / r : r);
}

/* musl __rem_pio2_large.c (FreeBSD k_rem_pio2.c; libm crate: src/math/rem_pio2_large.rs), the Payne-Hanek reduction
 * behind |x| >= 2^28*pi/2.  Restated with its general loops; the callers here are the f32 functions, which pass ONE
 * 24-bit chunk (nx = 1) at prec = 0 (jk = 3).  ipio2[] = the bits of 2/pi in 24-bit pieces (regenerated from a
 * big-integer pi, first entries 0xA2F983 0x6E4E44 ... as published); PIo2[] = pi/2 in 24-bit-mantissa pieces. */
static const int32_t o_ipio2[66] = {
    0xA2F983, 0x6E4E44, 0x1529FC, 0x2757D1, 0xF534DD, 0xC0DB62, 0x95993C, 0x439041, 0xFE5163, 0xABDEBB, 0xC561B7,
    0x246E3A, 0x424DD2, 0xE00649, 0x2EEA09, 0xD1921C, 0xFE1DEB, 0x1CB129, 0xA73EE8, 0x8235F5, 0x2EBB44, 0x84E99C,
    0x7026B4, 0x5F7E41, 0x3991D6, 0x398353, 0x39F49C, 0x845F8B, 0xBDF928, 0x3B1FF8, 0x97FFDE, 0x05980F, 0xEF2F11,
    0x8B5A0A, 0x6D1F6D, 0x367ECF, 0x27CB09, 0xB74F46, 0x3F669E, 0x5FEA2D, 0x7527BA, 0xC7EBE5, 0xF17B3D, 0x0739F7,
    0x8A5292, 0xEA6BFB, 0x5FB11F, 0x8D5D08, 0x560330, 0x46FC7B, 0x6BABF0, 0xCFBC20, 0x9AF436, 0x1DA9E3, 0x91615E,
    0xE61B08, 0x659985, 0x5F14A0, 0x68408D, 0xFFD880, 0x4D7327, 0x310606, 0x1556CA, 0x73A8C9, 0x60E27B, 0xC08C6B};
static const double o_PIo2[8] = {
    1.57079625129699707031e+00, /* 0x3FF921FB, 0x40000000 */
    7.54978941586159635335e-08, /* 0x3E74442D, 0x00000000 */
    5.39030252995776476554e-15, /* 0x3CF84698, 0x80000000 */
    3.28200341580791294123e-22, /* 0x3B78CC51, 0x60000000 */
    1.27065575308067607349e-29, /* 0x39F01B83, 0x80000000 */
    1.22933308981111328932e-36, /* 0x387A2520, 0x40000000 */
    2.73370053816464559624e-44, /* 0x36E38222, 0x80000000 */
    2.16741683877804819444e-51, /* 0x3569F31D, 0x00000000 */
};

static inline int o_rem_pio2_large(const double *x, double *y, int e0, int nx) {
    const int jk = 3, jp = 3; /* init_jk[prec = 0] */
    int32_t jz, jx, jv, carry, n, iq[20], i, j, k, m, q0, ih;
    double z, fw, f[20], fq[20], q[20];

    /* determine jx, jv, q0 (note that 3 > q0) */
    jx = nx - 1;
    jv = (e0 - 3) / 24;
    if (jv < 0) jv = 0;
    q0 = e0 - 24 * (jv + 1);

    /* set up f[0] to f[jx+jk] where f[jx+jk] = ipio2[jv+jk] */
    j = jv - jx;
    m = jx + jk;
    for (i = 0; i <= m; i++, j++) f[i] = j < 0 ? 0.0 : (double)o_ipio2[j];

    /* compute q[0], q[1], ... q[jk] */
    for (i = 0; i <= jk; i++) {
        for (j = 0, fw = 0.0; j <= jx; j++) fw += x[j] * f[jx + i - j];
        q[i] = fw;
    }

    jz = jk;
recompute:
    /* distill q[] into iq[] reversingly */
    for (i = 0, j = jz, z = q[jz]; j > 0; i++, j--) {
        fw = (double)(int32_t)(0x1p-24 * z);
        iq[i] = (int32_t)(z - 0x1p24 * fw);
        z = q[j - 1] + fw;
    }

    /* compute n */
    z = scalbn(z, q0);           /* actual value of z */
    z -= 8.0 * floor(z * 0.125); /* trim off integer >= 8 */
    n = (int32_t)z;
    z -= (double)n;
    ih = 0;
    if (q0 > 0) { /* need iq[jz-1] to determine n */
        i = iq[jz - 1] >> (24 - q0);
        n += i;
        iq[jz - 1] -= i << (24 - q0);
        ih = iq[jz - 1] >> (23 - q0);
    } else if (q0 == 0)
        ih = iq[jz - 1] >> 23;
    else if (z >= 0.5)
        ih = 2;

    if (ih > 0) { /* q > 0.5 */
        n += 1;
        carry = 0;
        for (i = 0; i < jz; i++) { /* compute 1-q */
            j = iq[i];
            if (carry == 0) {
                if (j != 0) {
                    carry = 1;
                    iq[i] = 0x1000000 - j;
                }
            } else
                iq[i] = 0xffffff - j;
        }
        if (q0 > 0) { /* rare case: chance is 1 in 12 */
            switch (q0) {
            case 1: iq[jz - 1] &= 0x7fffff; break;
            case 2: iq[jz - 1] &= 0x3fffff; break;
            }
        }
        if (ih == 2) {
            z = 1.0 - z;
            if (carry != 0) z -= scalbn(1.0, q0);
        }
    }

    /* check if recomputation is needed */
    if (z == 0.0) {
        j = 0;
        for (i = jz - 1; i >= jk; i--) j |= iq[i];
        if (j == 0) { /* need recomputation */
            for (k = 1; iq[jk - k] == 0; k++) {} /* k = no. of terms needed */
            for (i = jz + 1; i <= jz + k; i++) { /* add q[jz+1] to q[jz+k] */
                f[jx + i] = (double)o_ipio2[jv + i];
                for (j = 0, fw = 0.0; j <= jx; j++) fw += x[j] * f[jx + i - j];
                q[i] = fw;
            }
            jz += k;
            goto recompute;
        }
    }

    /* chop off zero terms */
    if (z == 0.0) {
        jz -= 1;
        q0 -= 24;
        while (iq[jz] == 0) {
            jz--;
            q0 -= 24;
        }
    } else { /* break z into 24-bit if necessary */
        z = scalbn(z, -q0);
        if (z >= 0x1p24) {
            fw = (double)(int32_t)(0x1p-24 * z);
            iq[jz] = (int32_t)(z - 0x1p24 * fw);
            jz += 1;
            q0 += 24;
            iq[jz] = (int32_t)fw;
        } else
            iq[jz] = (int32_t)z;
    }

    /* convert integer "bit" chunk to floating-point value */
    fw = scalbn(1.0, q0);
    for (i = jz; i >= 0; i--) {
        q[i] = fw * (double)iq[i];
        fw *= 0x1p-24;
    }

    /* compute PIo2[0,...,jp]*q[jz,...,0] */
    for (i = jz; i >= 0; i--) {
        for (fw = 0.0, k = 0; k <= jp && k <= jz - i; k++) fw += o_PIo2[k] * q[i + k];
        fq[jz - i] = fw;
    }

    /* compress fq[] into y[] (prec 0) */
    fw = 0.0;
    for (i = jz; i >= 0; i--) fw += fq[i];
    y[0] = ih == 0 ? fw : -fw;
    return n & 7;
}

/* musl __rem_pio2f.c (libm crate: src/math/rem_pio2f.rs): medium branch for |x| < 2^28*pi/2, inf/NaN -> NaN,
 * everything else through __rem_pio2_large on the mantissa scaled into [2^23, 2^24). */
static inline int o_rem_pio2f(float x, double *y) {
    static const double toint = 1.5 / DBL_EPSILON, invpio2 = 6.36619772367581382433e-01, /* 0x3FE45F30, 0x6DC9C883 */
        pio2_1 = 1.57079631090164184570e+00,                                           /* 0x3FF921FB, 0x50000000 */
        pio2_1t = 1.58932547735281966916e-08;                                          /* 0x3E5110b4, 0x611A6263 */
    uint32_t ix = o_f2u(x) & 0x7fffffff;
    double tx[1], ty[1];
    int n, sign, e0;
    if (ix < 0x4dc90fdb) { /* |x| ~< 2^28*(pi/2), medium size */
        double fn = (double)x * invpio2 + toint - toint;
        n = (int32_t)fn;
        *y = x - fn * pio2_1 - fn * pio2_1t;
        return n;
    }
    if (ix >= 0x7f800000) { /* x is inf or NaN */
        *y = x - x;
        return 0;
    }
    /* scale x into [2^23, 2^24-1] */
    sign = (int)(o_f2u(x) >> 31);
    e0 = (int)(ix >> 23) - (0x7f + 23); /* e0 = ilogb(|x|)-23, positive */
    tx[0] = (double)o_u2f(ix - ((uint32_t)e0 << 23));
    n = o_rem_pio2_large(tx, ty, e0, 1);
    if (sign) {
        *y = -ty[0];
        return -n;
    }
    *y = ty[0];
    return n;
}

#define O_PIO2 1.570796326794896558e+00 /* M_PI_2 */

/* musl sinf.c */
static inline float o_sinf(float x) {
    static const double s1pio2 = 1 * O_PIO2, s2pio2 = 2 * O_PIO2, s3pio2 = 3 * O_PIO2, s4pio2 = 4 * O_PIO2;
    double y;
    uint32_t ix = o_f2u(x);
    int n, sign = ix >> 31;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {    /* |x| ~<= pi/4 */
        if (ix < 0x39800000) { /* |x| < 2**-12 */
            return x;
        }
        return o_k_sindf(x);
    }
    if (ix <= 0x407b53d1) {     /* |x| ~<= 5*pi/4 */
        if (ix <= 0x4016cbe3) { /* |x| ~<= 3pi/4 */
            if (sign)
                return -o_k_cosdf(x + s1pio2);
            else
                return o_k_cosdf(x - s1pio2);
        }
        return o_k_sindf(sign ? -(x + s2pio2) : -(x - s2pio2));
    }
    if (ix <= 0x40e231d5) {     /* |x| ~<= 9*pi/4 */
        if (ix <= 0x40afeddf) { /* |x| ~<= 7*pi/4 */
            if (sign)
                return o_k_cosdf(x + s3pio2);
            else
                return -o_k_cosdf(x - s3pio2);
        }
        return o_k_sindf(sign ? x + s4pio2 : x - s4pio2);
    }
    if (ix >= 0x7f800000) return x - x;
    n = o_rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return o_k_sindf(y);
    case 1: return o_k_cosdf(y);
    case 2: return o_k_sindf(-y);
    default: return -o_k_cosdf(y);
    }
}

/* musl cosf.c */
static inline float o_cosf(float x) {
    static const double c1pio2 = 1 * O_PIO2, c2pio2 = 2 * O_PIO2, c3pio2 = 3 * O_PIO2, c4pio2 = 4 * O_PIO2;
    double y;
    uint32_t ix = o_f2u(x);
    unsigned n, sign = ix >> 31;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {    /* |x| ~<= pi/4 */
        if (ix < 0x39800000) { /* |x| < 2**-12 */
            return 1.0f;
        }
        return o_k_cosdf(x);
    }
    if (ix <= 0x407b53d1) {    /* |x| ~<= 5*pi/4 */
        if (ix > 0x4016cbe3)   /* |x|  ~> 3*pi/4 */
            return -o_k_cosdf(sign ? x + c2pio2 : x - c2pio2);
        else {
            if (sign)
                return o_k_sindf(x + c1pio2);
            else
                return o_k_sindf(c1pio2 - x);
        }
    }
    if (ix <= 0x40e231d5) {  /* |x| ~<= 9*pi/4 */
        if (ix > 0x40afeddf) /* |x| ~> 7*pi/4 */
            return o_k_cosdf(sign ? x + c4pio2 : x - c4pio2);
        else {
            if (sign)
                return o_k_sindf(-x - c3pio2);
            else
                return o_k_sindf(x - c3pio2);
        }
    }
    if (ix >= 0x7f800000) return x - x;
    n = (unsigned)o_rem_pio2f(x, &y);
    switch (n & 3) {
    case 0: return o_k_cosdf(y);
    case 1: return o_k_sindf(-y);
    case 2: return -o_k_cosdf(y);
    default: return o_k_sindf(y);
    }
}

/* musl tanf.c */
static inline float o_tanf(float x) {
    static const double t1pio2 = 1 * O_PIO2, t2pio2 = 2 * O_PIO2, t3pio2 = 3 * O_PIO2, t4pio2 = 4 * O_PIO2;
    double y;
    uint32_t ix = o_f2u(x);
    unsigned n, sign = ix >> 31;
    ix &= 0x7fffffff;
    if (ix <= 0x3f490fda) {    /* |x| ~<= pi/4 */
        if (ix < 0x39800000) { /* |x| < 2**-12 */
            return x;
        }
        return o_k_tandf(x, 0);
    }
    if (ix <= 0x407b53d1) {    /* |x| ~<= 5*pi/4 */
        if (ix <= 0x4016cbe3)  /* |x| ~<= 3pi/4 */
            return o_k_tandf((sign ? x + t1pio2 : x - t1pio2), 1);
        else
            return o_k_tandf((sign ? x + t2pio2 : x - t2pio2), 0);
    }
    if (ix <= 0x40e231d5) {    /* |x| ~<= 9*pi/4 */
        if (ix <= 0x40afeddf)  /* |x| ~<= 7*pi/4 */
            return o_k_tandf((sign ? x + t3pio2 : x - t3pio2), 1);
        else
            return o_k_tandf((sign ? x + t4pio2 : x - t4pio2), 0);
    }
    if (ix >= 0x7f800000) return x - x;
    n = (unsigned)o_rem_pio2f(x, &y);
    return o_k_tandf(y, n & 1);
}

/* musl expm1f.c (FreeBSD s_expm1f.c) */
static inline float o_expm1f(float x) {
    static const float ln2_hi = 6.9313812256e-01f, /* 0x3f317180 */
        ln2_lo = 9.0580006145e-06f,                /* 0x3717f7d1 */
        invln2 = 1.4426950216e+00f,                /* 0x3fb8aa3b */
        Q1 = -3.3333212137e-2f,                    /* -0x888868.0p-28 */
        Q2 = 1.5807170421e-3f;                     /*  0xcf3010.0p-33 */
    float y, hi, lo, c = 0.0f, t, e, hxs, hfx, r1, twopk;
    uint32_t ui = o_f2u(x);
    uint32_t hx = ui & 0x7fffffff;
    int k, sign = ui >> 31;

    if (hx >= 0x4195b844) { /* if |x|>=27*ln2 */
        if (hx > 0x7f800000) return x;
        if (sign) return -1.0f;
        if (x > 8.8721679688e+01f) { /* o_threshold 0x42b17180 */
            x *= 0x1p127f;
            return x;
        }
    }
    if (hx > 0x3eb17218) {     /* if  |x| > 0.5 ln2 */
        if (hx < 0x3F851592) { /* and |x| < 1.5 ln2 */
            if (!sign) {
                hi = x - ln2_hi;
                lo = ln2_lo;
                k = 1;
            } else {
                hi = x + ln2_hi;
                lo = -ln2_lo;
                k = -1;
            }
        } else {
            k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
            t = (float)k;
            hi = x - t * ln2_hi; /* t*ln2_hi is exact here */
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000) { /* when |x|<2**-25, return x */
        return x;
    } else
        k = 0;

    hfx = 0.5f * x;
    hxs = x * hfx;
    r1 = 1.0f + hxs * (Q1 + hxs * Q2);
    t = 3.0f - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0f - x * t));
    if (k == 0) /* c is 0 */
        return x - (x * e - hxs);
    e = x * (e - c) - c;
    e -= hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) {
        if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
        return 1.0f + 2.0f * (x - e);
    }
    twopk = o_u2f((uint32_t)(0x7f + k) << 23); /* 2^k */
    if (k < 0 || k > 56) {                     /* suffice to return exp(x)-1 */
        y = x - e + 1.0f;
        if (k == 128)
            y = y * 2.0f * 0x1p127f;
        else
            y = y * twopk;
        return y - 1.0f;
    }
    float uf = o_u2f((uint32_t)(0x7f - k) << 23); /* 2^-k */
    if (k < 23)
        y = (x - e + (1 - uf)) * twopk;
    else
        y = (x - (e + uf) + 1) * twopk;
    return y;
}

/* musl tanhf.c */
static inline float o_tanhf(float x) {
    uint32_t w = o_f2u(x);
    int sign = w >> 31;
    float t;
    w &= 0x7fffffff;
    x = o_u2f(w);
    if (w > 0x3f0c9f54) {     /* |x| > log(3)/2 ~= 0.5493 or nan */
        if (w > 0x41200000) { /* |x| > 10 */
            t = 1 + 0 / x;
        } else {
            t = o_expm1f(2 * x);
            t = 1 - 2 / (t + 2);
        }
    } else if (w > 0x3e82c578) { /* |x| > log(5/3)/2 ~= 0.2554 */
        t = o_expm1f(2 * x);
        t = t / (t + 2);
    } else if (w >= 0x00800000) { /* |x| >= 0x1p-126 */
        t = o_expm1f(-2 * x);
        t = -t / (t + 2);
    } else { /* |x| is subnormal */
        t = x;
    }
    return sign ? -t : t;
}

/* musl atanf.c (FreeBSD s_atanf.c) */
static inline float o_atanf(float x) {
    static const float atanhi[] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static const float atanlo[] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    static const float aT[] = {3.3333328366e-01f, -1.9999158382e-01f, 1.4253635705e-01f, -1.0648017377e-01f, 6.1687607318e-02f};
    float w, s1, s2, z;
    uint32_t ix = o_f2u(x), sign = ix >> 31;
    int id;
    ix &= 0x7fffffff;
    if (ix >= 0x4c800000) { /* if |x| >= 2**26 */
        if (ix > 0x7f800000) return x;
        z = atanhi[3] + 0x1p-120f;
        return sign ? -z : z;
    }
    if (ix < 0x3ee00000) {     /* |x| < 0.4375 */
        if (ix < 0x39800000) { /* |x| < 2**-12 */
            return x;
        }
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {     /* |x| < 1.1875 */
            if (ix < 0x3f300000) { /*  7/16 <= |x| < 11/16 */
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else { /* 11/16 <= |x| < 19/16 */
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000) { /* |x| < 2.4375 */
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else { /* 2.4375 <= |x| < 2**26 */
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * aT[4]));
    s2 = w * (aT[1] + w * aT[3]);
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return sign ? -z : z;
}

/* wide 1.1.1 f32x8::atan, one lane (vectorclass atan_f), unfused */
static inline float o_wide_atanf(float self) {
    const float P3 = 8.05374449538E-2f, P2 = -1.38776856032E-1f, P1 = 1.99777106478E-1f, P0 = -3.33329491539E-1f;
    const float SQRT2 = 1.41421356237309504880f, FRAC_PI_2 = 1.57079632679489661923f, FRAC_PI_4 = 0.785398163397448309616f;
    float t = fabsf(self);
    int notsmal = t >= SQRT2 - 1.0f;
    int notbig = t <= SQRT2 + 1.0f;
    float s = notbig ? FRAC_PI_4 : FRAC_PI_2;
    s = notsmal ? s : 0.0f;
    float a = notbig ? t : 0.0f;
    a = notsmal ? a - 1.0f : a;
    float b = notbig ? 1.0f : 0.0f;
    b = notsmal ? b + t : b;
    float z = a / b;
    float zz = z * z;
    float z4 = zz * zz;
    float re = (zz * P3 + P2) * z4 + (zz * P1 + P0); /* polynomial_3!(zz, P0, P1, P2, P3) */
    re = re * (zz * z) + z + s;
    return (o_f2u(self) >> 31) ? -re : re;
}

/* musl expf.c as of the 2018 port (FreeBSD e_expf.c; libm 0.2 expf.rs) */
static inline float o_expf(float x) {
    static const float half[2] = {0.5f, -0.5f}, ln2hi = 6.9314575195e-1f, /* 0x3f317200 */
        ln2lo = 1.4286067653e-6f,                                         /* 0x35bfbe8e */
        invln2 = 1.4426950216e+0f,                                        /* 0x3fb8aa3b */
        P1 = 1.6666625440e-1f,                                            /*  0xaaaa8f.0p-26 */
        P2 = -2.7667332906e-3f;                                           /* -0xb55215.0p-32 */
    float hi, lo, c, xx, y;
    int k, sign;
    uint32_t hx = o_f2u(x);
    sign = hx >> 31;
    hx &= 0x7fffffff;
    if (hx >= 0x42aeac50) { /* if |x| >= -87.33655f or NaN */
        if (hx > 0x7f800000) return x;
        if (hx >= 0x42b17218 && !sign) { /* x >= 88.722839f */
            x *= 0x1p127f;
            return x;
        }
        if (sign) {
            if (hx >= 0x42cff1b5) return 0; /* x <= -103.972084f */
        }
    }
    if (hx > 0x3eb17218) { /* if |x| > 0.5 ln2 */
        if (hx > 0x3f851592) /* if |x| > 1.5 ln2 */
            k = (int)(invln2 * x + half[sign]);
        else
            k = 1 - sign - sign;
        hi = x - k * ln2hi; /* k*ln2hi is exact here */
        lo = k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000) { /* |x| > 2**-14 */
        k = 0;
        hi = x;
        lo = 0;
    } else {
        return 1 + x;
    }
    xx = x * x;
    c = x - xx * (P1 + xx * P2);
    y = 1 + (x * c / (2 - c) - lo + hi);
    if (k == 0) return y;
    return scalbnf(y, k);
}

/* musl scalbnf / powf as of the 2018 port (FreeBSD e_powf.c; libm 0.2.15 scalbnf.rs, powf.rs).  Used by Dsf
 * (oscillator.rs:105-113).  Restated from the published algorithm: parity unpinned at bit level; accuracy is checked
 * against double pow in tests/test_oracle_math.py (< 1 ulp). */
static inline float o_scalbnf(float x, int n) {
    float y = x; /* musl scalbnf.c (libm 0.2 scalbnf.rs): two-step scaling, no double rounding into the subnormals */
    if (n > 127) {
        y *= 0x1p127f;
        n -= 127;
        if (n > 127) {
            y *= 0x1p127f;
            n -= 127;
            if (n > 127) n = 127;
        }
    } else if (n < -126) {
        y *= 0x1p-126f * 0x1p24f;
        n += 126 - 24;
        if (n < -126) {
            y *= 0x1p-126f * 0x1p24f;
            n += 126 - 24;
            if (n < -126) n = -126;
        }
    }
    return y * o_u2f((uint32_t)(0x7f + n) << 23);
}
static inline float o_powf(float x, float y) {
    const float bp[2] = {1.0f, 1.5f}, dp_h[2] = {0.0f, 5.84960938e-01f}, dp_l[2] = {0.0f, 1.56322085e-06f};
    const float two24 = 16777216.0f, huge = 1.0e30f, tiny = 1.0e-30f;
    const float L1 = 6.0000002384e-01f, L2 = 4.2857143283e-01f, L3 = 3.3333334327e-01f, L4 = 2.7272811532e-01f,
                L5 = 2.3066075146e-01f, L6 = 2.0697501302e-01f;
    const float P1 = 1.6666667163e-01f, P2 = -2.7777778450e-03f, P3 = 6.6137559770e-05f, P4 = -1.6533901999e-06f,
                P5 = 4.1381369442e-08f;
    const float lg2 = 6.9314718246e-01f, lg2_h = 6.93145752e-01f, lg2_l = 1.42860654e-06f, ovt = 4.2995665694e-08f;
    const float cp = 9.6179670095e-01f, cp_h = 9.6191406250e-01f, cp_l = -1.1736857402e-04f;
    const float ivln2 = 1.4426950216e+00f, ivln2_h = 1.4426879883e+00f, ivln2_l = 7.0526075433e-06f;
    float z, ax, z_h, z_l, p_h, p_l, y1, t1, t2, r, s, sn, t, u, v, w;
    int32_t i, j, k, yisint, n, hx, hy, ix, iy, is;
    hx = (int32_t)o_f2u(x);
    hy = (int32_t)o_f2u(y);
    ix = hx & 0x7fffffff;
    iy = hy & 0x7fffffff;
    if (iy == 0) return 1.0f;                              /* x**0 = 1, even if x is NaN */
    if (hx == 0x3f800000) return 1.0f;                     /* 1**y = 1, even if y is NaN */
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  /* NaN if either arg is NaN */
    yisint = 0;                                            /* is y an odd / even integer (only matters for x < 0) */
    if (hx < 0) {
        if (iy >= 0x4b800000) yisint = 2;
        else if (iy >= 0x3f800000) {
            k = (iy >> 23) - 0x7f;
            j = iy >> (23 - k);
            if ((j << (23 - k)) == iy) yisint = 2 - (j & 1);
        }
    }
    if (iy == 0x7f800000) { /* y is +-inf */
        if (ix == 0x3f800000) return 1.0f;
        else if (ix > 0x3f800000) return hy >= 0 ? y : 0.0f;
        else return hy >= 0 ? 0.0f : -y;
    }
    if (iy == 0x3f800000) return hy >= 0 ? x : 1.0f / x; /* y is +-1 */
    if (hy == 0x40000000) return x * x;                   /* y is 2 */
    if (hy == 0x3f000000) {                               /* y is 0.5 */
        if (hx >= 0) return sqrtf(x);
    }
    ax = fabsf(x);
    if (ix == 0x7f800000 || ix == 0 || ix == 0x3f800000) { /* x is +-0, +-inf, +-1 */
        z = ax;
        if (hy < 0) z = 1.0f / z;
        if (hx < 0) {
            if (((ix - 0x3f800000) | yisint) == 0) z = (z - z) / (z - z);
            else if (yisint == 1) z = -z;
        }
        return z;
    }
    sn = 1.0f; /* sign of the result */
    if (hx < 0) {
        if (yisint == 0) return (x - x) / (x - x);
        if (yisint == 1) sn = -1.0f;
    }
    if (iy > 0x4d000000) { /* |y| > 2**27 */
        if (ix < 0x3f7ffff8) return hy < 0 ? sn * huge * huge : sn * tiny * tiny;
        if (ix > 0x3f800007) return hy > 0 ? sn * huge * huge : sn * tiny * tiny;
        t = ax - 1;
        w = (t * t) * (0.5f - t * (0.333333333333f - t * 0.25f));
        u = ivln2_h * t;
        v = t * ivln2_l - w * ivln2;
        t1 = u + v;
        is = (int32_t)o_f2u(t1);
        t1 = o_u2f((uint32_t)is & 0xfffff000u);
        t2 = v - (t1 - u);
    } else {
        float s2, s_h, s_l, t_h, t_l;
        n = 0;
        if (ix < 0x00800000) { /* subnormal x */
            ax *= two24;
            n -= 24;
            ix = (int32_t)o_f2u(ax);
        }
        n += ((ix) >> 23) - 0x7f;
        j = ix & 0x007fffff;
        ix = j | 0x3f800000; /* normalize ix */
        if (j <= 0x1cc471) k = 0;      /* |x| < sqrt(3/2) */
        else if (j < 0x5db3d7) k = 1;  /* |x| < sqrt(3)   */
        else {
            k = 0;
            n += 1;
            ix -= 0x00800000;
        }
        ax = o_u2f((uint32_t)ix);
        u = ax - bp[k]; /* s = s_h + s_l = (x-1)/(x+1) or (x-1.5)/(x+1.5) */
        v = 1.0f / (ax + bp[k]);
        s = u * v;
        s_h = s;
        is = (int32_t)o_f2u(s_h);
        s_h = o_u2f((uint32_t)is & 0xfffff000u);
        is = (int32_t)((((uint32_t)ix >> 1) & 0xfffff000u) | 0x20000000u); /* t_h = ax + bp[k], high part */
        t_h = o_u2f((uint32_t)(is + 0x00400000 + (k << 21)));
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        s2 = s * s; /* log(ax) */
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + s);
        s2 = s_h * s_h;
        t_h = 3.0f + s2 + r;
        is = (int32_t)o_f2u(t_h);
        t_h = o_u2f((uint32_t)is & 0xfffff000u);
        t_l = r - ((t_h - 3.0f) - s2);
        u = s_h * t_h; /* u + v = s * (1 + ...) */
        v = s_l * t_h + t_l * s;
        p_h = u + v; /* 2/(3 log2) * (s + ...) */
        is = (int32_t)o_f2u(p_h);
        p_h = o_u2f((uint32_t)is & 0xfffff000u);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h; /* cp_h + cp_l = 2/(3 log2) */
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        t = (float)n; /* log2(ax) = (s + ..) * 2/(3 log2) = n + dp_h + z_h + z_l */
        t1 = (((z_h + z_l) + dp_h[k]) + t);
        is = (int32_t)o_f2u(t1);
        t1 = o_u2f((uint32_t)is & 0xfffff000u);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }
    is = (int32_t)o_f2u(y); /* split y into y1 + y2 and compute (y1 + y2) * (t1 + t2) */
    y1 = o_u2f((uint32_t)is & 0xfffff000u);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    j = (int32_t)o_f2u(z);
    if (j > 0x43000000) return sn * huge * huge; /* z > 128: overflow */
    else if (j == 0x43000000) {                  /* z == 128 */
        if (p_l + ovt > z - p_h) return sn * huge * huge;
    } else if ((j & 0x7fffffff) > 0x43160000) return sn * tiny * tiny; /* z < -150: underflow */
    else if ((uint32_t)j == 0xc3160000u) {                              /* z == -150 */
        if (p_l <= z - p_h) return sn * tiny * tiny;
    }
    i = j & 0x7fffffff; /* 2**(p_h + p_l) */
    k = (i >> 23) - 0x7f;
    n = 0;
    if (i > 0x3f000000) { /* |z| > 0.5: n = [z + 0.5] */
        n = j + (0x00800000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 23) - 0x7f; /* new k for n */
        t = o_u2f((uint32_t)(n & ~(0x007fffff >> k)));
        n = ((n & 0x007fffff) | 0x00800000) >> (23 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = p_l + p_h;
    is = (int32_t)o_f2u(t);
    t = o_u2f((uint32_t)is & 0xffff8000u);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / (t1 - 2.0f) - (w + z * w);
    z = 1.0f - (r - z);
    j = (int32_t)o_f2u(z);
    j += (int32_t)((uint32_t)n << 23);
    if ((j >> 23) <= 0) z = o_scalbnf(z, n); /* subnormal output */
    else z = o_u2f((uint32_t)j);
    return sn * z;
}

/* ---- wide 1.1.1 f32x8::sin, one lane (vectorclass sincos_f). mul_add / mul_neg_add are UNFUSED: that is
 *      what `wide` compiles to for a default x86-64 `cargo build` (no `fma` target feature). ---------- */
static inline float o_round_half_even(float x) { return nearbyintf(x); } /* default rounding mode */

/* wide's f32x8::round_int (NOT fast_round_int) is defined for every input: its x86 form masks NaN lanes to 0 and
 * flips lanes >= 2^31 to i32::MAX around cvtps2dq (the sequence it cites from V8), its wasm / NEON / portable forms
 * are saturating conversions -- so: NaN -> 0, >= 2^31 -> i32::MAX, <= -2^31 -> i32::MIN on every platform. */
static inline int32_t o_round_int_sat(float y) {
    if (y != y) return 0;
    if (y >= 2147483648.0f) return INT32_MAX;
    if (y <= -2147483648.0f) return INT32_MIN;
    return (int32_t)y;
}

static inline float o_wide_sinf(float self) {
    const float DP1F = 0.78515625f * 2.0f;
    const float DP2F = 2.4187564849853515625E-4f * 2.0f;
    const float DP3F = 3.77489497744594108E-8f * 2.0f;
    const float P0sinf = -1.6666654611E-1f, P1sinf = 8.3321608736E-3f, P2sinf = -1.9515295891E-4f;
    const float P0cosf = 4.166664568298827E-2f, P1cosf = -1.388731625493765E-3f, P2cosf = 2.443315711809948E-5f;
    const float TWO_OVER_PI = 2.0f / 3.14159274101257324f; /* 2.0 / core::f32::consts::PI, evaluated in f32 */

    float xa = fabsf(self);
    float y = o_round_half_even(xa * TWO_OVER_PI);
    int32_t q = o_round_int_sat(y); /* f32x8::round_int of an already-integral value */
    float x = xa - y * DP1F;
    x = x - y * DP2F;
    x = x - y * DP3F;
    float x2 = x * x;
    /* polynomial_2!(x2, c0, c1, c2) = (x2*x2)*c2 + (x2*c1 + c0) */
    float x4 = x2 * x2;
    float s = (x4 * P2sinf + (x2 * P1sinf + P0sinf)) * (x * x2) + x;
    float c = (x4 * P2cosf + (x2 * P1cosf + P0cosf)) * (x2 * x2) + (1.0f - 0.5f * x2);
    int swap = (q & 1) != 0;
    if (q > 0x2000000 && xa < INFINITY) { /* overflow: q unreliable */
        s = 0.0f;
        c = 1.0f;
    }
    float sin1 = swap ? c : s;
    uint32_t sign_sin = ((uint32_t)q << 30) ^ o_f2u(self);
    return o_u2f(o_f2u(sin1) ^ (sign_sin & 0x80000000u));
}

/* ---- integer hashing (bit-exact; src/math.rs:569-576, 592-599, 632-658; src/noise.rs:150-158) ------- */
static inline double o_rnd1(uint64_t x) {
    x = x ^ 0x5555555555555555ULL;
    x = x * 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    x = x ^ (x >> 31);
    return (double)(x >> 11) * (1.0 / (double)(1ULL << 53));
}

static inline uint64_t o_hash1(uint64_t x) {
    x = x ^ 0x5555555555555555ULL;
    x = x * 0x517cc1b727220a95ULL;
    x = (x ^ (x >> 32)) * 0xd6e8feb86659fd93ULL;
    x = (x ^ (x >> 32)) * 0xd6e8feb86659fd93ULL;
    return x ^ (x >> 32);
}

/* AttoHash::hash (math.rs:649-658) */
static inline uint64_t o_atto(uint64_t state, uint64_t data) {
    uint64_t r = (state << 5) | (state >> 59);
    return (r ^ data) * 0x517cc1b727220a95ULL;
}

static inline uint32_t o_hash32x(uint32_t x) {
    const uint32_t MUL_X = 0x45d9f3b;
    x = (x ^ (x >> 16)) * MUL_X;
    x = (x ^ (x >> 16)) * MUL_X;
    return (x ^ (x >> 16)) * MUL_X;
}

#endif
