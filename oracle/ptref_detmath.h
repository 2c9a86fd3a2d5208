/*
 * ptref_detmath.h -- deterministic f32 transcendental functions for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (part of oracle/).  Plain C99, no dependency on the product sources.
 *
 * The reference calls Rust's f32::sin / cos / powf / exp / acos / atan2 (tracer.rs:392,423-424,
 * 437-438,603-605,612-614; scene.rs:64-65,141,144-146), which lower to the platform libm and are
 * therefore not pinned bit-for-bit by the reference.  To make a bit-exact CPU<->GPU comparison
 * possible, the oracle can evaluate those calls with the functions below instead of libm
 * (PTREF_MATH_DET).  Every function is specified as a fixed sequence of IEEE-754 binary64
 * operations (+ - * / , floor, int<->double conversions, bit casts) followed by ONE
 * round-to-nearest conversion to binary32.  No fused multiply-add is used anywhere, so a compiler
 * that does not contract or reassociate (-ffp-contract=off, no fast-math) reproduces the same bits
 * on any IEEE machine.  Evaluated in binary64 the results are within ~1e-15 relative of the true
 * value, i.e. correctly rounded f32 except for arguments whose true result lies within ~1e-8 ulp
 * of a rounding boundary; that is inside the <1 ulp spread between platform libms.
 *
 * Polynomial coefficients for sin/cos/asin and the pi/2, ln2 splittings are the published
 * fdlibm (Sun Microsystems, freely distributable) constants; exp and log use plain Taylor /
 * atanh series.
 */
#ifndef PTREF_DETMATH_H
#define PTREF_DETMATH_H

#include <stdint.h>
#include <string.h>

static inline double ptd_from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t ptd_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline uint32_t ptd_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float ptd_f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* floor() for |x| < 2^51 without libm: truncate toward zero, then fix negatives. */
static inline double ptd_floor(double x) {
    double t = (double)(int64_t)x;
    return (t > x) ? t - 1.0 : t;
}

/* 2^k as a double for -1022 <= k <= 1023. */
static inline double ptd_pow2i(int k) { return ptd_from_bits((uint64_t)(k + 1023) << 52); }

/* ---- sin / cos -------------------------------------------------------------------------- */

/* Reduce x to r in [-pi/4, pi/4] with x = r + q*pi/2; returns q mod 4.  Two-term Cody-Waite with
 * a 33-bit leading part so q*PIO2_1 is exact for |q| < 2^20. */
static inline int ptd_rem_pio2(double x, double *r) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    double q = ptd_floor(x * TWO_OVER_PI + 0.5);
    double t = x - q * PIO2_1;
    *r = t - q * PIO2_1T;
    return (int)((int64_t)q & 3);
}

static inline double ptd_ksin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    return r + (r * z) * p;
}

static inline double ptd_kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    return (1.0 - 0.5 * z) + (z * z) * p;
}

/* Valid (error << 1 f32 ulp) for |x| < 2^20; NaN/Inf -> NaN; larger |x| lose accuracy gradually. */
static inline float ptd_sinf(float x) {
    double xd = (double)x, r;
    if (!(xd > -1.0e15 && xd < 1.0e15)) return x - x; /* NaN, Inf, absurd magnitudes */
    int q = ptd_rem_pio2(xd, &r);
    double v = (q & 1) ? ptd_kcos(r) : ptd_ksin(r);
    if (q & 2) v = -v;
    return (float)v;
}

static inline float ptd_cosf(float x) {
    double xd = (double)x, r;
    if (!(xd > -1.0e15 && xd < 1.0e15)) return x - x;
    int q = ptd_rem_pio2(xd, &r);
    double v = (q & 1) ? ptd_ksin(r) : ptd_kcos(r);
    if (((q + 1) & 2) != 0) v = -v;
    return (float)v;
}

/* ---- exp / log2 / pow ------------------------------------------------------------------- */

/* exp(r) for |r| <= 0.35 by Taylor series to r^14/14!. */
static inline double ptd_kexp(double r) {
    double p = 1.0 / 87178291200.0; /* 1/14! */
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p;
}

/* 2^z as a double, z finite, clamped so the later f32 conversion gives inf / 0 correctly. */
static inline double ptd_exp2d(double z) {
    const double LN2 = 6.93147180559945286227e-01;
    if (z > 200.0) z = 200.0;
    if (z < -200.0) z = -200.0;
    double k = ptd_floor(z + 0.5);
    double f = z - k;
    return ptd_kexp(f * LN2) * ptd_pow2i((int)k);
}

static inline float ptd_expf(float x) {
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double xd = (double)x;
    if (xd != xd) return x;
    if (xd > 100.0) xd = 100.0;   /* exp(100) overflows f32 -> +inf after conversion */
    if (xd < -120.0) xd = -120.0; /* exp(-120) underflows f32 -> 0 after conversion */
    double k = ptd_floor(xd * LOG2E + 0.5);
    double r = (xd - k * LN2_HI) - k * LN2_LO;
    return (float)(ptd_kexp(r) * ptd_pow2i((int)k));
}

/* log2(a) for finite a > 0 (normal double). */
static inline double ptd_log2d(double a) {
    const double LOG2E = 1.44269504088896338700e+00;
    uint64_t b = ptd_to_bits(a);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = ptd_from_bits((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    /* atanh series: ln(m) = 2 s (1 + z/3 + z^2/5 + ... + z^11/23) */
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double lnm = (2.0 * s) * p;
    return (double)e + lnm * LOG2E;
}

/* powf with the C99 Annex F special cases the path can reach. */
static inline float ptd_powf(float x, float y) {
    uint32_t xb = ptd_f32_bits(x), yb = ptd_f32_bits(y);
    uint32_t ax = xb & 0x7fffffffu, ay = yb & 0x7fffffffu;
    if (ay == 0) return 1.0f;                 /* pow(x, +-0) = 1, even for NaN */
    if (xb == 0x3f800000u) return 1.0f;       /* pow(1, y) = 1, even for NaN */
    if (ax > 0x7f800000u || ay > 0x7f800000u) return x + y; /* NaN */
    /* classify y: 0 = not an integer, 1 = odd integer, 2 = even integer */
    int yint = 0;
    if (ay >= 0x4b800000u) yint = 2;          /* |y| >= 2^24 (incl. inf): even integer */
    else if (ay >= 0x3f800000u) {
        int sh = 150 - (int)(ay >> 23);       /* number of fractional mantissa bits, 0..23 */
        uint32_t mant = (ay & 0x007fffffu) | 0x00800000u;
        if (sh <= 0) yint = 2;
        else if ((mant & ((1u << sh) - 1u)) == 0) yint = ((mant >> sh) & 1u) ? 1 : 2;
    }
    int yneg = (yb >> 31) != 0, xneg = (xb >> 31) != 0;
    if (ay == 0x7f800000u) {                  /* y = +-inf */
        if (ax == 0x3f800000u) return 1.0f;   /* pow(-1, +-inf) = 1 */
        int big = ax > 0x3f800000u;
        return (big != yneg) ? ptd_f32_from_bits(0x7f800000u) : 0.0f;
    }
    if (ax == 0 || ax == 0x7f800000u) {       /* x = +-0 or +-inf */
        int to_inf = (ax == 0) ? yneg : !yneg;
        float mag = to_inf ? ptd_f32_from_bits(0x7f800000u) : 0.0f;
        return (xneg && yint == 1) ? -mag : mag;
    }
    if (xneg && yint == 0) return ptd_f32_from_bits(0x7fc00000u); /* negative base, non-integer */
    if (yb == 0x40a00000u) { /* y == 5 (Schlick's Fresnel term): x^5 by three binary64 products; x*x is exact */
        double a = (double)ptd_f32_from_bits(ax);
        double a2 = a * a;
        double v5 = (a2 * a2) * a;
        return (float)(xneg ? -v5 : v5);
    }
    double z = (double)y * ptd_log2d((double)ptd_f32_from_bits(ax));
    double v = ptd_exp2d(z);
    if (xneg && yint == 1) v = -v;
    return (float)v;
}

/* ---- sqrt (f64, via f32 seed + Newton; avoids depending on a correctly rounded f64 sqrt) --- */
float ptref_sqrtf_ieee(float x); /* correctly rounded f32 sqrt, provided by ptref.c (sqrtf) */

static inline double ptd_sqrtd(double a) {
    /* a in [0, 4]: seed from the correctly rounded f32 sqrt, 3 Newton steps in binary64. */
    if (!(a > 0.0)) return 0.0;
    double y = (double)ptref_sqrtf_ieee((float)a);
    if (!(y > 0.0)) return 0.0;
    y = 0.5 * (y + a / y);
    y = 0.5 * (y + a / y);
    y = 0.5 * (y + a / y);
    return y;
}

/* ---- asin / acos ------------------------------------------------------------------------ */

/* asin(x) = x + x*R(x^2) on |x| <= 0.5 (fdlibm rational approximation). */
static inline double ptd_kasin_R(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}

static inline float ptd_acosf(float x) {
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    double xd = (double)x;
    if (xd != xd) return x;
    if (xd > 1.0 || xd < -1.0) return ptd_f32_from_bits(0x7fc00000u);
    if (xd >= -0.5 && xd <= 0.5) {
        double a = xd + xd * ptd_kasin_R(xd * xd);
        return (float)(PIO2 - a);
    }
    if (xd > 0.5) {
        double z = (1.0 - xd) * 0.5;
        double s = ptd_sqrtd(z);
        double a = s + s * ptd_kasin_R(z);
        return (float)(2.0 * a);
    } else {
        double z = (1.0 + xd) * 0.5;
        double s = ptd_sqrtd(z);
        double a = s + s * ptd_kasin_R(z);
        return (float)(PI - 2.0 * a);
    }
}

/* ---- atan / atan2 ----------------------------------------------------------------------- */

/* atan(t) for 0 <= t <= 1: argument halving twice (t -> t/(1+sqrt(1+t^2))) then odd Taylor series. */
static inline double ptd_katan01(double t) {
    /* reduce: atan(t) = 2 atan(t / (1 + sqrt(1 + t^2))) ; after two reductions t <= tan(pi/16) = 0.1989 */
    double t1 = t / (1.0 + ptd_sqrtd(1.0 + t * t));
    double t2 = t1 / (1.0 + ptd_sqrtd(1.0 + t1 * t1));
    double z = t2 * t2; /* <= 0.0396 */
    /* series to z^12/25: 0.0396^13/27 ~ 2e-20 */
    double p = 1.0 / 25.0;
    p = -p * z + 1.0 / 23.0;
    p = -p * z + 1.0 / 21.0;
    p = -p * z + 1.0 / 19.0;
    p = -p * z + 1.0 / 17.0;
    p = -p * z + 1.0 / 15.0;
    p = -p * z + 1.0 / 13.0;
    p = -p * z + 1.0 / 11.0;
    p = -p * z + 1.0 / 9.0;
    p = -p * z + 1.0 / 7.0;
    p = -p * z + 1.0 / 5.0;
    p = -p * z + 1.0 / 3.0;
    p = -p * z + 1.0;
    return 4.0 * (t2 * p);
}

static inline float ptd_atan2f(float y, float x) {
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    double yd = (double)y, xd = (double)x;
    if (yd != yd || xd != xd) return x + y;
    int yneg = (ptd_f32_bits(y) >> 31) != 0, xneg = (ptd_f32_bits(x) >> 31) != 0;
    double ay = yneg ? -yd : yd, ax = xneg ? -xd : xd;
    double r;
    int ainf = !(ax < 1.0e300), binf = !(ay < 1.0e300);
    if (ay == 0.0) r = 0.0;                                  /* includes (0,0): atan2(+-0, +x)=+-0, (+-0,-x)=+-pi */
    else if (ax == 0.0) r = PIO2;
    else if (ainf && binf) r = 0.5 * PIO2;
    else if (binf) r = PIO2;
    else if (ainf) r = 0.0;
    else if (ay <= ax) r = ptd_katan01(ay / ax);
    else r = PIO2 - ptd_katan01(ax / ay);
    if (xneg) r = PI - r;
    if (yneg) r = -r;
    return (float)r;
}

#endif /* PTREF_DETMATH_H */
