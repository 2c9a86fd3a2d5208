// pt_math.h -- scalar / vec3 / RNG / deterministic-transcendental layer of the gfx950 path tracer.
//
// Everything here is a fixed sequence of IEEE-754 operations: no fused multiply-add, no reassociation
// (build with -ffp-contract=off, never fast-math), correctly rounded f32 divide and sqrt.  That is what
// makes first-hit ids and quantised radiance reproducible bit for bit on any IEEE machine.
//
// Vector semantics stand in for the `hmath` crate the reference uses (reference src/common.rs:3,7-10):
// dot = (x*x' + y*y') + z*z', length = sqrt(dot(v,v)), normalize = v / length (three divides),
// right-handed cross, component-wise * and /.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_HD __host__ __device__ __forceinline__

namespace pt {

struct V3 { float x, y, z; };

PT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PT_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// Correctly rounded square root.  On the device the compiler's expansion of it is 16 instructions, five of which scale
// arguments below 2^-96 up and the result back down (the correction step's residual x - s'*s would lose bits there) and two
// pass zeros and infinities through.  A wave none of whose lanes holds such an argument takes the same expansion without
// those seven: hardware estimate s (1 ulp), then s - 1ulp when x - (s - 1ulp)*s <= 0, s + 1ulp when x - (s + 1ulp)*s > 0, both
// residuals by fused multiply-add from the estimate - the same operations on the same operands, hence the same bits; zeros,
// infinities, NaNs and negative arguments come out of the estimate right and fall through both corrections (every
// comparison with their NaN residuals is false; sqrt(+-0): the neighbours' residuals are 0 and NaN).
PT_HD float sqrt_ieee(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_FAST_SQRT)
    // (one comparison: it also sends waves with a zero or a negative argument the long way, which costs less than a second
    // instruction at every square root: measured on the sample scene +1.3 % against +0.9 %)
    if (__builtin_amdgcn_ballot_w64(x < 0x1p-96f) == 0ull) {
        const float est = __builtin_amdgcn_sqrtf(x);
        const float below = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, est) - 1u);
        const float above = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, est) + 1u);
        const float r_below = __builtin_fmaf(-below, est, x), r_above = __builtin_fmaf(-above, est, x);
        float s = r_below <= 0.0f ? below : est;
        s = r_above > 0.0f ? above : s;
        return s;
    }
#endif
    return __builtin_sqrtf(x);
}
PT_HD float length(V3 a) { return sqrt_ieee(dot(a, a)); }
// normalize = three correctly rounded divisions by the same length.  On the device the compiler expands each into eleven
// instructions: both operands scaled when their exponents call for it (v_div_scale), the hardware reciprocal of the
// denominator refined once, the quotient refined twice, the last step scaled back (v_div_fmas), special cases patched
// (v_div_fixup).  When NO scaling applies - every component at least 2^-60 in magnitude (no square underflows, so the
// length is at least about 2^-60 too), the length at most 2^20: the reciprocal is normal, every quotient is normal, no
// residual underflows - v_div_scale returns its operands unchanged, v_div_fmas is a plain fused multiply-add and
// v_div_fixup returns its argument, so the three divisions differ only in their numerators: the reciprocal and its
// refinement are computed ONCE and each component takes the remaining five operations, the same operations on the same
// operands as the compiler's expansion, hence the same bits (18 instructions and 3 for the test instead of 33).  A wave
// with a lane outside those conditions (a zero or vanishing component: a ray along an axis; a length beyond 2^20; NaN,
// which fails the comparison of the length; infinity) takes the divisions as written.
PT_HD V3 normalize(V3 a) {
    const float len = length(a);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_NO_SHARED_RCP)
    const float least = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const bool plain = (least >= 0x1p-60f) & (len <= 0x1p20f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!plain) == 0ull, 1)) {
        float r = __builtin_amdgcn_rcpf(len);
        r = __builtin_fmaf(__builtin_fmaf(-len, r, 1.0f), r, r);
        V3 q = v3(a.x * r, a.y * r, a.z * r);
        for (int round = 0; round < 2; ++round)
            q = v3(__builtin_fmaf(__builtin_fmaf(-len, q.x, a.x), r, q.x), __builtin_fmaf(__builtin_fmaf(-len, q.y, a.y), r, q.y),
                   __builtin_fmaf(__builtin_fmaf(-len, q.z, a.z), r, q.z));
        return q;
    }
#endif
    return a / len;
}

// Rust f32::max / f32::min: the non-NaN operand wins; for equal operands the second is returned.
PT_HD float max_rs(float a, float b) { if (b != b) return a; if (a != a) return b; return a > b ? a : b; }
PT_HD float min_rs(float a, float b) { if (b != b) return a; if (a != a) return b; return a < b ? a : b; }

// Rust `x as u32` / `x as u8` from f32: saturating, NaN -> 0.
PT_HD uint32_t f32_as_u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
PT_HD uint32_t f32_as_u8(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 255.0f) return 255u;
    return (uint32_t)x;
}

constexpr float PI_F = 3.14159274101257324219f;  // std::f32::consts::PI (reference src/common.rs:5)

// ---- bit casts ------------------------------------------------------------------------------------
PT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
PT_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
PT_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// ---- counter RNG ----------------------------------------------------------------------------------
// Replaces the reference's global xorshift32 (src/common.rs:28-36), which 8 threads race on.  A draw is
// a pure function of (seed, pixel, sample, path node, dimension); like xorshift32 it never returns 0,
// so random32 stays in (0, 1].
PT_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
PT_HD uint32_t rng_sample_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t h = mix32(seed + 0x9E3779B9U);
    h = mix32((h ^ pixel) + 0x85EBCA6BU);
    h = mix32((h ^ sample) + 0xC2B2AE35U);
    return h;
}
PT_HD uint32_t rng_node_key(uint32_t sample_key, uint32_t node) { return mix32((sample_key ^ node) + 0x27D4EB2FU); }
PT_HD uint32_t rng_draw(uint32_t node_key, uint32_t dim) {
    uint32_t h = mix32((node_key ^ dim) + 0x165667B1U);
    return h ? h : 1u;
}
// reference src/common.rs:38-41: r as f32 / u32::MAX as f32 (the divisor rounds to 2^32)
PT_HD float random32(uint32_t r) { return (float)r / 4294967296.0f; }

// ---- deterministic transcendental functions ----------------------------------------------------------
// The reference calls f32::sin/cos/powf/exp/acos/atan2, i.e. whatever libm the platform has.  These
// replacements evaluate in binary64 with plain + - * / only and round once to binary32, so device and
// host agree bit for bit and the result is the correctly rounded f32 value except within ~1e-8 ulp of
// a rounding boundary.  sin/cos/asin coefficients and the pi/2, ln2 splittings are fdlibm's.
namespace det {

PT_HD double floor_d(double x) { return __builtin_floor(x); }  // exact, so identical on every IEEE machine
PT_HD double pow2i(int k) { return u2d((uint64_t)(k + 1023) << 52); }

PT_HD int rem_pio2(double x, double &r) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    double q = floor_d(x * TWO_OVER_PI + 0.5);
    double t = x - q * PIO2_1;
    r = t - q * PIO2_1T;
    return (int)(q - 4.0 * floor_d(q * 0.25));  // q mod 4, exact in binary64
}
PT_HD double ksin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    return r + (r * z) * p;
}
PT_HD double kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    return (1.0 - 0.5 * z) + (z * z) * p;
}
PT_HD float sinf_(float x) {
    double xd = (double)x, r;
    if (!(xd > -1.0e15 && xd < 1.0e15)) return x - x;
    int q = rem_pio2(xd, r);
    double v = (q & 1) ? kcos(r) : ksin(r);
    if (q & 2) v = -v;
    return (float)v;
}
PT_HD float cosf_(float x) {
    double xd = (double)x, r;
    if (!(xd > -1.0e15 && xd < 1.0e15)) return x - x;
    int q = rem_pio2(xd, r);
    double v = (q & 1) ? ksin(r) : kcos(r);
    if (((q + 1) & 2) != 0) v = -v;
    return (float)v;
}
// sin and cos of the same argument share the range reduction.
PT_HD void sincosf_(float x, float &s, float &c) {
    double xd = (double)x, r;
    if (!(xd > -1.0e15 && xd < 1.0e15)) { s = x - x; c = x - x; return; }
    int q = rem_pio2(xd, r);
    double sv = ksin(r), cv = kcos(r);
    double so = (q & 1) ? cv : sv;
    double co = (q & 1) ? sv : cv;
    if (q & 2) so = -so;
    if (((q + 1) & 2) != 0) co = -co;
    s = (float)so; c = (float)co;
}

PT_HD double kexp(double r) {  // exp(r), |r| <= 0.35: Taylor to r^14/14!
    double p = 1.0 / 87178291200.0;
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
PT_HD double exp2d(double z) {
    const double LN2 = 6.93147180559945286227e-01;
    if (z > 200.0) z = 200.0;
    if (z < -200.0) z = -200.0;
    double k = floor_d(z + 0.5);
    double f = z - k;
    return kexp(f * LN2) * pow2i((int)k);
}
PT_HD float expf_(float x) {
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double xd = (double)x;
    if (xd != xd) return x;
    if (xd > 100.0) xd = 100.0;
    if (xd < -120.0) xd = -120.0;
    double k = floor_d(xd * LOG2E + 0.5);
    double r = (xd - k * LN2_HI) - k * LN2_LO;
    return (float)(kexp(r) * pow2i((int)k));
}
PT_HD double log2d(double a) {  // finite a > 0, normal double
    const double LOG2E = 1.44269504088896338700e+00;
    uint64_t b = d2u(a);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = u2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
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
PT_HD float powf_(float x, float y) {
    uint32_t xb = f2u(x), yb = f2u(y);
    uint32_t ax = xb & 0x7fffffffu, ay = yb & 0x7fffffffu;
    if (ay == 0) return 1.0f;
    if (xb == 0x3f800000u) return 1.0f;
    if (ax > 0x7f800000u || ay > 0x7f800000u) return x + y;
    int yint = 0;  // 0 not an integer, 1 odd, 2 even
    if (ay >= 0x4b800000u) yint = 2;
    else if (ay >= 0x3f800000u) {
        int sh = 150 - (int)(ay >> 23);
        uint32_t mant = (ay & 0x007fffffu) | 0x00800000u;
        if (sh <= 0) yint = 2;
        else if ((mant & ((1u << sh) - 1u)) == 0) yint = ((mant >> sh) & 1u) ? 1 : 2;
    }
    bool yneg = (yb >> 31) != 0, xneg = (xb >> 31) != 0;
    if (ay == 0x7f800000u) {
        if (ax == 0x3f800000u) return 1.0f;
        bool big = ax > 0x3f800000u;
        return (big != yneg) ? u2f(0x7f800000u) : 0.0f;
    }
    if (ax == 0 || ax == 0x7f800000u) {
        bool to_inf = (ax == 0) ? yneg : !yneg;
        float mag = to_inf ? u2f(0x7f800000u) : 0.0f;
        return (xneg && yint == 1) ? -mag : mag;
    }
    if (xneg && yint == 0) return u2f(0x7fc00000u);
    if (yb == 0x40a00000u) {  // y == 5 (Schlick's Fresnel term): x^5 by three binary64 products; x*x is exact
        double a = (double)u2f(ax);
        double a2 = a * a;
        double v5 = (a2 * a2) * a;
        return (float)(xneg ? -v5 : v5);
    }
    double z = (double)y * log2d((double)u2f(ax));
    double v = exp2d(z);
    if (xneg && yint == 1) v = -v;
    return (float)v;
}

PT_HD double sqrtd(double a) {  // a in [0, 4]: correctly rounded f32 sqrt as seed + 3 Newton steps
    if (!(a > 0.0)) return 0.0;
    double y = (double)sqrt_ieee((float)a);
    if (!(y > 0.0)) return 0.0;
    y = 0.5 * (y + a / y);
    y = 0.5 * (y + a / y);
    y = 0.5 * (y + a / y);
    return y;
}
PT_HD double kasin_R(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}
PT_HD float acosf_(float x) {
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    double xd = (double)x;
    if (xd != xd) return x;
    if (xd > 1.0 || xd < -1.0) return u2f(0x7fc00000u);
    if (xd >= -0.5 && xd <= 0.5) {
        double a = xd + xd * kasin_R(xd * xd);
        return (float)(PIO2 - a);
    }
    if (xd > 0.5) {
        double z = (1.0 - xd) * 0.5;
        double s = sqrtd(z);
        double a = s + s * kasin_R(z);
        return (float)(2.0 * a);
    }
    double z = (1.0 + xd) * 0.5;
    double s = sqrtd(z);
    double a = s + s * kasin_R(z);
    return (float)(PI - 2.0 * a);
}
PT_HD double katan01(double t) {  // atan(t), 0 <= t <= 1
    double t1 = t / (1.0 + sqrtd(1.0 + t * t));
    double t2 = t1 / (1.0 + sqrtd(1.0 + t1 * t1));
    double z = t2 * t2;
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
PT_HD float atan2f_(float y, float x) {
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    double yd = (double)y, xd = (double)x;
    if (yd != yd || xd != xd) return x + y;
    bool yneg = (f2u(y) >> 31) != 0, xneg = (f2u(x) >> 31) != 0;
    double ay = yneg ? -yd : yd, ax = xneg ? -xd : xd;
    double r;
    bool ainf = !(ax < 1.0e300), binf = !(ay < 1.0e300);
    if (ay == 0.0) r = 0.0;
    else if (ax == 0.0) r = PIO2;
    else if (ainf && binf) r = 0.5 * PIO2;
    else if (binf) r = PIO2;
    else if (ainf) r = 0.0;
    else if (ay <= ax) r = katan01(ay / ax);
    else r = PIO2 - katan01(ax / ay);
    if (xneg) r = PI - r;
    if (yneg) r = -r;
    return (float)r;
}

}  // namespace det
}  // namespace pt
