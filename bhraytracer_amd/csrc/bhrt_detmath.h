// bhrt_detmath.h — deterministic elementary functions for the shading path.
//
// The reference calls libm (sinf, cosf, tanf, acosf, asinf, atan2f, powf) inside its samplers
// and BRDF (Materials/Blinn/MtlBlinn.cpp:107-109,175,325,529,602-716; Objects/Sphere/Sphere.cpp:62-63;
// Objects/Plane/Plane.cpp:57-58; Scenes/scene.h:326-328,416; Main.cpp:223-225).  glibc's libm and the
// GPU's ocml do not round identically, and one differing ulp in a sampled direction changes which
// object a secondary ray hits.  To keep the HIP path and the CPU oracle bit-comparable, both
// evaluate these functions with THIS header in "device math" mode: every function is built from
// IEEE-754 double + - * / and exact bit manipulation only, so host (x86-64 SSE2) and device
// (gfx950) produce identical bits when compiled with -ffp-contract=off.  Results are within
// about 1 ulp (float) of the correctly rounded value, i.e. as close to libm as libm is to itself
// across platforms; the oracle's libm mode (pinned against the compiled reference) and its
// device-math mode are compared statistically in tests/.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BHRT_DM __host__ __device__ inline
#else
#define BHRT_DM inline
#endif

namespace bhrt {
namespace dm {

BHRT_DM uint64_t dbits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
BHRT_DM double bitsd(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
BHRT_DM uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
BHRT_DM float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
BHRT_DM bool isnan_d(double x) { return x != x; }
BHRT_DM double fabs_d(double x) { return bitsd(dbits(x) & 0x7fffffffffffffffULL); }
BHRT_DM double quiet_nan() { return bitsd(0x7ff8000000000000ULL); }
BHRT_DM double inf_d() { return bitsd(0x7ff0000000000000ULL); }

// round to nearest integer for |x| < 2^51 (add-magic trick; exact, no libm)
BHRT_DM double rint_small(double x)
{
    const double magic = 6755399441055744.0; // 1.5 * 2^52
    return (x + magic) - magic;
}
// 2^k for -1022 <= k <= 1023
BHRT_DM double pow2i(int k) { return bitsd((uint64_t)(k + 1023) << 52); }

// sin and cos of a double argument of moderate size (|x| < ~1e5): Cody-Waite reduction by pi/2
BHRT_DM void sincos_d(double x, double *s, double *c)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00; // 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11; // pi/2 - pio2_hi
    double kd = rint_small(x * two_over_pi);
    double r = (x - kd * pio2_hi) - kd * pio2_lo;
    long long k = (long long)kd;
    double r2 = r * r;
    // Taylor polynomials on |r| <= pi/4 (error < 1e-16 relative to 1)
    double ps = r2 * (-1.0 / 6 + r2 * (1.0 / 120 + r2 * (-1.0 / 5040 + r2 * (1.0 / 362880 + r2 * (-1.0 / 39916800 + r2 * (1.0 / 6227020800.0 + r2 * (-1.0 / 1307674368000.0)))))));
    double sn = r + r * ps;
    double cs = 1.0 + r2 * (-0.5 + r2 * (1.0 / 24 + r2 * (-1.0 / 720 + r2 * (1.0 / 40320 + r2 * (-1.0 / 3628800 + r2 * (1.0 / 479001600.0 + r2 * (-1.0 / 87178291200.0)))))));
    switch ((int)(k & 3)) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}

BHRT_DM float sinf_(float x)
{
    if (!(x == x) || x - x != 0.f) return bitsf(0x7fc00000u);
    double s, c;
    sincos_d((double)x, &s, &c);
    return (float)s;
}
BHRT_DM float cosf_(float x)
{
    if (!(x == x) || x - x != 0.f) return bitsf(0x7fc00000u);
    double s, c;
    sincos_d((double)x, &s, &c);
    return (float)c;
}
BHRT_DM float tanf_(float x)
{
    if (!(x == x) || x - x != 0.f) return bitsf(0x7fc00000u);
    double s, c;
    sincos_d((double)x, &s, &c);
    return (float)(s / c);
}

BHRT_DM double sqrt_d(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __dsqrt_rn(x);
#else
    return __builtin_sqrt(x);
#endif
}

// atan on [0, +inf) in double
BHRT_DM double atan_pos_d(double t)
{
    const double pio2 = 1.57079632679489661923, pio4 = 0.78539816339744830962;
    double base = 0.0;
    bool inv = false;
    if (t > 1.0) { t = 1.0 / t; inv = true; }
    if (t > 0.41421356237309504880) { t = (t - 1.0) / (t + 1.0); base = pio4; }
    double z = t * t;
    // odd series, |t| <= 0.4142 -> z <= 0.1716; 16 terms -> < 1e-13
    double p = 1.0 / 31;
    p = 1.0 / 29 - z * p; p = 1.0 / 27 - z * p; p = 1.0 / 25 - z * p; p = 1.0 / 23 - z * p;
    p = 1.0 / 21 - z * p; p = 1.0 / 19 - z * p; p = 1.0 / 17 - z * p; p = 1.0 / 15 - z * p;
    p = 1.0 / 13 - z * p; p = 1.0 / 11 - z * p; p = 1.0 / 9 - z * p; p = 1.0 / 7 - z * p;
    p = 1.0 / 5 - z * p; p = 1.0 / 3 - z * p; p = 1.0 - z * p;
    double a = base + t * p;
    return inv ? pio2 - a : a;
}

BHRT_DM float atan2f_(float yf, float xf)
{
    const double pi = 3.14159265358979323846;
    double y = yf, x = xf;
    if (isnan_d(x) || isnan_d(y)) return bitsf(0x7fc00000u);
    bool yneg = (fbits(yf) >> 31) != 0, xneg = (fbits(xf) >> 31) != 0;
    double r;
    if (y == 0.0) r = xneg ? pi : 0.0;
    else if (x == 0.0) r = pi / 2;
    else {
        double ay = fabs_d(y), ax = fabs_d(x);
        double a;
        if (ay == inf_d() && ax == inf_d()) a = pi / 4;
        else if (ay == inf_d()) a = pi / 2;
        else if (ax == inf_d()) a = 0.0;
        else a = atan_pos_d(ay / ax);
        r = xneg ? pi - a : a;
    }
    float rf = (float)r;
    return yneg ? -rf : rf;
}

// asin on [0,1] in double
BHRT_DM double asin_pos_d(double x)
{
    const double pio2 = 1.57079632679489661923;
    bool big = x > 0.5;
    double t = x;
    if (big) t = sqrt_d((1.0 - x) * 0.5);
    double z = t * t;
    // asin t = t * sum c_n z^n, c_n = (2n)! / (4^n (n!)^2 (2n+1)) as correctly rounded literals;
    // z <= 0.25 and 15 terms leave < 1e-11 relative, Horner form, no run-time division
    double sum = 0.005153309682319905;
    sum = 0.005740037670841924 + z * sum;
    sum = 0.006447210311889649 + z * sum;
    sum = 0.0073125258735988454 + z * sum;
    sum = 0.008390335809616815 + z * sum;
    sum = 0.009761609529194078 + z * sum;
    sum = 0.011551800896139705 + z * sum;
    sum = 0.01396484375 + z * sum;
    sum = 0.017352764423076924 + z * sum;
    sum = 0.022372159090909092 + z * sum;
    sum = 0.030381944444444444 + z * sum;
    sum = 0.044642857142857144 + z * sum;
    sum = 0.075 + z * sum;
    sum = 0.16666666666666666 + z * sum;
    sum = 1.0 + z * sum;
    double a = t * sum;
    return big ? pio2 - 2.0 * a : a;
}
BHRT_DM float asinf_(float xf)
{
    double x = xf;
    if (isnan_d(x) || fabs_d(x) > 1.0) return bitsf(0x7fc00000u);
    double a = asin_pos_d(fabs_d(x));
    return (float)(x < 0 ? -a : a);
}
BHRT_DM float acosf_(float xf)
{
    const double pi = 3.14159265358979323846, pio2 = 1.57079632679489661923;
    double x = xf;
    if (isnan_d(x) || fabs_d(x) > 1.0) return bitsf(0x7fc00000u);
    double r;
    if (x > 0.5) r = 2.0 * asin_pos_d(sqrt_d((1.0 - x) * 0.5));
    else if (x < -0.5) r = pi - 2.0 * asin_pos_d(sqrt_d((1.0 + x) * 0.5));
    else r = pio2 - (x < 0 ? -asin_pos_d(-x) : asin_pos_d(x));
    return (float)r;
}

// natural log of a positive finite double
BHRT_DM double log_pos_d(double x)
{
    const double ln2 = 0.69314718055994530942;
    uint64_t u = dbits(x);
    int e = (int)((u >> 52) & 0x7ff);
    if (e == 0) { // subnormal: scale up
        x = x * 18014398509481984.0; // 2^54
        u = dbits(x);
        e = (int)((u >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    double m = bitsd((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); // [1,2)
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0); // |s| <= 0.1716
    double z = s * s;
    double p = 1.0 / 23;
    p = 1.0 / 21 + z * p; p = 1.0 / 19 + z * p; p = 1.0 / 17 + z * p; p = 1.0 / 15 + z * p;
    p = 1.0 / 13 + z * p; p = 1.0 / 11 + z * p; p = 1.0 / 9 + z * p; p = 1.0 / 7 + z * p;
    p = 1.0 / 5 + z * p; p = 1.0 / 3 + z * p; p = 1.0 + z * p;
    return (double)e * ln2 + 2.0 * s * p;
}

// e^z in double, result clamped to [0, +inf]
BHRT_DM double exp_d(double z)
{
    if (isnan_d(z)) return z;
    if (z > 709.0) return inf_d();
    if (z < -745.0) return 0.0;
    const double inv_ln2 = 1.44269504088896340736;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    double kd = rint_small(z * inv_ln2);
    double r = (z - kd * ln2_hi) - kd * ln2_lo; // |r| <= 0.3466
    double p = 1.0 / 6227020800.0;              // 1/13!
    p = 1.0 / 479001600.0 + r * p; p = 1.0 / 39916800.0 + r * p; p = 1.0 / 3628800.0 + r * p;
    p = 1.0 / 362880.0 + r * p; p = 1.0 / 40320.0 + r * p; p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p; p = 1.0 / 120.0 + r * p; p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p; p = 0.5 + r * p; p = 1.0 + r * p; p = 1.0 + r * p;
    int k = (int)kd;
    // split the scaling so that 2^k never leaves the normal range
    int k1 = k / 2, k2 = k - k1;
    return p * pow2i(k1) * pow2i(k2);
}

// powf with libm's special cases for the arguments this path produces
BHRT_DM float powf_(float xf, float yf)
{
    double x = xf, y = yf;
    if (y == 0.0) return 1.0f;
    if (x == 1.0) return 1.0f;
    if (isnan_d(x) || isnan_d(y)) return bitsf(0x7fc00000u);
    bool y_is_int = false, y_odd = false;
    {
        double ay = fabs_d(y);
        if (ay >= 9007199254740992.0) { y_is_int = true; }
        else if (ay >= 1.0 || ay == 0.0) {
            double fl = rint_small(ay);
            if (ay < 4503599627370496.0 && fl == ay) { y_is_int = true; y_odd = ((long long)fl & 1) != 0; }
            else if (ay >= 4503599627370496.0) { y_is_int = true; y_odd = ((long long)ay & 1) != 0; }
        }
    }
    if (x == 0.0) {
        bool neg = (fbits(xf) >> 31) != 0 && y_odd;
        if (y > 0) return neg ? -0.0f : 0.0f;
        return neg ? -bitsf(0x7f800000u) : bitsf(0x7f800000u);
    }
    double ax = fabs_d(x);
    double sign = 1.0;
    if (x < 0) {
        if (!y_is_int) return bitsf(0x7fc00000u);
        if (y_odd) sign = -1.0;
    }
    double r;
    if (ax == inf_d()) r = y > 0 ? inf_d() : 0.0;
    else if (fabs_d(y) == inf_d()) r = ((ax > 1.0) == (y > 0)) ? inf_d() : (ax == 1.0 ? 1.0 : 0.0);
    else r = exp_d(y * log_pos_d(ax));
    return (float)(sign * r);
}

} // namespace dm
} // namespace bhrt
