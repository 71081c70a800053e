// bhrt_detmath.h — deterministic elementary functions for the shading path.
//
// The reference calls libm (sinf, cosf, tanf, acosf, asinf, atan2f, powf) inside its samplers
// and BRDF (Materials/Blinn/MtlBlinn.cpp:107-109,175,325,529,602-716; Objects/Sphere/Sphere.cpp:62-63;
// Objects/Plane/Plane.cpp:57-58; Scenes/scene.h:326-328,416; Main.cpp:223-225).  glibc's libm and the
// GPU's ocml do not round identically, and one differing ulp in a sampled direction changes which
// object a secondary ray hits.  To keep the HIP path and the CPU oracle bit-comparable, both
// evaluate these functions with THIS header in "device math" mode: every function is built from
// IEEE-754 float/double + - * / sqrt and exact bit manipulation only, so host (x86-64 SSE2) and device
// (gfx950) produce identical bits when compiled with -ffp-contract=off.  Results are within
// 1-3 ulp (float) of the correctly rounded value (trigonometry in single precision, pow through a
// double-precision log/exp), i.e. the accuracy class of libm's own float functions; the oracle's libm mode (pinned against the compiled reference) and its
// device-math mode are compared statistically in tests/.
#pragma once
#include <math.h>
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

// ---------------------------------------------------------------------------------------------------------
// Trigonometric functions: single precision only (full-rate VALU on gfx950), Cody-Waite reduction + the classic
// Cephes single-precision minimax polynomials, every operation a plain IEEE float op in a fixed order.
// Accuracy about 1-2 ulp — the same class as libm's own float functions across platforms.
// ---------------------------------------------------------------------------------------------------------
// correctly rounded square roots: sqrt()/sqrtf() are IEEE on both sides (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt); HIP's __fsqrt_rn is NOT (it lowers to the native approximation)
BHRT_DM double sqrt_d(double x) { return sqrt(x); }
BHRT_DM float sqrt_f(float x) { return sqrtf(x); }
BHRT_DM float fabs_f(float x) { return bitsf(fbits(x) & 0x7fffffffu); }

// sin and cos of |x| < 2^20 or so; anything larger or non-finite gives NaN (never produced on this path)
BHRT_DM void sincos_f(float x, float *s, float *c)
{
    if (!(fabs_f(x) < 1.0e6f)) { *s = *c = bitsf(0x7fc00000u); return; }
    const float magic = 12582912.0f;                        // 1.5 * 2^23: round-to-nearest-integer by add/subtract
    const float kf = (x * 0.636619772367581343f + magic) - magic;
    const int k = (int)kf;
    // pi/2 in three pieces (2 x the Cephes DP1..DP3): k * P1 is exact for the k that occur here
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    const float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r;
    ps = ps + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    switch (k & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
    }
}
BHRT_DM float sinf_(float x) { float s, c; sincos_f(x, &s, &c); return s; }
BHRT_DM float cosf_(float x) { float s, c; sincos_f(x, &s, &c); return c; }
BHRT_DM float tanf_(float x) { float s, c; sincos_f(x, &s, &c); return s / c; }

// asin on [0, 0.5]: x + x * z * P(z), z = x*x (Cephes asinf)
BHRT_DM float asin_core_f(float x, float z)
{
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x;
    return p + x;
}
BHRT_DM float asinf_(float x)
{
    const float a = fabs_f(x);
    if (!(a <= 1.0f)) return bitsf(0x7fc00000u);
    float r;
    if (a > 0.5f) {
        const float z = 0.5f * (1.0f - a);
        const float t = asin_core_f(sqrt_f(z), z);
        r = 1.5707963267948966f - (t + t);
    } else
        r = asin_core_f(a, a * a);
    return (fbits(x) >> 31) ? -r : r;
}
BHRT_DM float acosf_(float x)
{
    if (!(fabs_f(x) <= 1.0f)) return bitsf(0x7fc00000u);
    if (x > 0.5f) {
        const float z = 0.5f * (1.0f - x);
        const float t = asin_core_f(sqrt_f(z), z);
        return t + t;
    }
    if (x < -0.5f) {
        const float z = 0.5f * (1.0f + x);
        const float t = asin_core_f(sqrt_f(z), z);
        return 3.14159265358979323846f - (t + t);
    }
    const float a = fabs_f(x);
    const float t = asin_core_f(a, a * a);
    return 1.5707963267948966f - ((fbits(x) >> 31) ? -t : t);
}
// atan on [0, +inf) (Cephes atanf)
BHRT_DM float atan_pos_f(float x)
{
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    const float z = x * x;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return y + p;
}
BHRT_DM float atan2f_(float y, float x)
{
    if (x != x || y != y) return bitsf(0x7fc00000u);
    const bool yneg = (fbits(y) >> 31) != 0, xneg = (fbits(x) >> 31) != 0;
    const float ay = fabs_f(y), ax = fabs_f(x);
    const float inf = bitsf(0x7f800000u);
    float r;
    if (ay == 0.0f) r = xneg ? 3.14159265358979323846f : 0.0f;
    else if (ax == 0.0f) r = 1.5707963267948966f;
    else {
        float a;
        if (ay == inf && ax == inf) a = 0.7853981633974483f;
        else if (ay == inf) a = 1.5707963267948966f;
        else if (ax == inf) a = 0.0f;
        else a = atan_pos_f(ay / ax);
        r = xneg ? 3.14159265358979323846f - a : a;
    }
    return yneg ? -r : r;
}

// (double)r / RAND_MAX of the reference's Rnd01 (MtlBlinn.cpp:43) as one multiplication: the device-math form of
// the conversion (the sequential/libm oracle mode keeps the division, which is what the compiled reference does)
BHRT_DM float rand_to_unit(int r) { return (float)((double)r * 4.656612875245796924105750827168e-10); }

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
    // s = (m-1)/(m+1), |s| <= 0.1716: reciprocal seeded by ONE float division and refined in double (error ~2^-46)
    const double den = m + 1.0;
    const double r0 = (double)(1.0f / (float)den);
    const double r1 = r0 * (2.0 - den * r0);
    double s = (m - 1.0) * r1;
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
    // The case this path produces almost always — positive finite base, finite exponent — first and behind ONE branch
    // (the special cases below are a dozen branch points that every call used to walk through).  x == 1 and y == 0 need
    // no special case here: log_pos_d(1) is exactly 0 and exp_d(+-0) exactly 1.
    if (x > 0.0 && x < inf_d() && fabs_d(y) < inf_d()) return (float)exp_d(y * log_pos_d(x));
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
