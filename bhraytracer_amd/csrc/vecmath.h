// vecmath.h — float vector / matrix / colour operations in EXACTLY the operation order of the
// reference's cy* headers, usable from host C++ and from HIP device code.
//
// Bit-exact parity with the reference needs the same IEEE-754 single operations in the same
// order: every TU that includes this header is compiled with
// -ffp-contract=off (no FMA fusion), division is a true division, sqrt is correctly rounded.
// Citations: DataStructure/cyVector.h:271-386 (Vec3), cyMatrix.h:399-406,522-533,671-689,799-819
// (Matrix3, column-major), cyColor.h (Color), cyCore.h:187-200 (Min/Max/Clamp/Sqrt).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BHRT_FN __host__ __device__ inline
#else
#define BHRT_FN inline
#endif

namespace bhrt {

struct V3 {
    float x, y, z;
};

BHRT_FN V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
BHRT_FN V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
BHRT_FN V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
BHRT_FN V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
BHRT_FN V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
BHRT_FN V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); } // cyVector.h:278: v*p == p*v
BHRT_FN V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
BHRT_FN V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
BHRT_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                                  // cyVector.h:381
BHRT_FN V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // cyVector.h:379
BHRT_FN float length_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                                   // cyVector.h:308
BHRT_FN float length(V3 a) { return sqrtf(length_sq(a)); }                                                     // cyVector.h:309 (SSE sqrtss)
BHRT_FN V3 normalized(V3 a) { return a / length(a); }                                                          // cyVector.h:307
BHRT_FN bool is_zero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
BHRT_FN float fmin_cy(float a, float b) { return a <= b ? a : b; } // cyCore.h:188
BHRT_FN float fmax_cy(float a, float b) { return a >= b ? a : b; } // cyCore.h:187

// Matrix3 * Vec3 (cyMatrix.h:682-687), m column-major
BHRT_FN V3 mat_mul(const float *m, V3 p)
{
    return v3(p.x * m[0] + p.y * m[3] + p.z * m[6], p.x * m[1] + p.y * m[4] + p.z * m[7], p.x * m[2] + p.y * m[5] + p.z * m[8]);
}
// Transformation::TransposeMult (scene.h:238-245): d.k = column(k) . dir
BHRT_FN V3 mat_tmul(const float *m, V3 d)
{
    return v3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[3] * d.x + m[4] * d.y + m[5] * d.z, m[6] * d.x + m[7] * d.y + m[8] * d.z);
}

} // namespace bhrt
