// bhrt_oracle.cpp — CPU restatement of the reference's per-pixel render path.
//
// TEST INFRASTRUCTURE ONLY (see bhrt_oracle.h).  Structured like the reference (recursive
// functions, AoS HitInfo) on purpose: it is the checker for the differently-structured HIP
// wavefront path, and is itself pinned bit-for-bit against the compiled reference
// (oracle/_ref) in sequential-RNG + libm mode.
//
// Every function cites the reference code it follows (paths relative to
// /root/reference/BHRayTracer).  Floating point: compiled with -O2 -ffp-contract=off on
// x86-64 (SSE2 scalar = strict IEEE single/double), same operation order as the reference,
// including its implicit float->double promotions.
#include "bhrt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <type_traits>
#include <string>
#include <vector>

#include "bhrt_detmath.h"
#include "bhrt_flat.h"
#include "bhrt_rng.h"

namespace {

std::string g_err;

#define O_PI 3.14159265   /* the reference's own PI macro (double): Main.cpp:39, MtlBlinn.cpp:12, Sphere.cpp:4 */
#define O_BIAS 0.0001f    /* MtlBlinn.cpp:10, TriObj.cpp:9 */
#define O_EULER 2.7182818f /* MtlBlinn.cpp:11 */
#define O_SHADOW_BIAS 0.00001f /* GenLight.cpp:5 */
#define O_PERP 0.001745f  /* TriObj.cpp:12 */
#define O_MAXLOOP (1 << 20) /* safety cap on the reference's unbounded rejection loops (also in the HIP path) */

// ------------------------------------------------------------------------------------------------
// cyVector.h:271-386 / cyColor.h — same operation order
// ------------------------------------------------------------------------------------------------
struct Vec3 {
    float x, y, z;
    Vec3() {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    Vec3 operator+(const Vec3 &p) const { return Vec3(x + p.x, y + p.y, z + p.z); }
    Vec3 operator-(const Vec3 &p) const { return Vec3(x - p.x, y - p.y, z - p.z); }
    Vec3 operator*(const Vec3 &p) const { return Vec3(x * p.x, y * p.y, z * p.z); }
    Vec3 operator*(float v) const { return Vec3(x * v, y * v, z * v); }
    Vec3 operator/(float v) const { return Vec3(x / v, y / v, z / v); }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    float Dot(const Vec3 &p) const { return x * p.x + y * p.y + z * p.z; }
    Vec3 Cross(const Vec3 &p) const { return Vec3(y * p.z - z * p.y, z * p.x - x * p.z, x * p.y - y * p.x); }
    float LengthSquared() const { return x * x + y * y + z * z; }
    float Length() const { return sqrtf(LengthSquared()); } // cy::Sqrt<float> = sqrtss (cyCore.h:197)
    Vec3 GetNormalized() const { return *this / Length(); }
    bool IsZero() const { return x == 0.f && y == 0.f && z == 0.f; }
};
inline Vec3 operator*(float v, const Vec3 &p) { return p * v; }

struct Color {
    float r, g, b;
    Color() {}
    Color(float a, float c, float d) : r(a), g(c), b(d) {}
    Color operator+(const Color &c) const { return Color(r + c.r, g + c.g, b + c.b); }
    Color operator*(const Color &c) const { return Color(r * c.r, g * c.g, b * c.b); }
    Color operator*(float n) const { return Color(r * n, g * n, b * n); }
    Color operator/(float n) const { return Color(r / n, g / n, b / n); }
    Color &operator+=(const Color &c) { r += c.r; g += c.g; b += c.b; return *this; }
    bool IsBlack() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
    float Gray() const { return (r + g + b) / 3.0f; }
};
inline Color operator*(float v, const Color &c) { return c * v; }
inline float MinF(float a, float b) { return a <= b ? a : b; } // cyCore.h:187-188
inline float MaxF(float a, float b) { return a >= b ? a : b; }
inline Color Black() { return Color(0, 0, 0); }
inline Color NanPurple() { return Color(1.0f, 0.0f, 1.0f); } // cyColor.h:130
inline void ClampColorToWhite(Color &c) // MtlBlinn.cpp:79-83
{
    if (c.r > 1) c.r = 1.f;
    if (c.g > 1) c.g = 1.f;
    if (c.b > 1) c.b = 1.f;
}

struct Ray {
    Vec3 p, dir;
};

struct HitInfo { // scene.h:62-75 (+ face id, which the reference does not keep)
    float z;
    Vec3 p, N, uvw, duvw[2];
    int node;
    bool front;
    int mtlID;
    int face;
    HitInfo() { Init(); }
    void Init()
    {
        z = BHRT_BIGFLOAT; node = -1; front = true; uvw = Vec3(0.5f, 0.5f, 0.5f);
        duvw[0] = Vec3(0, 0, 0); duvw[1] = Vec3(0, 0, 0); mtlID = 0; face = -1;
    }
};

// ------------------------------------------------------------------------------------------------
// math policies: libm (pins the oracle to the compiled reference) / device math (pins the HIP path)
// ------------------------------------------------------------------------------------------------
struct MathLibm {
    static float Sin(float x) { return sinf(x); }
    static float Cos(float x) { return cosf(x); }
    static float Tan(float x) { return tanf(x); }
    static float Acos(float x) { return acosf(x); }
    static float Asin(float x) { return asinf(x); }
    static float Atan2(float y, float x) { return atan2f(y, x); }
    static float Pow(float x, float y) { return powf(x, y); }
    static double PowInt(double x, int n) { return pow(x, (double)n); } // pow(float,int) -> double pow (C++11)
};
struct MathDevice {
    static float Sin(float x) { return bhrt::dm::sinf_(x); }
    static float Cos(float x) { return bhrt::dm::cosf_(x); }
    static float Tan(float x) { return bhrt::dm::tanf_(x); }
    static float Acos(float x) { return bhrt::dm::acosf_(x); }
    static float Asin(float x) { return bhrt::dm::asinf_(x); }
    static float Atan2(float y, float x) { return bhrt::dm::atan2f_(y, x); }
    static float Pow(float x, float y) { return bhrt::dm::powf_(x, y); }
    static double PowInt(double x, int n) { double r = x; for (int i = 1; i < n; i++) r = r * x; return r; }
};

// ------------------------------------------------------------------------------------------------
// scene view over the flat blob
// ------------------------------------------------------------------------------------------------
struct Scene {
    const uint8_t *blob;
    const bhrt_flat_header *H;
    const bhrt_node *nodes;
    const bhrt_mesh *meshes;
    const bhrt_material *materials;
    const bhrt_light *lights;
    const bhrt_texmap *texmaps;
    const bhrt_texture *textures;
    std::vector<std::vector<int>> children; // children[0] = root's, children[i+1] = node i's
    Vec3 dd_x, dd_y;                        // Plane.cpp:3-4 reads the camera globals

    bool Init(const void *b)
    {
        blob = (const uint8_t *)b;
        H = (const bhrt_flat_header *)b;
        if (!b || H->magic != BHRT_FLAT_MAGIC || H->version != BHRT_FLAT_VERSION) { g_err = "bad scene blob"; return false; }
        nodes = (const bhrt_node *)(blob + H->off_nodes);
        meshes = (const bhrt_mesh *)(blob + H->off_meshes);
        materials = (const bhrt_material *)(blob + H->off_materials);
        lights = (const bhrt_light *)(blob + H->off_lights);
        texmaps = (const bhrt_texmap *)(blob + H->off_texmaps);
        textures = (const bhrt_texture *)(blob + H->off_textures);
        children.assign(H->n_nodes + 1, std::vector<int>());
        for (uint32_t i = 0; i < H->n_nodes; i++) children[nodes[i].parent + 1].push_back((int)i);
        dd_x = Vec3(H->camera.dd_x[0], H->camera.dd_x[1], H->camera.dd_x[2]);
        dd_y = Vec3(H->camera.dd_y[0], H->camera.dd_y[1], H->camera.dd_y[2]);
        return true;
    }
};

// Transformation (scene.h:220-227) ---------------------------------------------------------------
inline Vec3 MatMul(const float *c, const Vec3 &p) // cyMatrix.h:682-687
{
    return Vec3(p.x * c[0] + p.y * c[3] + p.z * c[6], p.x * c[1] + p.y * c[4] + p.z * c[7], p.x * c[2] + p.y * c[5] + p.z * c[8]);
}
inline Vec3 TransposeMult(const float *m, const Vec3 &d) // scene.h:238-245
{
    return Vec3(Vec3(m[0], m[1], m[2]).Dot(d), Vec3(m[3], m[4], m[5]).Dot(d), Vec3(m[6], m[7], m[8]).Dot(d));
}
inline Vec3 TransformTo(const bhrt_xform &t, const Vec3 &p) { return MatMul(t.itm, p - Vec3(t.pos[0], t.pos[1], t.pos[2])); }
inline Vec3 TransformFrom(const bhrt_xform &t, const Vec3 &p) { return MatMul(t.tm, p) + Vec3(t.pos[0], t.pos[1], t.pos[2]); }
inline Vec3 VectorTransformFrom(const bhrt_xform &t, const Vec3 &d) { return TransposeMult(t.itm, d); }
const bhrt_xform &IdentityXform()
{
    static bhrt_xform I = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1}};
    return I;
}
inline Ray ToNodeCoords(const bhrt_xform &t, const Ray &ray) // scene.h:490-496
{
    Ray r;
    r.p = TransformTo(t, ray.p);
    r.dir = TransformTo(t, ray.p + ray.dir) - r.p;
    return r;
}
inline void FromNodeCoords(const bhrt_xform &t, HitInfo &h) // scene.h:497-501
{
    h.p = TransformFrom(t, h.p);
    h.N = VectorTransformFrom(t, h.N).GetNormalized();
}

// ------------------------------------------------------------------------------------------------
// random numbers
// ------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t sample_key = 0, key = 0, ctr = 0;
    bool keyed = false;
    void BeginSample(uint32_t seed, uint32_t pixel, uint32_t sample)
    {
        sample_key = bhrt_sample_key(seed, pixel, sample);
        key = sample_key;
        ctr = 0;
    }
    void Section(uint64_t path_code, uint32_t section)
    {
        if (!keyed) return;
        key = bhrt_section_key(sample_key, path_code, section);
        ctr = 0;
    }
    uint32_t wrap = 0xffffffffu; // the counter bits that advance (include/bhrt_rng.h: the low 16 for a keyed photon emission)
    int Rand() { const uint32_t c = ctr; ctr = (c & ~wrap) | ((c + 1u) & wrap); return bhrt_rand31(key, c); }
    bool device_math = false; // device-math mode converts with one multiplication (bhrt_detmath.h::rand_to_unit)
    float ToUnit(int r) const { return device_math ? bhrt::dm::rand_to_unit(r) : (float)((double)r / (BHRT_RAND_MAX)); }
    float Rnd01() // MtlBlinn.cpp:42-49
    {
        float rnd = ToUnit(Rand());
        int guard = 0;
        while ((rnd == 0.0f || rnd == 1.0f) && guard++ < O_MAXLOOP) rnd = ToUnit(Rand());
        return rnd;
    }
};

struct Counters {
    uint64_t closest = 0, shadow = 0, shade = 0;
};

// ------------------------------------------------------------------------------------------------
// the tracer (objects, scene graph, shadows), templated on the math policy only where the
// reference calls libm inside intersection code (sphere uv, plane ray differentials)
// ------------------------------------------------------------------------------------------------
template <class M> struct Tracer {
    const Scene &S;
    Counters *cnt;
    explicit Tracer(const Scene &s, Counters *c) : S(s), cnt(c) {}

    // Sphere::IntersectRay, Objects/Sphere/Sphere.cpp:8-75
    bool IntersectSphere(const Ray &ray, HitInfo &hInfo, int hitSide) const
    {
        Vec3 dir = ray.dir, oc = ray.p;
        float A = dir.Dot(dir);
        float B = 2 * dir.Dot(oc);
        float C = oc.Dot(oc) - 1;
        float DD = B * B - 4 * A * C;
        if (DD > 0) {
            float t1 = (-B + sqrtf(DD)) / (2 * A);
            float t2 = (-B - sqrtf(DD)) / (2 * A);
            float t = BHRT_BIGFLOAT;
            bool hitFront = true;
            if (t1 < 0 && t2 < 0) return false;
            else if (t1 * t2 <= 0) {
                if (hitSide == BHRT_HIT_FRONT) return false;
                t = t1;
                hitFront = false;
            } else if (t1 > 0 && t2 > 0) {
                if (hitSide == BHRT_HIT_FRONT || hitSide == BHRT_HIT_FRONT_AND_BACK) { t = t2; hitFront = true; }
                else if (hitSide == BHRT_HIT_BACK) { t = t1; hitFront = false; }
            }
            if (hInfo.z < t || t <= 0) return false;
            hInfo.z = t;
            hInfo.p = oc + hInfo.z * dir;
            hInfo.N = hInfo.p;
            hInfo.front = hitFront;
            Vec3 d = hInfo.N.GetNormalized();
            Vec3 uvw;
            uvw.x = (float)(0.5f + M::Atan2(d.y, d.x) / (2 * O_PI)); // float / double -> double
            uvw.y = (float)(0.5f - M::Asin(d.z) / (O_PI));
            uvw.z = 0; // the reference leaves uvw.z uninitialised (Sphere.cpp:60); it only reaches textures
            hInfo.uvw = uvw;
            hInfo.duvw[0] = Vec3(0, 0, 0);
            hInfo.duvw[1] = Vec3(0, 0, 0);
            hInfo.face = -1;
            return true;
        }
        return false;
    }

    // Plane::IntersectRay, Objects/Plane/Plane.cpp:8-77
    bool IntersectPlane(const Ray &ray, HitInfo &hInfo, int hitSide) const
    {
        float rayPz = ray.p.z, rayDz = ray.dir.z;
        if (rayDz == 0.0f) return false;
        float t = -rayPz / rayDz;
        if (t <= 0 || t > hInfo.z) return false;
        Vec3 x = ray.p + t * ray.dir;
        if (x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1) return false;
        Vec3 faceNormal(0, 0, 1);
        bool hitFront = ((-ray.dir).Dot(faceNormal) > 0);
        if (!hitFront && hitSide == BHRT_HIT_FRONT) return false;
        else if (hitFront && hitSide == BHRT_HIT_BACK) return false;
        hInfo.p = x;
        hInfo.N = faceNormal;
        hInfo.z = t;
        hInfo.front = hitFront;
        Vec3 uvw(0, 0, 0);
        uvw.x = (1 + hInfo.p.x) / 2.f;
        uvw.y = (1 + hInfo.p.y) / 2.f;
        hInfo.uvw = uvw;
        { // ray differentials from the CAMERA's dd_x/dd_y, Plane.cpp:51-70
            const Vec3 &dd_x = S.dd_x, &dd_y = S.dd_y;
            Vec3 nd = ray.dir.GetNormalized();
            float scaled_t = (t * ray.dir).Length();
            Vec3 dDX = (nd.Dot(nd) * dd_x - nd.Dot(dd_x) * nd) / M::Pow(nd.Dot(nd), 1.5f);
            Vec3 dDY = (nd.Dot(nd) * dd_y - nd.Dot(dd_y) * nd) / M::Pow(nd.Dot(nd), 1.5f);
            float d_t_x = -(0 + scaled_t * dDX.Dot(hInfo.N) / nd.Dot(hInfo.N));
            float d_t_y = -(0 + scaled_t * dDY.Dot(hInfo.N) / nd.Dot(hInfo.N));
            Vec3 hx = (scaled_t * dDX + Vec3(0, 0, 0)) + d_t_x * nd; // "0 + v" = v + 0 (cyVector.h:276)
            Vec3 hy = (scaled_t * dDY + Vec3(0, 0, 0)) + d_t_y * nd;
            hInfo.duvw[0] = hx / 2.f;
            hInfo.duvw[1] = hy / 2.f;
        }
        hInfo.face = -1;
        return true;
    }

    // Box::IntersectRay, Objects/Box/Box.cpp:3-46
    static bool IntersectBox(const float *b, const Ray &r, float t_max, float &t_min)
    {
        // n.Dot(v) with an axis vector keeps the zero products: e.g. (0,0,1).(x,y,z) = 0*x + 0*y + 1*z
        float nDotrr = 0 * r.dir.x + 0 * r.dir.y + 1 * r.dir.z;
        float nDotrp = 0 * r.p.x + 0 * r.p.y + 1 * r.p.z;
        float tz1 = (nDotrr != 0) ? ((0 * b[0] + 0 * b[1] + 1 * b[2]) - nDotrp) / nDotrr : BHRT_BIGFLOAT;
        float tz2 = (nDotrr != 0) ? ((0 * b[3] + 0 * b[4] + 1 * b[5]) - nDotrp) / nDotrr : -BHRT_BIGFLOAT;
        nDotrr = 0 * r.dir.x + 1 * r.dir.y + 0 * r.dir.z;
        nDotrp = 0 * r.p.x + 1 * r.p.y + 0 * r.p.z;
        float ty1 = (nDotrr != 0) ? ((0 * b[0] + 1 * b[1] + 0 * b[2]) - nDotrp) / nDotrr : BHRT_BIGFLOAT;
        float ty2 = (nDotrr != 0) ? ((0 * b[3] + 1 * b[4] + 0 * b[5]) - nDotrp) / nDotrr : -BHRT_BIGFLOAT;
        nDotrr = 1 * r.dir.x + 0 * r.dir.y + 0 * r.dir.z;
        nDotrp = 1 * r.p.x + 0 * r.p.y + 0 * r.p.z;
        float tx1 = (nDotrr != 0) ? ((1 * b[0] + 0 * b[1] + 0 * b[2]) - nDotrp) / nDotrr : BHRT_BIGFLOAT;
        float tx2 = (nDotrr != 0) ? ((1 * b[3] + 0 * b[4] + 0 * b[5]) - nDotrp) / nDotrr : -BHRT_BIGFLOAT;
        float tMin = MaxF(MaxF(MinF(tx1, tx2), MinF(ty1, ty2)), MinF(tz1, tz2));
        float tMax = MinF(MinF(MaxF(tx1, tx2), MaxF(ty1, ty2)), MaxF(tz1, tz2));
        if (tMin <= tMax && tMin < t_max) { t_min = tMin; return true; }
        return false;
    }

    struct MeshView {
        const float *v, *vn, *vt;
        const uint32_t *f, *fn, *ft;
        const bhrt_bvh_node *bvh;
        const uint32_t *elems;
        Vec3 V(uint32_t i) const { return Vec3(v[i * 3], v[i * 3 + 1], v[i * 3 + 2]); }
    };
    MeshView View(int mi) const
    {
        const bhrt_mesh &m = S.meshes[mi];
        MeshView w;
        w.v = (const float *)(S.blob + m.off_v); w.vn = (const float *)(S.blob + m.off_vn); w.vt = (const float *)(S.blob + m.off_vt);
        w.f = (const uint32_t *)(S.blob + m.off_f); w.fn = (const uint32_t *)(S.blob + m.off_fn); w.ft = (const uint32_t *)(S.blob + m.off_ft);
        w.bvh = (const bhrt_bvh_node *)(S.blob + m.off_bvh); w.elems = (const uint32_t *)(S.blob + m.off_elems);
        return w;
    }
    static Vec3 Interpolate(const float *a, const uint32_t *fa, uint32_t face, const Vec3 &bc) // cyTriMesh.h:191
    {
        const float *p0 = a + 3 * fa[face * 3], *p1 = a + 3 * fa[face * 3 + 1], *p2 = a + 3 * fa[face * 3 + 2];
        return Vec3(p0[0], p0[1], p0[2]) * bc.x + Vec3(p1[0], p1[1], p1[2]) * bc.y + Vec3(p2[0], p2[1], p2[2]) * bc.z;
    }

    // TriObj::IntersectTriangle, Objects/TriObj/TriObj.cpp:68-189
    bool IntersectTriangle(const MeshView &w, const Ray &ray, HitInfo &hInfo, int hitSide, uint32_t faceID) const
    {
        Vec3 v0 = w.V(w.f[faceID * 3]), v1 = w.V(w.f[faceID * 3 + 1]), v2 = w.V(w.f[faceID * 3 + 2]);
        Vec3 vN = (v1 - v0).Cross(v2 - v0);
        float t_divisor = vN.Dot(ray.dir);
        if (t_divisor == 0) return false;
        float perp = t_divisor / (vN.Length() * ray.dir.Length());
        if (perp > -O_PERP && perp < O_PERP) return false;
        float t = (vN.Dot(v0) - vN.Dot(ray.p)) / t_divisor;
        if (t <= 0 || t > hInfo.z) return false;
        bool hitFront = t_divisor < 0;
        if (!hitFront && hitSide == BHRT_HIT_FRONT) return false;
        else if (hitFront && hitSide == BHRT_HIT_BACK) return false;
        Vec3 vX = ray.p + t * ray.dir;
        float ax = fabsf(vN.x), ay = fabsf(vN.y), az = fabsf(vN.z);
        float p0x = 0, p0y = 0, p1x = 0, p1y = 0, p2x = 0, p2y = 0, pXx = 0, pXy = 0; // Vec2f() is uninitialised in the reference; a branch below is always taken for finite normals
        if (ax >= ay && ax >= az) { p0x = v0.y; p0y = v0.z; p1x = v1.y; p1y = v1.z; p2x = v2.y; p2y = v2.z; pXx = vX.y; pXy = vX.z; }
        else if (ay >= ax && ay >= az) { p0x = v0.x; p0y = v0.z; p1x = v1.x; p1y = v1.z; p2x = v2.x; p2y = v2.z; pXx = vX.x; pXy = vX.z; }
        else if (az >= ay && az >= ax) { p0x = v0.x; p0y = v0.y; p1x = v1.x; p1y = v1.y; p2x = v2.x; p2y = v2.y; pXx = vX.x; pXy = vX.y; }
        // Vec2::Cross(p) = Vec2(-y,x).Dot(p) = (-y)*p.x + x*p.y   (cyVector.h:260,262)
        auto cross2 = [](float ax_, float ay_, float bx_, float by_) { return (-ay_) * bx_ + ax_ * by_; };
        float a0 = cross2(p1x - pXx, p1y - pXy, p2x - pXx, p2y - pXy) / 2.f;
        float a1 = cross2(p2x - pXx, p2y - pXy, p0x - pXx, p0y - pXy) / 2.f;
        float a2 = cross2(p0x - pXx, p0y - pXy, p1x - pXx, p1y - pXy) / 2.f;
        if ((a2 < 0 || a1 < 0 || a0 < 0) && !(a0 < 0 && a1 < 0 && a2 < 0)) return false;
        float a = a0 + a1 + a2;
        Vec3 bc(a0 / a, a1 / a, a2 / a);
        hInfo.z = t;
        hInfo.N = Interpolate(w.vn, w.fn, faceID, bc);
        hInfo.p = vX;
        hInfo.front = hitFront;
        hInfo.uvw = Interpolate(w.vt, w.ft, faceID, bc);
        hInfo.duvw[0] = Vec3(0, 0, 0);
        hInfo.duvw[1] = Vec3(0, 0, 0);
        hInfo.face = (int)faceID;
        return true;
    }

    // TriObj::TraceBVHNode, TriObj.cpp:192-270
    bool TraceBVHNode(const MeshView &w, const Ray &ray, HitInfo &hInfo, int hitSide, uint32_t nodeID) const
    {
        const bhrt_bvh_node &n = w.bvh[nodeID];
        if (n.data & 0x80000000u) {
            uint32_t count = ((n.data >> 28) & 7u) + 1, off = n.data & 0x0fffffffu;
            bool bHit = false;
            for (uint32_t i = 0; i < count; i++)
                if (IntersectTriangle(w, ray, hInfo, hitSide, w.elems[off + i])) bHit = true;
            return bHit;
        }
        uint32_t c1 = n.data & 0x7fffffffu, c2 = c1 + 1;
        float tmin1 = BHRT_BIGFLOAT, tmin2 = BHRT_BIGFLOAT;
        bool b1 = IntersectBox(w.bvh[c1].b, ray, hInfo.z, tmin1);
        bool b2 = IntersectBox(w.bvh[c2].b, ray, hInfo.z, tmin2);
        if (!b1 && !b2) return false;
        if (tmin1 < tmin2) {
            if (TraceBVHNode(w, ray, hInfo, hitSide, c1)) {
                HitInfo temp = hInfo;
                if (temp.z > tmin2)
                    if (TraceBVHNode(w, ray, temp, hitSide, c2)) hInfo = temp;
                return true;
            }
            return TraceBVHNode(w, ray, hInfo, hitSide, c2);
        } else {
            if (TraceBVHNode(w, ray, hInfo, hitSide, c2)) {
                HitInfo temp = hInfo;
                if (temp.z > tmin1)
                    if (TraceBVHNode(w, ray, temp, hitSide, c1)) hInfo = temp;
                return true;
            }
            return TraceBVHNode(w, ray, hInfo, hitSide, c1);
        }
    }
    // TriObj::IntersectRay, TriObj.cpp:17-39
    bool IntersectMesh(int mi, const Ray &ray, HitInfo &hInfo, int hitSide) const
    {
        MeshView w = View(mi);
        float tmin = -1;
        if (IntersectBox(w.bvh[1].b, ray, hInfo.z, tmin)) return TraceBVHNode(w, ray, hInfo, hitSide, 1);
        return false;
    }
    // TriObj::TraceBVHShadow, TriObj.cpp:272-307.  `A | B` evaluates A (child1) first in the g++ build of the
    // reference (checked against oracle/_ref by tests/test_ref_parity.py).
    bool TraceBVHShadow(const MeshView &w, const Ray &ray, float &t_min, bool &hitOnce, uint32_t nodeID) const
    {
        if (hitOnce) return hitOnce;
        const bhrt_bvh_node &n = w.bvh[nodeID];
        if (n.data & 0x80000000u) {
            uint32_t count = ((n.data >> 28) & 7u) + 1, off = n.data & 0x0fffffffu;
            HitInfo h;
            for (uint32_t i = 0; i < count; i++)
                if (IntersectTriangle(w, ray, h, BHRT_HIT_FRONT, w.elems[off + i])) { hitOnce = true; t_min = h.z; }
            return hitOnce;
        }
        uint32_t c1 = n.data & 0x7fffffffu, c2 = c1 + 1;
        float tmin1 = BHRT_BIGFLOAT, tmin2 = BHRT_BIGFLOAT;
        bool b1 = IntersectBox(w.bvh[c1].b, ray, BHRT_BIGFLOAT, tmin1);
        bool b2 = IntersectBox(w.bvh[c2].b, ray, BHRT_BIGFLOAT, tmin2);
        if (!b1 && !b2) return false;
        bool r1 = TraceBVHShadow(w, ray, t_min, hitOnce, c1);
        bool r2 = TraceBVHShadow(w, ray, t_min, hitOnce, c2);
        return r1 | r2;
    }
    // TriObj::ShadowRecursive, TriObj.cpp:41-66
    bool MeshShadow(int mi, const Ray &ray, float t_max) const
    {
        MeshView w = View(mi);
        float tmin = -1, t_min = BHRT_BIGFLOAT;
        bool bHit = false;
        if (IntersectBox(w.bvh[1].b, ray, BHRT_BIGFLOAT, tmin)) TraceBVHShadow(w, ray, t_min, bHit, 1);
        return bHit && t_min > O_BIAS && t_min < t_max;
    }

    // recursive(), Main.cpp:389-413.  root = -1 is rootNode.
    void Recursive(int root, const Ray &ray, HitInfo &outHit, bool &bHit, int hitSide) const
    {
        const std::vector<int> &ch = S.children[root + 1];
        if (ch.empty()) return;
        for (int c : ch) {
            const bhrt_node &n = S.nodes[c];
            Ray tr = ToNodeCoords(n.xf, ray);
            bool hit = false;
            switch (n.obj_type) {
            case BHRT_OBJ_SPHERE: hit = IntersectSphere(tr, outHit, hitSide); break;
            case BHRT_OBJ_PLANE: hit = IntersectPlane(tr, outHit, hitSide); break;
            case BHRT_OBJ_MESH: hit = IntersectMesh(n.mesh, tr, outHit, hitSide); break;
            default: break;
            }
            if (hit) {
                outHit.node = c;
                bHit = true;
                FromNodeCoords(n.xf, outHit);
            }
            Recursive(c, tr, outHit, bHit, hitSide);
        }
        for (int c : ch)
            if (c == outHit.node) {
                FromNodeCoords(root < 0 ? IdentityXform() : S.nodes[root].xf, outHit);
                break;
            }
    }
    void Closest(const Ray &ray, HitInfo &h, bool &bHit, int hitSide) const
    {
        if (cnt) cnt->closest++;
        Recursive(-1, ray, h, bHit, hitSide);
    }

    // ShadowRayRecursive, Lights/GenLight.cpp:15-69
    bool ShadowRayRecursive(int node, const Ray &ray, float t_max) const
    {
        const bhrt_xform &xf = node < 0 ? IdentityXform() : S.nodes[node].xf;
        Ray tr = ToNodeCoords(xf, ray);
        for (int c : S.children[node + 1])
            if (ShadowRayRecursive(c, tr, t_max)) return true;
        if (node < 0) return false;
        const bhrt_node &n = S.nodes[node];
        if (n.obj_type == BHRT_OBJ_SPHERE) {
            Vec3 dir = tr.dir, oc = tr.p;
            float A = dir.Dot(dir);
            float B = 2 * dir.Dot(oc);
            float C = oc.Dot(oc) - 1;
            float DD = B * B - 4 * A * C;
            if (DD > 0) {
                float t1 = (-B + sqrtf(DD)) / (2 * A);
                float t2 = (-B - sqrtf(DD)) / (2 * A);
                float t = MinF(t1, t2);
                if (t < 0) return false;
                if (t < t_max && t > O_SHADOW_BIAS) return true;
            }
        } else if (n.obj_type == BHRT_OBJ_PLANE) {
            float t = -tr.p.z / tr.dir.z;
            if (t < 0) return false;
            Vec3 x = ray.p + t * ray.dir; // the UN-transformed ray (GenLight.cpp:54)
            if (x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1) return false;
            if (t < t_max && t > O_SHADOW_BIAS) return true;
            return false;
        } else if (n.obj_type == BHRT_OBJ_MESH) {
            return MeshShadow(n.mesh, tr, t_max);
        }
        return false;
    }
    float Shadow(const Ray &ray, float t_max) const // GenLight::Shadow, GenLight.cpp:10-13
    {
        if (cnt) cnt->shadow++;
        return ShadowRayRecursive(-1, ray, t_max) ? 0.f : 1.f;
    }
};

// ------------------------------------------------------------------------------------------------
// textures (Scenes/scene.h:137-146,318-337,344-354,364-422; Textures/Texture.cpp:97-136)
// ------------------------------------------------------------------------------------------------
inline float Halton(int index, int base) // scene.h:137-146
{
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (i % base);
        f /= (float)base;
    }
    return r;
}
template <class M> struct Textures {
    const Scene &S;
    float tapx[32], tapy[32];
    explicit Textures(const Scene &s) : S(s)
    {
        for (int i = 1; i < 32; i++) { // scene.h:322-329 (elliptic)
            float x = Halton(i, 2), y = Halton(i, 3);
            float r = sqrtf(x) * 0.5f;
            x = r * M::Sin(y * (float)M_PI * 2);
            y = r * M::Cos(y * (float)M_PI * 2);
            tapx[i] = x; tapy[i] = y;
        }
    }
    static Vec3 TileClamp(const Vec3 &uvw) // scene.h:344-354
    {
        Vec3 u;
        u.x = uvw.x - (int)uvw.x; u.y = uvw.y - (int)uvw.y; u.z = uvw.z - (int)uvw.z;
        if (u.x < 0) u.x += 1;
        if (u.y < 0) u.y += 1;
        if (u.z < 0) u.z += 1;
        return u;
    }
    Color SampleTexture(const bhrt_texture &t, const Vec3 &uvw) const
    {
        if (t.type == BHRT_TEX_CHECKER) { // Texture.cpp:127-136
            Vec3 u = TileClamp(uvw);
            Color c1(t.color1[0], t.color1[1], t.color1[2]), c2(t.color2[0], t.color2[1], t.color2[2]);
            if (u.x <= 0.5f) return u.y <= 0.5f ? c1 : c2;
            return u.y <= 0.5f ? c2 : c1;
        }
        // TextureFile::Sample, Texture.cpp:97-123
        int width = t.width, height = t.height;
        if (width + height == 0) return Color(0, 0, 0);
        const uint8_t *data = S.blob + t.off_data;
        Vec3 u = TileClamp(uvw);
        float x = width * u.x, y = height * u.y;
        int ix = (int)x, iy = (int)y;
        float fx = x - ix, fy = y - iy;
        // (the products wrap in 32 bits like the reference's compiled code does for a huge |x|: unsigned arithmetic, no signed overflow)
        if (ix < 0) ix = (int)((uint32_t)ix - (uint32_t)(ix / width - 1) * (uint32_t)width);
        if (ix >= width) ix = (int)((uint32_t)ix - (uint32_t)(ix / width) * (uint32_t)width);
        int ixp = ix + 1;
        if (ixp >= width) ixp -= width;
        if (iy < 0) iy = (int)((uint32_t)iy - (uint32_t)(iy / height - 1) * (uint32_t)height);
        if (iy >= height) iy = (int)((uint32_t)iy - (uint32_t)(iy / height) * (uint32_t)height);
        int iyp = iy + 1;
        if (iyp >= height) iyp -= height;
        auto texel = [&](int yy, int xx) { const uint8_t *p = data + 3 * ((size_t)yy * width + xx); return Color(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f); };
        return texel(iy, ix) * ((1 - fx) * (1 - fy)) + texel(iy, ixp) * (fx * (1 - fy)) + texel(iyp, ix) * ((1 - fx) * fy) + texel(iyp, ixp) * (fx * fy);
    }
    Color SampleFiltered(const bhrt_texture &t, const Vec3 &uvw, const Vec3 duvw[2]) const // scene.h:318-337
    {
        Color c = SampleTexture(t, uvw);
        if (duvw[0].LengthSquared() + duvw[1].LengthSquared() == 0) return c;
        for (int i = 1; i < 32; i++) c += SampleTexture(t, uvw + tapx[i] * duvw[0] + tapy[i] * duvw[1]);
        return c / float(32);
    }
    Color SampleMap(int map, const Vec3 &uvw) const // TextureMap::Sample(uvw), scene.h:371
    {
        const bhrt_texmap &m = S.texmaps[map];
        if (m.texture < 0) return Color(0, 0, 0);
        return SampleTexture(S.textures[m.texture], TransformTo(m.xf, uvw));
    }
    Color SampleMap(int map, const Vec3 &uvw, const Vec3 duvw[2]) const // scene.h:372-380
    {
        const bhrt_texmap &m = S.texmaps[map];
        if (m.texture < 0) return Color(0, 0, 0);
        Vec3 u = TransformTo(m.xf, uvw);
        Vec3 d[2];
        d[0] = TransformTo(m.xf, duvw[0] + uvw) - u;
        d[1] = TransformTo(m.xf, duvw[1] + uvw) - u;
        return SampleFiltered(S.textures[m.texture], u, d);
    }
    static Color ColorOf(const bhrt_texcolor &tc) { return Color(tc.color[0], tc.color[1], tc.color[2]); }
    Color Sample(const bhrt_texcolor &tc, const Vec3 &uvw) const { return tc.map >= 0 ? ColorOf(tc) * SampleMap(tc.map, uvw) : ColorOf(tc); } // scene.h:410
    Color Sample(const bhrt_texcolor &tc, const Vec3 &uvw, const Vec3 duvw[2]) const { return tc.map >= 0 ? ColorOf(tc) * SampleMap(tc.map, uvw, duvw) : ColorOf(tc); }
    Color SampleEnvironment(const bhrt_texcolor &tc, const Vec3 &dir) const // scene.h:414-420
    {
        float z = M::Asin(-dir.z) / float(M_PI) + 0.5f;
        float x = dir.x / (fabsf(dir.x) + fabsf(dir.y));
        float y = dir.y / (fabsf(dir.x) + fabsf(dir.y));
        return Sample(tc, Vec3(0.5f, 0.5f, 0.0f) + z * (x * Vec3(0.5f, 0.5f, 0) + y * Vec3(-0.5f, 0.5f, 0)));
    }
};

// ------------------------------------------------------------------------------------------------
// photon map hook (filled in by the photon section below)
// ------------------------------------------------------------------------------------------------
struct PhotonMapView;
PhotonMapView *g_photon_map = nullptr;
bool PhotonEstimate(const PhotonMapView *pm, Color &irrad, Vec3 &dir, float radius, const Vec3 &pos, const Vec3 &normal);

// ------------------------------------------------------------------------------------------------
// MtlBlinn — the integrator (Materials/Blinn/MtlBlinn.cpp)
// ------------------------------------------------------------------------------------------------
template <class M> struct Shader {
    const Scene &S;
    const Tracer<M> &T;
    const Textures<M> &X;
    Rng &rng;
    Counters *cnt;
    bool photon_gather;
    Shader(const Scene &s, const Tracer<M> &t, const Textures<M> &x, Rng &r, Counters *c, bool pg) : S(s), T(t), X(x), rng(r), cnt(c), photon_gather(pg) {}

    static float GetK(const bhrt_texcolor &c) { return MaxF(MaxF(c.color[0], c.color[1]), c.color[2]); } // MtlBlinn.cpp:68-69

    Vec3 GetRandomCrossingVector(const Vec3 &V) // MtlBlinn.cpp:591-600
    {
        Vec3 rndVec(0, 0, 1);
        int guard = 0;
        while (V.Cross(rndVec).IsZero() && guard++ < O_MAXLOOP) {
            float c = rng.Rnd01(), b = rng.Rnd01(), a = rng.Rnd01(); // g++ evaluates the constructor arguments right to left (checked against oracle/_ref)
            rndVec = Vec3(a, b, c);
        }
        return rndVec;
    }
    Vec3 GetSampleAlongNormal(const Vec3 &N, float R) // MtlBlinn.cpp:602-617
    {
        float r = rng.Rnd01();
        r = sqrtf(r) * R;
        float theta = (float)(rng.Rnd01() * 2 * O_PI);
        float x = r * M::Cos(theta), y = r * M::Sin(theta);
        Vec3 axis1 = GetRandomCrossingVector(N).Cross(N);
        Vec3 axis2 = axis1.Cross(N);
        return N + axis1.GetNormalized() * x + axis2.GetNormalized() * y;
    }
    Vec3 GetSampleAlongLightDirection(const Vec3 &N, float glossiness, float &o_theta) // MtlBlinn.cpp:619-635
    {
        float u = rng.Rnd01();
        float p = M::Pow(u, 1.f / (glossiness + 1.f));
        float weightTheta = M::Acos(MinF(1.f, MaxF(-1.f, p))); // ACosSafe, cyCore.h:193
        o_theta = weightTheta;
        float R = M::Tan(weightTheta);
        float phi = (float)(rng.Rnd01() * 2 * O_PI);
        float x = R * M::Cos(phi), y = R * M::Sin(phi);
        Vec3 axis1 = GetRandomCrossingVector(N).Cross(N);
        Vec3 axis2 = axis1.Cross(N);
        return N + axis1.GetNormalized() * x + axis2.GetNormalized() * y;
    }
    Vec3 GetSampleInSemiSphere(const Vec3 &N, float &o_theta) // MtlBlinn.cpp:697-716 (tail recursion -> loop)
    {
        for (int guard = 0; guard < O_MAXLOOP; guard++) {
            Vec3 axisY = (N.Cross(GetRandomCrossingVector(N))).GetNormalized();
            Vec3 axisX = N.Cross(axisY);
            float phi = (float)(rng.Rnd01() * 2 * O_PI);
            float rnd = rng.Rnd01();
            float theta = 0.5f * M::Acos(MinF(1.f, MaxF(-1.f, 1 - 2 * rnd)));
            o_theta = theta;
            float sinTheta = M::Sin(theta);
            Vec3 retVec = sinTheta * M::Cos(phi) * axisX + sinTheta * M::Sin(phi) * axisY + M::Cos(theta) * N;
            if (N.Dot(retVec) <= 0) continue;
            return retVec;
        }
        return N;
    }
    // MtlBlinn.cpp:354-378
    Vec3 GIUseSpecularDirOrDiffuseDir(bool &useSpecular, const Vec3 &vN, const Vec3 &vV, float kd, float ks, float glossiness)
    {
        float diffuseTheta = 0;
        Vec3 diffuseRayDir = GetSampleInSemiSphere(vN, diffuseTheta).GetNormalized();
        float p_diffuseTheta = M::Sin(2 * diffuseTheta);
        float specularTheta = 0;
        float cosvVvN = vN.Dot(vV);
        Vec3 vR = 2 * cosvVvN * vN - vV;
        Vec3 specRayDir = GetSampleAlongLightDirection(vR, glossiness, specularTheta);
        float p_specularTheta = M::Pow(M::Cos(specularTheta), glossiness);
        float P_Diffuse = kd * p_diffuseTheta;
        float P_sum = P_Diffuse + ks * p_specularTheta;
        float P_Diffuse_Norm = P_Diffuse / P_sum;
        float rnd = rng.Rnd01();
        useSpecular = rnd >= P_Diffuse_Norm;
        return useSpecular ? specRayDir : diffuseRayDir;
    }
    // MtlBlinn.cpp:637-695
    Vec3 GetSampleInLight(const bhrt_texcolor &diffuse, const bhrt_texcolor &specular, const bhrt_light &light, const HitInfo &hInfo, float glossiness)
    {
        if (light.type == BHRT_LIGHT_POINT) {
            float kd = GetK(diffuse), ks = GetK(specular);
            float p_diffuse = 0, p_specular = 0;
            Vec3 diffuse_vL, specular_vL;
            Vec3 vL = Vec3(light.vec[0], light.vec[1], light.vec[2]) - hInfo.p;
            {
                float diffuseTheta = 0;
                diffuse_vL = GetSampleAlongLightDirection(vL.GetNormalized(), glossiness, diffuseTheta);
                p_diffuse = M::Pow(M::Cos(diffuseTheta), glossiness);
            }
            if (ks == 0 && kd != 0) return diffuse_vL.GetNormalized();
            {
                float r = rng.Rnd01();
                float R = sqrtf(r) * (int)light.size; // PointLight::GetSize() returns int (lights.h:76)
                float specularTheta = (float)(rng.Rnd01() * 2 * O_PI);
                float x = R * M::Cos(specularTheta), y = R * M::Sin(specularTheta);
                Vec3 axis1 = GetRandomCrossingVector(vL).Cross(vL);
                Vec3 axis2 = axis1.Cross(vL);
                specular_vL = vL + axis1.GetNormalized() * x + axis2.GetNormalized() * y;
                p_specular = 2 * r / (R * R);
            }
            if (ks != 0 && kd == 0) return specular_vL.GetNormalized();
            float P_Diffuse = kd * p_diffuse;
            float P_Specular = ks * p_specular;
            float P_sum = P_Diffuse + P_Specular;
            float P_Diffuse_Norm = P_Diffuse / P_sum;
            float rnd = rng.Rnd01();
            bool useSpecular = rnd >= P_Diffuse_Norm;
            return useSpecular ? specular_vL.GetNormalized() : diffuse_vL.GetNormalized();
        }
        // -light->Direction(p).GetNormalized(): direct light = its direction; ambient = (0,0,0) -> NaN (lights.h:33,51)
        Vec3 d = light.type == BHRT_LIGHT_DIRECT ? Vec3(light.vec[0], light.vec[1], light.vec[2]) : Vec3(0, 0, 0);
        return -(d.GetNormalized());
    }
    // Light::Illuminate: lights.h:32,50 and Lights/PointLight.cpp:7-18
    Color Illuminate(const bhrt_light &l, const Vec3 &p, const Vec3 &N)
    {
        Color intensity(l.intensity[0], l.intensity[1], l.intensity[2]);
        if (l.type == BHRT_LIGHT_AMBIENT) return intensity;
        if (l.type == BHRT_LIGHT_DIRECT) {
            Ray r; r.p = p; r.dir = -Vec3(l.vec[0], l.vec[1], l.vec[2]);
            return T.Shadow(r, BHRT_BIGFLOAT) * intensity;
        }
        Vec3 centerDir = Vec3(l.vec[0], l.vec[1], l.vec[2]) - p;
        float r = centerDir.Length();
        float rr = r * r;
        if (rr == 0) return Color(1, 1, 1) * BHRT_BIGFLOAT;
        Ray sr; sr.p = p;
        if (l.size > 0) sr.dir = GetSampleAlongNormal(centerDir, l.size);
        else sr.dir = centerDir;
        return T.Shadow(sr, 1) * intensity / rr;
    }

    // PathTracing_DiffuseNSpecular, MtlBlinn.cpp:304-351
    Color DiffuseNSpecular(const bhrt_texcolor &diffuse, const bhrt_texcolor &specular, float glossiness, const HitInfo &hInfo, const Vec3 &vN, const Vec3 &vV)
    {
        Color outColor = Black();
        if (S.H->n_lights > 0) { // zero lights is UB in the reference: defined here as "no direct term"
            float rnd = rng.Rnd01();
            uint32_t i = 0;
            while (rnd > Color(S.lights[i].intensity[0], S.lights[i].intensity[1], S.lights[i].intensity[2]).Gray() / S.H->all_light_intensity && i < S.H->n_lights - 1) i++;
            const bhrt_light &light = S.lights[i];
            Vec3 vL = GetSampleInLight(diffuse, specular, light, hInfo, glossiness);
            float cosTheta = vL.Dot(vN);
            if (cosTheta > 0) {
                Vec3 vH = (vL + vV).GetNormalized();
                Color irrad = Illuminate(light, hInfo.p, vN);
                Color brdfXCosTheta = X.Sample(diffuse, hInfo.uvw, hInfo.duvw) * cosTheta + X.Sample(specular, hInfo.uvw, hInfo.duvw) * M::Pow(vH.Dot(vN), glossiness);
                outColor += irrad * brdfXCosTheta;
            }
        }
        if (photon_gather && g_photon_map) { // MtlBlinn.cpp:329-342 (#ifdef USE_PhotonMap)
            Vec3 vL(0, 0, 0);
            Color irr = Black();
            PhotonEstimate(g_photon_map, irr, vL, 0.5f, hInfo.p, hInfo.N);
            float cosTheta = -vL.Dot(vN);
            if (cosTheta > 0) {
                Vec3 vH = (vL + vV).GetNormalized();
                Color brdf = X.Sample(diffuse, hInfo.uvw, hInfo.duvw) + X.Sample(specular, hInfo.uvw, hInfo.duvw) * M::Pow(vH.Dot(vN), glossiness) / cosTheta;
                outColor += brdf * irr;
            }
        }
        ClampColorToWhite(outColor);
        if (isnan(outColor.r)) return Black();
        return outColor;
    }

    // PathTracing_GlobalIllumination, MtlBlinn.cpp:383-433
    Color GlobalIllumination(const bhrt_texcolor &diffuse, const bhrt_texcolor &specular, float glossiness, const HitInfo &hInfo, const Vec3 &vN, const Vec3 &vV, int bounce, int gi, uint64_t code)
    {
        if (gi < 0) return Black();
        Color outColor = Black();
        bool useSpecular;
        Ray GIRay;
        GIRay.dir = GIUseSpecularDirOrDiffuseDir(useSpecular, vN, vV, GetK(diffuse), GetK(specular), glossiness);
        GIRay.p = hInfo.p + vN * O_BIAS;
        HitInfo reflH;
        bool bHit = false;
        T.Closest(GIRay, reflH, bHit, BHRT_HIT_FRONT);
        if (bHit && reflH.node >= 0) {
            Color indirect = Black();
            if (fabsf(reflH.z) > O_BIAS) indirect = Shade(GIRay, reflH, bounce, gi - 1, code * 2 + 1);
            outColor += indirect * X.Sample(useSpecular ? specular : diffuse, hInfo.uvw, hInfo.duvw);
        } else {
            Vec3 d = GIRay.dir;
            if (d.x == d.y && d.x == 0) outColor += NanPurple();
            else {
                Color env = X.SampleEnvironment(S.H->environment, d) * X.Sample(useSpecular ? specular : diffuse, hInfo.uvw, hInfo.duvw);
                if (!(isnan(env.r) || isnan(env.g) || isnan(env.b))) outColor += env;
            }
        }
        if (isnan(outColor.r)) return NanPurple();
        ClampColorToWhite(outColor);
        return outColor;
    }

    // HandleRayWhenRefractionRayOut, MtlBlinn.cpp:543-589
    Ray HandleRayWhenRefractionRayOut(const Ray &inRay, const HitInfo &inHit, float ior, bool &toOut, float refractionGlossiness)
    {
        Vec3 vN = inHit.N;
        Vec3 vV = -inRay.dir;
        float cosPhi1 = vV.Dot(-vN);
        float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
        float sinPhi2 = ior * sinPhi1;
        if (sinPhi2 <= 1) {
            float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
            Vec3 vTn = vN * cosPhi2;
            Vec3 vNxV = vN.Cross(vV);
            Vec3 vTp = vN.Cross(vNxV).GetNormalized() * sinPhi2;
            Vec3 vT = vTn + vTp;
            Vec3 vT_sampled = vT.GetNormalized();
            if (refractionGlossiness > 0) {
                float dotSign = 0;
                int guard = 0;
                while (dotSign <= 0 && guard++ < O_MAXLOOP) {
                    float theta = 0;
                    vT_sampled = GetSampleAlongLightDirection(vT, refractionGlossiness, theta);
                    dotSign = vT_sampled.Dot(vN);
                }
            }
            Ray o;
            o.dir = vT_sampled.GetNormalized();
            o.p = inHit.p + vN * O_BIAS;
            toOut = true;
            return o;
        }
        Vec3 vR = (-2 * cosPhi1 * vN - vV);
        Ray r;
        r.dir = vR;
        r.p = inHit.p - vN * O_BIAS;
        toOut = false;
        return r;
    }
    // RefractionOut, MtlBlinn.cpp:521-541
    Color RefractionOut(const Ray &outRay, const Color &absorption, const Color &refraction, int bounce, int gi, uint64_t code)
    {
        Color outColor = Black();
        HitInfo h;
        bool bHit = false;
        T.Closest(outRay, h, bHit, BHRT_HIT_FRONT);
        if (bHit && h.node >= 0) {
            float fr = M::Pow(O_EULER, -absorption.r * h.z);
            float fg = M::Pow(O_EULER, -absorption.g * h.z);
            float fb = M::Pow(O_EULER, -absorption.b * h.z);
            outColor = refraction * Color(fr, fg, fb) * Shade(outRay, h, bounce, gi - 1, code * 2);
        } else
            outColor = refraction * X.SampleEnvironment(S.H->environment, outRay.dir);
        ClampColorToWhite(outColor);
        return outColor;
    }
    // RefractionRecusive, MtlBlinn.cpp:476-519
    Color RefractionRecursive(const Color &refraction, float ior, const Vec3 &vT, const HitInfo &hInfo, const Vec3 &vN, float refrGloss, const Color &absorption, int bounce, int gi, uint64_t code)
    {
        Ray in;
        in.dir = vT;
        in.p = hInfo.p - vN * O_BIAS;
        HitInfo h;
        bool bHit = false; // uninitialised in the reference (MtlBlinn.cpp:482); it also tests node != nullptr
        T.Closest(in, h, bHit, BHRT_HIT_FRONT_AND_BACK);
        if (bHit && h.node >= 0) {
            Color c = Black();
            if (!h.front) {
                bool out;
                Ray next = HandleRayWhenRefractionRayOut(in, h, ior, out, refrGloss);
                if (out) c = RefractionOut(next, absorption, refraction, bounce, gi, code);
                else {
                    if (bounce <= 0) c = Black();
                    else {
                        bounce--;
                        c = RefractionRecursive(refraction, ior, next.dir, h, h.N, refrGloss, absorption, bounce, gi, code);
                    }
                }
            } else
                c = Shade(in, h, bounce, gi - 1, code * 2);
            ClampColorToWhite(c);
            return c;
        }
        return NanPurple();
    }
    // PathTracing_Refraction, MtlBlinn.cpp:437-473
    Color Refraction(const Color &refraction, const Color &absorption, float ior, const HitInfo &hInfo, float cosPhi1, const Vec3 &vN, const Vec3 &vV, int bounce, int gi, float refrGloss, uint64_t code)
    {
        Color c = Black();
        if (bounce <= 0) return c;
        if (!refraction.IsBlack()) {
            float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
            float sinPhi2 = sinPhi1 / ior;
            float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
            Vec3 vTn = -cosPhi2 * vN;
            Vec3 vNxV = vN.Cross(vV);
            Vec3 vTp = vN.Cross(vNxV).GetNormalized() * sinPhi2;
            Vec3 vT = vTn + vTp;
            Vec3 vT_sampled = vT.GetNormalized();
            if (refrGloss > 0) {
                float dotSign = 0;
                int guard = 0;
                while (dotSign >= 0 && guard++ < O_MAXLOOP) {
                    float theta = 0;
                    vT_sampled = GetSampleAlongLightDirection(vT, refrGloss, theta);
                    dotSign = vT_sampled.Dot(vN);
                }
            }
            c = RefractionRecursive(refraction, ior, vT_sampled.GetNormalized(), hInfo, vN, refrGloss, absorption, bounce, gi, code);
        }
        ClampColorToWhite(c);
        return c;
    }

    // MtlBlinn::Shade, MtlBlinn.cpp:89-138 (+ MultiMtl::Shade, materials.h:71)
    Color Shade(const Ray &ray, const HitInfo &hInfo, int bounce, int gi, uint64_t code)
    {
        if (cnt) cnt->shade++;
        int mi = hInfo.node >= 0 ? S.nodes[hInfo.node].material : -1;
        if (mi < 0) return Black(); // node without material: null deref in the reference; defined as black
        const bhrt_material &m = S.materials[mi];
        if (m.kind == BHRT_MTL_WHITE) return Color(1, 1, 1);
        Color outColor = Black();
        Vec3 vN = hInfo.N.GetNormalized();
        Vec3 vV = (ray.p - hInfo.p).GetNormalized();
        float cosPhi1 = vN.Dot(vV);
        if (cosPhi1 > 1) cosPhi1 = 1;
        if (cosPhi1 <= 0) cosPhi1 = 0;
        float ior = m.ior;
        float R0 = (float)M::PowInt((double)((1 - ior) / (1 + ior)), 2);
        float fresnel = (float)(R0 + (1 - R0) * M::PowInt((double)(1 - cosPhi1), 5));
        Color refrC(m.refraction.color[0], m.refraction.color[1], m.refraction.color[2]);
        Color fresSpec = Color(m.specular.color[0], m.specular.color[1], m.specular.color[2]) + fresnel * refrC;
        ClampColorToWhite(fresSpec);
        bhrt_texcolor newSpecular = m.specular;
        newSpecular.color[0] = fresSpec.r; newSpecular.color[1] = fresSpec.g; newSpecular.color[2] = fresSpec.b;
        float refrGloss = 0;
        if (m.glossiness > 50) refrGloss = m.glossiness;
        rng.Section(code, BHRT_SEC_REFRACTION);
        outColor += Refraction((1 - fresnel) * refrC, Color(m.absorption[0], m.absorption[1], m.absorption[2]), ior, hInfo, cosPhi1, vN, vV, bounce, gi, refrGloss, code);
        if (outColor.r >= 1 && outColor.g >= 1 && outColor.b >= 1) return outColor;
        rng.Section(code, BHRT_SEC_GI);
        outColor += GlobalIllumination(m.diffuse, newSpecular, m.glossiness, hInfo, vN, vV, bounce, gi, code);
        if (outColor.r >= 1 && outColor.g >= 1 && outColor.b >= 1) return outColor;
        rng.Section(code, BHRT_SEC_DIRECT);
        outColor += DiffuseNSpecular(m.diffuse, newSpecular, m.glossiness, hInfo, vN, vV);
        if (outColor.r >= 1 && outColor.g >= 1 && outColor.b >= 1) return outColor;
        if (isnan(outColor.r)) outColor = NanPurple();
        return outColor;
    }
};

// ------------------------------------------------------------------------------------------------
// frame loop
// ------------------------------------------------------------------------------------------------
inline uint8_t FloatToByte(float r) // cyColor.h:271-272
{
    int v = int(r * 255 + 0.5f);
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <class M>
int RenderT(const Scene &S, const oracle_opts &o, float *samples, float *radiance, uint8_t *rgb8, oracle_stats *stats)
{
    const bhrt_camera &cam = S.H->camera;
    const int W = cam.width, Hh = cam.height;
    int x0 = o.x0, y0 = o.y0, x1 = o.x1, y1 = o.y1;
    if (x1 <= 0 || y1 <= 0) { x0 = 0; y0 = 0; x1 = W; y1 = Hh; }
    if (x0 < 0 || y0 < 0 || x1 > W || y1 > Hh || x0 >= x1 || y0 >= y1) { g_err = "bad region"; return 2; }
    const int rw = x1 - x0, rh = y1 - y0, spp = o.spp > 0 ? o.spp : 1;
    const Vec3 topLeft(cam.top_left[0], cam.top_left[1], cam.top_left[2]);
    const Vec3 dd_x = S.dd_x, dd_y = S.dd_y;
    const Vec3 camPos(cam.pos[0], cam.pos[1], cam.pos[2]);
    Textures<M> X(S);
    uint64_t tot_closest = 0, tot_shadow = 0, tot_shade = 0;
    auto t0 = std::chrono::steady_clock::now();
    const int nthreads = o.threads > 0 ? o.threads : 0;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tot_closest, tot_shadow, tot_shade) num_threads(o.threads > 0 ? o.threads : 1)
    for (int jj = 0; jj < rh; jj++) {
        Counters cnt;
        Tracer<M> T(S, &cnt);
        Rng rng;
        rng.keyed = o.rng_mode == ORACLE_RNG_KEYED;
        rng.device_math = std::is_same<M, MathDevice>::value;
        Shader<M> sh(S, T, X, rng, &cnt, o.photon_gather != 0);
        const int j = y0 + jj;
        for (int i = x0; i < x1; i++) {
            // PathTracing(), Main.cpp:143-172; pixel "centre" = corner because 1/2 == 0
            Vec3 pixelCenter = topLeft + (float)(i + 1 / 2) * dd_x - (float)(j + 1 / 2) * dd_y;
            const float pixelLen = dd_x.Length();
            Color colorSum = Black();
            size_t pix = (size_t)jj * rw + (i - x0);
            for (int s = 0; s < spp; s++) {
                rng.BeginSample(o.seed, (uint32_t)(j * W + i), (uint32_t)s);
                Vec3 target = pixelCenter;
                if (o.jitter) { // RandomPositionInPixel, Main.cpp:132-139 (raw rand(), double arithmetic)
                    const Vec3 ux = dd_x.GetNormalized(), uy = dd_y.GetNormalized();
                    // `unit_dx * (double expr) * len / 2`: the double narrows to float at the first Vec3*T
                    float fx = (float)(((double)rng.Rand() / (BHRT_RAND_MAX)) * 2 - 1);
                    target = target + ((ux * fx) * pixelLen) / 2.f;
                    float fy = (float)(((double)rng.Rand() / (BHRT_RAND_MAX)) * 2 - 1);
                    target = target + ((uy * fy) * pixelLen) / 2.f;
                }
                Ray ray;
                ray.p = camPos;
                ray.dir = target - camPos;
                bool bHit = false;
                HitInfo h;
                T.Closest(ray, h, bHit, BHRT_HIT_FRONT);
                Color c;
                if (bHit) c = sh.Shade(ray, h, o.internal_bounces, o.gi_bounces, 1);
                else c = X.Sample(S.H->background, Vec3((float)i / cam.width, (float)j / cam.height, 0.0f));
                colorSum += c;
                if (samples) { float *d = samples + (pix * spp + s) * 3; d[0] = c.r; d[1] = c.g; d[2] = c.b; }
            }
            Color out = colorSum / (float)spp; // Main.cpp:170
            if (radiance) { radiance[pix * 3] = out.r; radiance[pix * 3 + 1] = out.g; radiance[pix * 3 + 2] = out.b; }
            if (rgb8) { // gamma (Main.cpp:220-226) + Color24 (cyColor.h:271)
                const float inv = 1 / 2.2f;
                rgb8[pix * 3] = FloatToByte(M::Pow(out.r, inv));
                rgb8[pix * 3 + 1] = FloatToByte(M::Pow(out.g, inv));
                rgb8[pix * 3 + 2] = FloatToByte(M::Pow(out.b, inv));
            }
        }
        tot_closest += cnt.closest; tot_shadow += cnt.shadow; tot_shade += cnt.shade;
    }
    if (stats) {
        stats->closest_rays = tot_closest; stats->shadow_rays = tot_shadow; stats->shade_calls = tot_shade;
        stats->samples = (uint64_t)rw * rh * spp;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

template <class M> void PushAttrs(float *a, const HitInfo &h)
{
    a[0] = h.z; a[1] = h.p.x; a[2] = h.p.y; a[3] = h.p.z; a[4] = h.N.x; a[5] = h.N.y; a[6] = h.N.z;
    a[7] = h.uvw.x; a[8] = h.uvw.y; a[9] = h.uvw.z;
    a[10] = h.duvw[0].x; a[11] = h.duvw[0].y; a[12] = h.duvw[0].z; a[13] = h.duvw[1].x; a[14] = h.duvw[1].y; a[15] = h.duvw[1].z;
}

// Images beside the colour image: first hit of the un-jittered camera ray (Main.cpp:145,153), z = HitInfo::z (the store
// commented out at Main.cpp:231), world-space normal and diffuse.Sample(uvw, duvw) (MtlBlinn.cpp:393) of the hit material.
template <class M> int FirstHitT(const Scene &S, float *z, float *normal, float *albedo)
{
    const bhrt_camera &cam = S.H->camera;
    const Vec3 topLeft(cam.top_left[0], cam.top_left[1], cam.top_left[2]);
    const Vec3 dd_x = S.dd_x, dd_y = S.dd_y;
    const Vec3 camPos(cam.pos[0], cam.pos[1], cam.pos[2]);
    Textures<M> X(S);
    Tracer<M> T(S, nullptr);
    for (int j = 0; j < cam.height; j++)
        for (int i = 0; i < cam.width; i++) {
            const size_t pix = (size_t)j * cam.width + i;
            Ray ray;
            ray.p = camPos;
            ray.dir = (topLeft + (float)(i + 1 / 2) * dd_x - (float)(j + 1 / 2) * dd_y) - camPos;
            bool bHit = false;
            HitInfo h;
            T.Closest(ray, h, bHit, BHRT_HIT_FRONT);
            Vec3 N(0, 0, 0);
            Color kd = Black();
            if (bHit) {
                N = h.N;
                const int mi = S.nodes[h.node].material;
                if (mi >= 0 && S.materials[mi].kind == BHRT_MTL_BLINN) kd = X.Sample(S.materials[mi].diffuse, h.uvw, h.duvw);
                else if (mi >= 0 && S.materials[mi].kind == BHRT_MTL_WHITE) kd = Color(1, 1, 1);
            }
            if (z) z[pix] = bHit ? h.z : BHRT_BIGFLOAT;
            if (normal) { normal[pix * 3] = N.x; normal[pix * 3 + 1] = N.y; normal[pix * 3 + 2] = N.z; }
            if (albedo) { albedo[pix * 3] = kd.r; albedo[pix * 3 + 1] = kd.g; albedo[pix * 3 + 2] = kd.b; }
        }
    return 0;
}

} // namespace

// The note in GetRandomCrossingVector: `Vec3f(Rnd01(), Rnd01(), Rnd01())` has unspecified argument evaluation
// order.  g++ on x86-64 evaluates constructor arguments right to left, i.e. z first; see the fix-up below: the
// order actually used by the compiled reference is established empirically by tests/test_ref_parity.py
// (a plane normal (0,0,1) forces this path on every GI bounce) and encoded in ORACLE_RCV_ORDER.

extern "C" {

const char *oracle_last_error(void) { return g_err.c_str(); }

int oracle_trace_closest(const void *blob, const float *rays, int hit_side, size_t n, int32_t *node, int32_t *face, int32_t *front, float *attrs)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    Tracer<MathLibm> T(S, nullptr);
    for (size_t i = 0; i < n; i++) {
        Ray r;
        r.p = Vec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
        r.dir = Vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        HitInfo h;
        bool bHit = false;
        T.Closest(r, h, bHit, hit_side);
        if (node) node[i] = bHit ? h.node : -1;
        if (face) face[i] = bHit ? h.face : -1;
        if (front) front[i] = h.front ? 1 : 0;
        if (attrs) PushAttrs<MathLibm>(attrs + i * ORACLE_HIT_FLOATS, h);
    }
    return 0;
}

int oracle_trace_shadow(const void *blob, const float *rays, const float *tmax, size_t n, float *vis)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    Tracer<MathLibm> T(S, nullptr);
    for (size_t i = 0; i < n; i++) {
        Ray r;
        r.p = Vec3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
        r.dir = Vec3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        vis[i] = T.Shadow(r, tmax[i]);
    }
    return 0;
}

int oracle_render(const void *blob, const oracle_opts *opts, float *samples, float *radiance, uint8_t *rgb8, oracle_stats *stats)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    if (opts->math_mode == ORACLE_MATH_DEVICE) return RenderT<MathDevice>(S, *opts, samples, radiance, rgb8, stats);
    return RenderT<MathLibm>(S, *opts, samples, radiance, rgb8, stats);
}

int oracle_first_hit(const void *blob, int math_mode, float *z, float *normal, float *albedo)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    return math_mode == ORACLE_MATH_DEVICE ? FirstHitT<MathDevice>(S, z, normal, albedo) : FirstHitT<MathLibm>(S, z, normal, albedo);
}

// RenderImage::ComputeZBufferImage, scene.h:578-600
int oracle_zbuffer_image(const float *zbuffer, size_t size, uint8_t *zbufferImg)
{
    float zmin = BHRT_BIGFLOAT, zmax = 0;
    for (size_t i = 0; i < size; i++) {
        if (zbuffer[i] == BHRT_BIGFLOAT) continue;
        if (zmin > zbuffer[i]) zmin = zbuffer[i];
        if (zmax < zbuffer[i]) zmax = zbuffer[i];
    }
    for (size_t i = 0; i < size; i++) {
        if (zbuffer[i] == BHRT_BIGFLOAT) zbufferImg[i] = 0;
        else {
            float f = (zmax - zbuffer[i]) / (zmax - zmin);
            int c = int(f * 255);
            if (c < 0) c = 0;
            if (c > 255) c = 255;
            zbufferImg[i] = (uint8_t)c;
        }
    }
    return 0;
}

// colorArray of BeginRender, Main.cpp:219-229: pow(outColor, 1/2.2f) per channel, kept as floats
int oracle_color_image(const float *radiance, size_t n_floats, int gamma, int math_mode, float *color)
{
    const float inverseGama = 1 / 2.2f;
    for (size_t i = 0; i < n_floats; i++)
        color[i] = !gamma ? radiance[i] : (math_mode == ORACLE_MATH_DEVICE ? MathDevice::Pow(radiance[i], inverseGama) : MathLibm::Pow(radiance[i], inverseGama));
    return 0;
}

int oracle_math_eval(int fn, int math_mode, const float *a, const float *b, size_t n, float *out)
{
    for (size_t i = 0; i < n; i++) {
        float x = a[i], y = b ? b[i] : 0.f;
        if (math_mode == ORACLE_MATH_DEVICE) {
            switch (fn) {
            case 0: out[i] = MathDevice::Sin(x); break;
            case 1: out[i] = MathDevice::Cos(x); break;
            case 2: out[i] = MathDevice::Tan(x); break;
            case 3: out[i] = MathDevice::Acos(x); break;
            case 4: out[i] = MathDevice::Asin(x); break;
            case 5: out[i] = MathDevice::Atan2(x, y); break;
            case 6: out[i] = MathDevice::Pow(x, y); break;
            case 7: { uint32_t u; memcpy(&u, &x, 4); out[i] = bhrt::dm::rand_to_unit((int)u); } break;
            case 8: out[i] = x / y; break;
            case 9: out[i] = sqrtf(x); break;
            default: return 1;
            }
        } else {
            switch (fn) {
            case 0: out[i] = MathLibm::Sin(x); break;
            case 1: out[i] = MathLibm::Cos(x); break;
            case 2: out[i] = MathLibm::Tan(x); break;
            case 3: out[i] = MathLibm::Acos(x); break;
            case 4: out[i] = MathLibm::Asin(x); break;
            case 5: out[i] = MathLibm::Atan2(x, y); break;
            case 6: out[i] = MathLibm::Pow(x, y); break;
            default: return 1;
            }
        }
    }
    return 0;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// cyBVH build restated (DataStructure/cyBVH.h:122-142,242-328,356-375) — independent of the
// front-end's builder in bhraytracer_amd/csrc/scene_host.cpp; tests compare the two node for node.
// ------------------------------------------------------------------------------------------------
namespace {
struct OBox {
    float b[6];
    OBox() { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }
    void operator+=(const OBox &o)
    {
        for (int i = 0; i < 3; i++) {
            if (b[i] > o.b[i]) b[i] = o.b[i];
            if (b[i + 3] < o.b[i + 3]) b[i + 3] = o.b[i + 3];
        }
    }
};
struct OTemp {
    OTemp *c1 = nullptr, *c2 = nullptr;
    OBox box;
    unsigned count, offset;
    OTemp(unsigned c, unsigned o, const OBox &b) : box(b), count(c), offset(o) {}
    ~OTemp() { delete c1; delete c2; }
    unsigned NumNodes() const { return 1 + (c1 ? c1->NumNodes() : 0) + (c2 ? c2->NumNodes() : 0); }
};
struct OBuilder {
    const float *v;
    const uint32_t *f;
    std::vector<uint32_t> elements;
    unsigned maxPer;
    OBox Bounds(unsigned i) const // cyBVH.h:356-368
    {
        OBox x;
        const float *p = v + 3 * f[i * 3];
        x.b[0] = x.b[3] = p[0]; x.b[1] = x.b[4] = p[1]; x.b[2] = x.b[5] = p[2];
        for (int j = 1; j < 3; j++) {
            const float *q = v + 3 * f[i * 3 + j];
            for (int k = 0; k < 3; k++) {
                if (x.b[k] > q[k]) x.b[k] = q[k];
                if (x.b[k + 3] < q[k]) x.b[k + 3] = q[k];
            }
        }
        return x;
    }
    float Center(unsigned i, int d) const { return (v[3 * f[i * 3] + d] + v[3 * f[i * 3 + 1] + d] + v[3 * f[i * 3 + 2] + d]) / 3.0f; }
    unsigned MeanSplit(unsigned n, uint32_t *e, const float *box) const
    {
        if (n <= maxPer) return 0;
        float d[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
        unsigned sd[3];
        sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
        sd[1] = (sd[0] + 1) % 3;
        sd[2] = (sd[0] + 2) % 3;
        if (d[sd[1]] < d[sd[2]]) { unsigned t = sd[1]; sd[1] = sd[2]; sd[2] = t; }
        unsigned c1n = 0;
        for (int s = 0; s < 3; s++) {
            unsigned dim = sd[s];
            float splitPos = 0.5f * (box[dim] + box[dim + 3]);
            unsigned i = 0, j = n;
            while (i < j) {
                float c = Center(e[i], dim);
                if (c <= splitPos) i++;
                else { j--; uint32_t t = e[i]; e[i] = e[j]; e[j] = t; }
            }
            if (i < n && i > 0) { c1n = i; break; }
        }
        return c1n;
    }
    void Split(OTemp *t)
    {
        uint32_t *e = &elements[t->offset];
        unsigned c1n = MeanSplit(t->count, e, t->box.b);
        if (c1n == 0 || c1n >= t->count) {
            if (t->count > 8) c1n = t->count / 2;
            else return;
        }
        OBox b1, b2;
        for (unsigned i = 0; i < c1n; i++) b1 += Bounds(e[i]);
        for (unsigned i = c1n; i < t->count; i++) b2 += Bounds(e[i]);
        t->c1 = new OTemp(c1n, t->offset, b1);
        t->c2 = new OTemp(t->count - c1n, t->offset + c1n, b2);
        Split(t->c1);
        Split(t->c2);
    }
    unsigned Convert(uint32_t *out, unsigned id, OTemp *t, unsigned childIndex, unsigned parent)
    {
        uint32_t *o = out + 8 * (size_t)id;
        memcpy(o, t->box.b, 24);
        o[7] = parent;
        if (!t->c1) { o[6] = (t->offset & 0x0fffffffu) | ((t->count - 1) << 28) | 0x80000000u; return childIndex; }
        o[6] = childIndex & 0x7fffffffu;
        unsigned next = Convert(out, childIndex, t->c1, childIndex + 2, id);
        return Convert(out, childIndex + 1, t->c2, next, id);
    }
};
} // namespace

extern "C" int oracle_bvh_build(const float *v, const uint32_t *f, uint32_t nf, uint32_t max_per_leaf, uint32_t *nodes_out, size_t nodes_cap, uint32_t *elems_out)
{
    if (nf == 0) return 0;
    OBuilder B;
    B.v = v; B.f = f; B.maxPer = max_per_leaf > 8 ? 8 : max_per_leaf;
    B.elements.resize(nf);
    for (uint32_t i = 0; i < nf; i++) B.elements[i] = i;
    OBox box;
    for (uint32_t i = 0; i < nf; i++) { OBox b = B.Bounds(i); box += b; }
    OTemp *root = new OTemp(nf, 0, box);
    B.Split(root);
    unsigned n = root->NumNodes();
    if ((size_t)n + 1 > nodes_cap) { delete root; g_err = "nodes_out too small"; return -1; }
    memset(nodes_out, 0, 32);
    B.Convert(nodes_out, 1, root, 2, 0);
    delete root;
    memcpy(elems_out, B.elements.data(), (size_t)nf * 4);
    return (int)n + 1;
}

// ------------------------------------------------------------------------------------------------
// caustic photon map: DataStructure/cyPhotonMap.h, Main.cpp:319-386, MtlBlinn.cpp:203-303, PointLight.cpp:20-34
// ------------------------------------------------------------------------------------------------
namespace {

#pragma pack(push, 1)
struct Photon { // cyPhotonMap.h:72-90, 24 bytes (= the record of Resource/causticPhotonMap.dat and of PhotonMapViz.cpp:30-36)
    float pos[3];
    float power;
    uint8_t color[3];
    uint8_t planeAndDirZ;
    int16_t dirX, dirY;
};
#pragma pack(pop)
static_assert(sizeof(Photon) == 24, "photon record must be 24 bytes");

struct PhotonMapView {
    std::vector<Photon> photons; // slot 0 unused (zero), photons 1..n
    int numStored = 0, halfStored = 0;
    std::vector<Photon> unbalanced;
};
PhotonMapView g_pm_storage;

inline void PhotonSetPower(Photon &p, const Color &c) // cyPhotonMap.h:172-178
{
    p.power = c.r;
    if (p.power < c.g) p.power = c.g;
    if (p.power < c.b) p.power = c.b;
    Color q = c / p.power;
    p.color[0] = FloatToByte(q.r); p.color[1] = FloatToByte(q.g); p.color[2] = FloatToByte(q.b);
}
inline void PhotonSetDirection(Photon &p, const Vec3 &dir) // cyPhotonMap.h:180-190
{
    p.dirX = (int16_t)(dir.x * 0x7FFF);
    p.dirY = (int16_t)(dir.y * 0x7FFF);
    if (dir.z > 0) p.planeAndDirZ &= 0x7;
    else p.planeAndDirZ = 0x8 | (p.planeAndDirZ & 0x7);
}
inline Vec3 PhotonGetDirection(const Photon &p) // cyPhotonMap.h:192-214 incl. the missing dirY^2
{
    Vec3 dir;
    dir.x = float(p.dirX) / float(0x7FFF);
    dir.y = float(p.dirY) / float(0x7FFF);
    int dirXY2 = p.dirX * p.dirX + p.dirY - p.dirY;
    if (dirXY2 > 0x3FFF0001) dirXY2 = 0x3FFF0001;
    int dirZ2 = 0x3FFF0001 - dirXY2;
    int dirZ = 0, place = 0x40000000, remainder = dirZ2;
    while (place > remainder) place = place >> 2;
    while (place) {
        if (remainder >= dirZ + place) { remainder = remainder - dirZ - place; dirZ = dirZ + (place << 1); }
        dirZ = dirZ >> 1;
        place = place >> 2;
    }
    dir.z = float(dirZ) / float(0x7FFF);
    if (p.planeAndDirZ & 0x8) dir.z = -dir.z;
    return dir;
}
inline Color PhotonGetPower(const Photon &p) { return Color(p.color[0] / 255.0f, p.color[1] / 255.0f, p.color[2] / 255.0f) * p.power; } // cyColor ToColor * power

// PhotonMap::BalanceSegment, cyPhotonMap.h:262-328
void BalanceSegment(std::vector<Photon> &photons, std::vector<Photon> &balanced, const Vec3 &boxMin, const Vec3 &boxMax, int index, int start, int end)
{
    int median = 1;
    while ((4 * median) <= (end - start + 1)) median += median;
    if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
    else median = end - median + 1;
    int axis = 2;
    Vec3 boxDif = boxMax - boxMin;
    if (boxDif.x > boxDif.y) { if (boxDif.x > boxDif.z) axis = 0; }
    else if (boxDif.y > boxDif.z) axis = 1;
    int left = start, right = end;
    auto swapP = [&](int i, int j) { Photon t = photons[i]; photons[i] = photons[j]; photons[j] = t; };
    while (right > left) {
        const float v = photons[right].pos[axis];
        int i = left - 1, j = right;
        while (photons[++i].pos[axis] < v) {}
        while (photons[--j].pos[axis] > v && j > left) {}
        while (i < j) {
            swapP(i, j);
            while (photons[++i].pos[axis] < v) {}
            while (photons[--j].pos[axis] > v && j > left) {}
        }
        swapP(i, right);
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
    balanced[index] = photons[median];
    balanced[index].planeAndDirZ = (balanced[index].planeAndDirZ & 0x8) | (axis & 0x3); // SetPlane
    if (median > start) {
        if (start < median - 1) {
            Vec3 tMax = boxMax;
            (&tMax.x)[axis] = balanced[index].pos[axis];
            BalanceSegment(photons, balanced, boxMin, tMax, 2 * index, start, median - 1);
        } else balanced[2 * index] = photons[start];
    }
    if (median < end) {
        if (median + 1 < end) {
            Vec3 tMin = boxMin;
            (&tMin.x)[axis] = balanced[index].pos[axis];
            BalanceSegment(photons, balanced, tMin, boxMax, 2 * index + 1, median + 1, end);
        } else balanced[2 * index + 1] = photons[end];
    }
}
// PhotonMap::PrepareForIrradianceEstimation, cyPhotonMap.h:236-258 (bbox starts from the unused slot 0, Q11)
void PreparePhotonMap(PhotonMapView &pm)
{
    if (pm.photons.empty() || pm.numStored == 0) return;
    Vec3 boxMin(pm.photons[0].pos[0], pm.photons[0].pos[1], pm.photons[0].pos[2]), boxMax = boxMin;
    for (int i = 1; i <= pm.numStored; i++) {
        const float *q = pm.photons[i].pos;
        if (boxMin.x > q[0]) boxMin.x = q[0];
        if (boxMax.x < q[0]) boxMax.x = q[0];
        if (boxMin.y > q[1]) boxMin.y = q[1];
        if (boxMax.y < q[1]) boxMax.y = q[1];
        if (boxMin.z > q[2]) boxMin.z = q[2];
        if (boxMax.z < q[2]) boxMax.z = q[2];
    }
    std::vector<Photon> balanced(pm.numStored + 1);
    memset(balanced.data(), 0, sizeof(Photon) * balanced.size());
    BalanceSegment(pm.photons, balanced, boxMin, boxMax, 1, 1, pm.numStored);
    balanced.swap(pm.photons);
    pm.halfStored = pm.numStored / 2 - 1;
}

// PhotonMap::LocatePhotons, cyPhotonMap.h:421-498 (normScale == 0: ellipticity 1 as called at MtlBlinn.cpp:334)
struct Nearest {
    Vec3 pos, normal;
    int maxPhotons, found;
    float *dist2;
    int *idx;
};
void LocatePhotons(const PhotonMapView &pm, Nearest &np, int index)
{
    const Photon &p = pm.photons[index];
    int axis = p.planeAndDirZ & 0x3;
    if (index < pm.halfStored) {
        float dist = (&np.pos.x)[axis] - p.pos[axis];
        if (dist > 0) {
            LocatePhotons(pm, np, 2 * index + 1);
            if (dist * dist < np.dist2[0]) LocatePhotons(pm, np, 2 * index);
        } else {
            LocatePhotons(pm, np, 2 * index);
            if (dist * dist < np.dist2[0]) LocatePhotons(pm, np, 2 * index + 1);
        }
    }
    Vec3 dif = Vec3(p.pos[0], p.pos[1], p.pos[2]) - np.pos;
    float dist2 = dif.LengthSquared();
    if (dist2 < np.dist2[0]) {
        Vec3 dir = PhotonGetDirection(p);
        if (dir.Dot(np.normal) >= 0) return;
        if (np.found < np.maxPhotons) {
            np.found++;
            np.dist2[np.found] = dist2;
            np.idx[np.found] = index;
            if (np.found == np.maxPhotons) { // build a max-heap
                int half_found = np.found >> 1;
                for (int k = half_found; k >= 1; k--) {
                    int parent = k;
                    int tp = np.idx[k];
                    float td2 = np.dist2[k];
                    while (parent <= half_found) {
                        int j = parent + parent;
                        if (j < np.found && np.dist2[j] < np.dist2[j + 1]) j++;
                        if (td2 >= np.dist2[j]) break;
                        np.dist2[parent] = np.dist2[j];
                        np.idx[parent] = np.idx[j];
                        parent = j;
                    }
                    np.idx[parent] = tp;
                    np.dist2[parent] = td2;
                }
            }
        } else {
            int parent = 1, j = 2;
            while (j <= np.found) {
                if (j < np.found && np.dist2[j] < np.dist2[j + 1]) j++;
                if (dist2 > np.dist2[j]) break;
                np.dist2[parent] = np.dist2[j];
                np.idx[parent] = np.idx[j];
                parent = j;
                j <<= 1;
            }
            np.idx[parent] = index;
            np.dist2[parent] = dist2;
            np.dist2[0] = np.dist2[1];
        }
    }
}
// PhotonMap::EstimateIrradiance<1000>(irrad, direction, radius, pos, &normal), cyPhotonMap.h:332-382 (constant filter)
bool PhotonEstimate(const PhotonMapView *pm, Color &irrad, Vec3 &direction, float radius, const Vec3 &pos, const Vec3 &normal)
{
    irrad = Black();
    direction = Vec3(0, 0, 0);
    if (!pm || pm->numStored == 0) return false;
    const int maxPhotons = 1000; // MAX_PhotonCountInArea, MtlBlinn.cpp:28
    float found_dist2[maxPhotons + 1];
    int found_idx[maxPhotons + 1];
    Nearest np;
    np.pos = pos; np.normal = normal; np.maxPhotons = maxPhotons; np.found = 0; np.dist2 = found_dist2; np.idx = found_idx;
    np.dist2[0] = radius * radius;
    LocatePhotons(*pm, np, 1);
    for (int i = 1; i <= np.found; i++) {
        const Photon &ph = pm->photons[np.idx[i]];
        Color power = PhotonGetPower(ph);
        float filter = 1;
        irrad += filter * power;
        Vec3 dir = PhotonGetDirection(ph);
        direction = direction + dir * (filter * ph.power);
    }
    if (np.found > 0) {
        float area = (float)M_PI * np.dist2[0];
        if (area > 0) {
            const float one_over_area = 1.0f / area;
            irrad = irrad * one_over_area;
        }
        direction = direction / direction.Length(); // Normalize()
    }
    return np.found > 0;
}

// photon emission -------------------------------------------------------------------------------
template <class M> struct PhotonTracer {
    const Scene &S;
    Tracer<M> T;
    Textures<M> X;
    Rng rng;
    Shader<M> sh;
    PhotonMapView &pm;
    int maxPhotons;
    bool global_map = false; // false: caustic map (BuildCausticPhotonMap); true: global map (BuildPhotonMap, Main.cpp:251-317)
    PhotonTracer(const Scene &s, PhotonMapView &p, int maxp) : S(s), T(s, nullptr), X(s), sh(s, T, X, rng, nullptr, false), pm(p), maxPhotons(maxp) {}

    bool AddPhoton(const Vec3 &pos, const Vec3 &dir, const Color &power) // cyPhotonMap.h:218-232
    {
        if (pm.numStored >= maxPhotons) return false;
        int i = ++pm.numStored;
        Photon p;
        memset(&p, 0, sizeof p); // the reference leaves planeAndDirZ's low bits uninitialised; zero here
        p.pos[0] = pos.x; p.pos[1] = pos.y; p.pos[2] = pos.z;
        PhotonSetDirection(p, dir);
        PhotonSetPower(p, power);
        pm.photons[i] = p;
        return true;
    }
    // MtlBlinn::RandomPhotonBounceForCaustic, MtlBlinn.cpp:203-303
    bool BounceForCaustic(const bhrt_material &m, Ray &r, Color &c, const HitInfo &hInfo)
    {
        float rnd = rng.Rnd01();
        Vec3 vN = hInfo.N.GetNormalized();
        Vec3 vV = -(r.dir.GetNormalized());
        Color refrC(m.refraction.color[0], m.refraction.color[1], m.refraction.color[2]);
        if (refrC.Gray() > 0) {
            float cosPhi1 = vN.Dot(vV);
            float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
            float sinPhi2 = sinPhi1 / m.ior;
            float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
            Vec3 vTn = -cosPhi2 * vN;
            Vec3 vNxV = vN.Cross(vV);
            Vec3 vTp = vN.Cross(vNxV).GetNormalized() * sinPhi2;
            Vec3 vT = vTn + vTp;
            Ray in;
            in.dir = vT;
            in.p = hInfo.p - vN * O_BIAS;
            HitInfo h;
            bool bHit = false;
            T.Closest(in, h, bHit, BHRT_HIT_BACK);
            if (bHit && h.node >= 0) {
                bool out;
                Ray next = sh.HandleRayWhenRefractionRayOut(in, h, m.ior, out, m.refraction_glossiness);
                if (out) { r = next; return true; }
                return false;
            }
            return false;
        }
        if (rnd < 0.3f) return false; // Photon_AbsorbChance, MtlBlinn.cpp:27
        float diffuseTheta = 0;
        Vec3 diffuseRayDir = sh.GetSampleInSemiSphere(vN, diffuseTheta).GetNormalized();
        (void)diffuseRayDir;
        float p_diffuseTheta = M::Sin(2 * diffuseTheta);
        float specularTheta = 0;
        float cosvVvN = vN.Dot(vV);
        Vec3 vR = 2 * cosvVvN * vN - vV;
        Vec3 specRayDir = sh.GetSampleAlongLightDirection(vR, m.glossiness, specularTheta);
        float p_specularTheta = M::Pow(M::Cos(specularTheta), m.glossiness);
        float P_Diffuse = Shader<M>::GetK(m.diffuse) * p_diffuseTheta;
        float P_sum = P_Diffuse + Shader<M>::GetK(m.specular) * p_specularTheta;
        float p_Diff = (P_Diffuse / P_sum) * (1 - 0.3f) + 0.3f;
        float p_Spec = (1 - p_Diff) * (1 - 0.3f) + 0.3f;
        bool useSpecular = rnd >= p_Diff;
        if (!useSpecular) return false;
        Color ksf = Color(m.specular.color[0], m.specular.color[1], m.specular.color[2]) / p_Spec;
        c = c * ksf;
        r.dir = specRayDir;
        r.p = hInfo.p + hInfo.N * O_BIAS;
        return true;
    }
    // MtlBlinn::RandomPhotonBounce, MtlBlinn.cpp:140-202 (global map): a transmissive surface ends the photon, otherwise
    // absorb / diffuse / specular roulette with the power rescaled by the chosen lobe's probability
    bool Bounce(const bhrt_material &m, Ray &r, Color &c, const HitInfo &hInfo)
    {
        float rnd = rng.Rnd01();
        Vec3 vN = hInfo.N.GetNormalized();
        Vec3 vV = -(r.dir.GetNormalized());
        Color refrC(m.refraction.color[0], m.refraction.color[1], m.refraction.color[2]);
        if (refrC.Gray() > 0) return false;
        if (rnd < 0.3f) return false; // Photon_AbsorbChance, MtlBlinn.cpp:27
        float diffuseTheta = 0;
        Vec3 diffuseRayDir = sh.GetSampleInSemiSphere(vN, diffuseTheta).GetNormalized();
        float p_diffuseTheta = M::Sin(2 * diffuseTheta);
        float specularTheta = 0;
        float cosvVvN = vN.Dot(vV);
        Vec3 vR = 2 * cosvVvN * vN - vV;
        Vec3 specRayDir = sh.GetSampleAlongLightDirection(vR, m.glossiness, specularTheta);
        float p_specularTheta = M::Pow(M::Cos(specularTheta), m.glossiness);
        float P_Diffuse = Shader<M>::GetK(m.diffuse) * p_diffuseTheta;
        float P_sum = P_Diffuse + Shader<M>::GetK(m.specular) * p_specularTheta;
        float p_Diff = (P_Diffuse / P_sum) * (1 - 0.3f) + 0.3f;
        float p_Spec = (1 - p_Diff) * (1 - 0.3f) + 0.3f;
        bool useSpecular = rnd >= p_Diff;
        Color kdf = Color(m.diffuse.color[0], m.diffuse.color[1], m.diffuse.color[2]) / p_Diff;
        Color ksf = Color(m.specular.color[0], m.specular.color[1], m.specular.color[2]) / p_Spec;
        c = c * (useSpecular ? ksf : kdf);
        r.dir = useSpecular ? specRayDir : diffuseRayDir;
        r.p = hInfo.p + hInfo.N * O_BIAS;
        return true;
    }
    // TraceCausticPhotonRay / TracePhotonRay, Main.cpp:296-340 (tail recursion -> loop)
    void TracePhoton(Ray ray, Color intensity)
    {
        bool first = true;
        for (int guard = 0; guard < O_MAXLOOP; guard++) {
            bool bHit = false;
            HitInfo h;
            T.Closest(ray, h, bHit, BHRT_HIT_FRONT);
            if (!bHit) return;
            int mi = S.nodes[h.node].material;
            if (mi < 0) return; // null material: the reference dereferences null
            const bhrt_material &m = S.materials[mi];
            const bool photonSurface = Color(m.diffuse.color[0], m.diffuse.color[1], m.diffuse.color[2]).Gray() > 0; // materials.h:47
            if (!first && photonSurface) AddPhoton(h.p, ray.dir.GetNormalized(), intensity);
            if (m.kind != BHRT_MTL_BLINN) return; // MultiMtl has no caustic bounce (scene.h:298)
            Ray nr = ray;
            Color ni = intensity;
            if (!(global_map ? Bounce(m, nr, ni, h) : BounceForCaustic(m, nr, ni, h))) return;
            ray = nr;
            intensity = ni;
            first = false;
        }
    }
    // PointLight::RandomPhoton, PointLight.cpp:20-34
    Ray RandomPhoton(const bhrt_light &l)
    {
        float phi = (float)(rng.Rnd01() * 2 * O_PI);
        float theta = M::Acos(MinF(1.f, MaxF(-1.f, 1 - 2 * rng.Rnd01())));
        Vec3 axisZ(0, 0, 1), axisX(1, 0, 0), axisY(0, 1, 0);
        Ray ray;
        ray.dir = M::Sin(theta) * (axisX * M::Cos(phi) + axisY * M::Sin(phi)) + axisZ * M::Cos(theta);
        ray.p = Vec3(l.vec[0], l.vec[1], l.vec[2]);
        return ray;
    }
    // BuildCausticPhotonMap, Main.cpp:342-386
    uint64_t Build(uint32_t seed, bool keyed)
    {
        std::vector<const bhrt_light *> pl;
        for (uint32_t i = 0; i < S.H->n_lights; i++)
            if (S.lights[i].type == BHRT_LIGHT_POINT) pl.push_back(&S.lights[i]); // the reference reinterpret_casts EVERY light (Q19)
        if (pl.empty()) return 0;
        auto key = [](const bhrt_light *l) { return Color(l->intensity[0], l->intensity[1], l->intensity[2]).Gray() * (int)l->size; };
        std::sort(pl.begin(), pl.end(), [&](const bhrt_light *a, const bhrt_light *b) { return key(a) < key(b); });
        float sum = 0;
        for (auto *l : pl) sum += key(l);
        uint64_t emitted = 0;
        rng.keyed = false;
        rng.device_math = std::is_same<M, MathDevice>::value;
        rng.key = bhrt_photon_key_sequential(seed);
        rng.ctr = 0;
        while (pm.numStored < maxPhotons) {
            if (keyed) { bhrt_photon_stream(seed, emitted, &rng.key, &rng.ctr); rng.wrap = BHRT_PHOTON_WINDOW_MASK; }
            float rnd = rng.Rnd01();
            size_t i = 0;
            while (rnd > Color(pl[i]->intensity[0], pl[i]->intensity[1], pl[i]->intensity[2]).Gray() * pl[i]->size / sum && i < pl.size() - 1) i++;
            Ray ray = RandomPhoton(*pl[i]);
            TracePhoton(ray, Color(pl[i]->intensity[0], pl[i]->intensity[1], pl[i]->intensity[2]));
            emitted++;
            if (emitted > (uint64_t)maxPhotons * 4096ull + (1ull << 24)) break; // no photon surface reachable: give up (the reference loops forever)
        }
        return emitted;
    }
};

} // namespace

extern "C" int oracle_photon_build(const void *blob, const oracle_opts *opts, uint32_t max_photons, void *photons_out, uint32_t *n_stored, uint64_t *n_emitted)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    PhotonMapView &pm = g_pm_storage;
    pm.photons.assign((size_t)max_photons + 1, Photon());
    memset(pm.photons.data(), 0, sizeof(Photon) * pm.photons.size());
    pm.numStored = 0;
    uint64_t emitted;
    if (opts->math_mode == ORACLE_MATH_DEVICE) { PhotonTracer<MathDevice> t(S, pm, (int)max_photons); emitted = t.Build(opts->seed, opts->rng_mode == ORACLE_RNG_KEYED); }
    else { PhotonTracer<MathLibm> t(S, pm, (int)max_photons); emitted = t.Build(opts->seed, opts->rng_mode == ORACLE_RNG_KEYED); }
    if (pm.numStored > 0) {
        const float scale = 1.f / pm.numStored; // ScalePhotonPowers, Main.cpp:380
        for (int i = 1; i <= pm.numStored; i++) pm.photons[i].power *= scale;
    }
    pm.photons.resize((size_t)pm.numStored + 1);
    pm.unbalanced.assign(pm.photons.begin() + 1, pm.photons.end());
    PreparePhotonMap(pm);
    if (photons_out && pm.numStored) memcpy(photons_out, &pm.photons[1], sizeof(Photon) * pm.numStored);
    if (n_stored) *n_stored = (uint32_t)pm.numStored;
    if (n_emitted) *n_emitted = emitted;
    g_photon_map = &pm;
    return 0;
}

// BuildPhotonMap (global map; nothing in the reference gathers from it, so it is not attached): balanced records and the
// records in emission order (after ScalePhotonPowers) out
extern "C" int oracle_photon_build_global(const void *blob, const oracle_opts *opts, uint32_t max_photons, void *photons_out, void *emitted_out,
                                          uint32_t *n_stored, uint64_t *n_emitted)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    static PhotonMapView gm;
    gm = PhotonMapView();
    gm.photons.assign((size_t)max_photons + 1, Photon());
    memset(gm.photons.data(), 0, sizeof(Photon) * gm.photons.size());
    gm.numStored = 0;
    uint64_t emitted = 0;
    if (opts->math_mode == ORACLE_MATH_DEVICE) { PhotonTracer<MathDevice> t(S, gm, (int)max_photons); t.global_map = true; emitted = t.Build(opts->seed, opts->rng_mode == ORACLE_RNG_KEYED); }
    else { PhotonTracer<MathLibm> t(S, gm, (int)max_photons); t.global_map = true; emitted = t.Build(opts->seed, opts->rng_mode == ORACLE_RNG_KEYED); }
    if (gm.numStored > 0) {
        const float scale = 1.f / gm.numStored;
        for (int i = 1; i <= gm.numStored; i++) gm.photons[i].power *= scale;
    }
    gm.photons.resize((size_t)gm.numStored + 1);
    if (emitted_out && gm.numStored) memcpy(emitted_out, &gm.photons[1], sizeof(Photon) * gm.numStored);
    PreparePhotonMap(gm);
    if (photons_out && gm.numStored) memcpy(photons_out, &gm.photons[1], sizeof(Photon) * gm.numStored);
    if (n_stored) *n_stored = (uint32_t)gm.numStored;
    if (n_emitted) *n_emitted = emitted;
    return 0;
}

extern "C" int oracle_photon_unbalanced(void *out)
{
    if (!out) return 1;
    if (!g_pm_storage.unbalanced.empty()) memcpy(out, g_pm_storage.unbalanced.data(), sizeof(Photon) * g_pm_storage.unbalanced.size());
    return 0;
}

extern "C" int oracle_photon_attach(const void *photons, uint32_t n)
{
    PhotonMapView &pm = g_pm_storage;
    pm.photons.assign((size_t)n + 1, Photon());
    memset(pm.photons.data(), 0, sizeof(Photon));
    if (n) memcpy(&pm.photons[1], photons, sizeof(Photon) * n);
    pm.numStored = (int)n;
    pm.halfStored = pm.numStored / 2 - 1;
    g_photon_map = n ? &pm : nullptr;
    return 0;
}

extern "C" int oracle_photon_balance(const void *emitted, uint32_t n, void *balanced_out)
{
    PhotonMapView pm;
    pm.photons.assign((size_t)n + 1, Photon());
    memset(pm.photons.data(), 0, sizeof(Photon));
    if (n) memcpy(&pm.photons[1], emitted, sizeof(Photon) * n);
    pm.numStored = (int)n;
    PreparePhotonMap(pm);
    if (n) memcpy(balanced_out, &pm.photons[1], sizeof(Photon) * n);
    return 0;
}

extern "C" int oracle_photon_gather(const float *p, const float *nrm, size_t cnt, float radius, float *irrad, float *dir)
{
    if (!g_photon_map) { g_err = "no photon map attached"; return 1; }
    for (size_t i = 0; i < cnt; i++) {
        Color c;
        Vec3 d;
        PhotonEstimate(g_photon_map, c, d, radius, Vec3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]), Vec3(nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2]));
        irrad[i * 3] = c.r; irrad[i * 3 + 1] = c.g; irrad[i * 3 + 2] = c.b;
        dir[i * 3] = d.x; dir[i * 3 + 1] = d.y; dir[i * 3 + 2] = d.z;
    }
    return 0;
}

// The list LocatePhotons leaves behind (the photons the estimate sums, as indices into the balanced map, ascending), how many, and
// np.dist2[0] at the end — what the HIP path's selection pass has to arrive at as a SET (tests/test_photon.py).
extern "C" int oracle_photon_knn(const float *p, const float *nrm, size_t cnt, float radius, uint32_t *idx /* cnt x 1000 */, uint32_t *count, float *d2max)
{
    if (!g_photon_map) { g_err = "no photon map attached"; return 1; }
    const int maxPhotons = 1000;
    for (size_t i = 0; i < cnt; i++) {
        float found_dist2[maxPhotons + 1];
        int found_idx[maxPhotons + 1];
        Nearest np;
        np.pos = Vec3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); np.normal = Vec3(nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2]);
        np.maxPhotons = maxPhotons; np.found = 0; np.dist2 = found_dist2; np.idx = found_idx;
        np.dist2[0] = radius * radius;
        if (g_photon_map->numStored > 0) LocatePhotons(*g_photon_map, np, 1);
        std::sort(found_idx + 1, found_idx + 1 + np.found);
        for (int k = 0; k < maxPhotons; k++) idx[i * maxPhotons + k] = k < np.found ? (uint32_t)found_idx[k + 1] : 0u;
        count[i] = (uint32_t)np.found;
        d2max[i] = np.dist2[0];
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BeginRender() as a whole program (Main.cpp:178-242), the way the reference itself runs it with one OpenMP thread: ONE
// rand() stream for the process (libc's global state), consumed first by BuildCausticPhotonMap (Main.cpp:195-198, -DUSE_PhotonMap
// builds) and then by the pixel loop in ITS order — `for i in [0, W)` outside, `for j in [0, H)` inside (Main.cpp:204-211),
// PT_SampleCount samples per pixel, no reset anywhere.  Pinned against the reference's real BeginRender() / BuildCausticPhotonMap()
// (oracle/ref_harness `beginrender`, tests/golden/begin_render.npz): this is the check that the per-(pixel, sample) form of
// RenderT above restates the frame loop faithfully (camera frame, jitter, sample average, gamma, Color24, emission loop header).
namespace {
template <class M> int BeginRenderT(const Scene &S, const oracle_opts &o, uint32_t photon_budget, uint8_t *rgb8, void *photons_out, uint32_t *n_stored,
                                    uint64_t *n_emitted, uint64_t *draws)
{
    const bhrt_camera &cam = S.H->camera;
    const int W = cam.width, Hh = cam.height, spp = o.spp > 0 ? o.spp : 32;
    uint32_t ctr = 0;
    const uint32_t key = bhrt_photon_key_sequential(o.seed); // the one stream (ref_harness: g_key)
    if (photon_budget) {
        PhotonMapView &pm = g_pm_storage;
        pm.photons.assign((size_t)photon_budget + 1, Photon());
        memset(pm.photons.data(), 0, sizeof(Photon) * pm.photons.size());
        pm.numStored = 0;
        PhotonTracer<M> t(S, pm, (int)photon_budget);
        const uint64_t emitted = t.Build(o.seed, false);
        ctr = t.rng.ctr;
        if (pm.numStored > 0) {
            const float scale = 1.f / pm.numStored; // ScalePhotonPowers, Main.cpp:380
            for (int i = 1; i <= pm.numStored; i++) pm.photons[i].power *= scale;
        }
        pm.photons.resize((size_t)pm.numStored + 1);
        pm.unbalanced.assign(pm.photons.begin() + 1, pm.photons.end());
        PreparePhotonMap(pm);
        if (photons_out && pm.numStored) memcpy(photons_out, &pm.photons[1], sizeof(Photon) * pm.numStored);
        if (n_stored) *n_stored = (uint32_t)pm.numStored;
        if (n_emitted) *n_emitted = emitted;
        g_photon_map = &pm;
    }
    const Vec3 topLeft(cam.top_left[0], cam.top_left[1], cam.top_left[2]);
    const Vec3 dd_x = S.dd_x, dd_y = S.dd_y;
    const Vec3 camPos(cam.pos[0], cam.pos[1], cam.pos[2]);
    Textures<M> X(S);
    Counters cnt;
    Tracer<M> T(S, &cnt);
    Rng rng;
    rng.keyed = false;
    rng.device_math = std::is_same<M, MathDevice>::value;
    rng.key = key;
    rng.ctr = ctr;
    Shader<M> sh(S, T, X, rng, &cnt, photon_budget != 0);
    for (int i = 0; i < W; i++)
        for (int j = 0; j < Hh; j++) {
            Vec3 pixelCenter = topLeft + (float)(i + 1 / 2) * dd_x - (float)(j + 1 / 2) * dd_y; // Main.cpp:145
            const float pixelLen = dd_x.Length();
            Color colorSum = Black();
            for (int s = 0; s < spp; s++) {
                Vec3 target = pixelCenter; // RandomPositionInPixel, Main.cpp:132-139
                const Vec3 ux = dd_x.GetNormalized(), uy = dd_y.GetNormalized();
                float fx = (float)(((double)rng.Rand() / (BHRT_RAND_MAX)) * 2 - 1);
                target = target + ((ux * fx) * pixelLen) / 2.f;
                float fy = (float)(((double)rng.Rand() / (BHRT_RAND_MAX)) * 2 - 1);
                target = target + ((uy * fy) * pixelLen) / 2.f;
                Ray ray;
                ray.p = camPos;
                ray.dir = target - camPos;
                bool bHit = false;
                HitInfo h;
                T.Closest(ray, h, bHit, BHRT_HIT_FRONT);
                if (bHit) colorSum += sh.Shade(ray, h, o.internal_bounces, o.gi_bounces, 1);
                else colorSum += X.Sample(S.H->background, Vec3((float)i / cam.width, (float)j / cam.height, 0.0f));
            }
            const Color out = colorSum / (float)spp; // Main.cpp:170
            const float inv = 1 / 2.2f;              // Main.cpp:220-230
            const size_t pix = (size_t)j * W + i;
            rgb8[pix * 3] = FloatToByte(M::Pow(out.r, inv));
            rgb8[pix * 3 + 1] = FloatToByte(M::Pow(out.g, inv));
            rgb8[pix * 3 + 2] = FloatToByte(M::Pow(out.b, inv));
        }
    if (draws) *draws = rng.ctr;
    return 0;
}
} // namespace

extern "C" int oracle_begin_render(const void *blob, const oracle_opts *opts, uint32_t photon_budget, uint8_t *rgb8, void *photons_out, uint32_t *n_stored,
                                   uint64_t *n_emitted, uint64_t *draws)
{
    Scene S;
    if (!S.Init(blob)) return 1;
    if (!opts || !rgb8) { g_err = "null argument"; return 2; }
    if (opts->math_mode == ORACLE_MATH_DEVICE) return BeginRenderT<MathDevice>(S, *opts, photon_budget, rgb8, photons_out, n_stored, n_emitted, draws);
    return BeginRenderT<MathLibm>(S, *opts, photon_budget, rgb8, photons_out, n_stored, n_emitted, draws);
}
