// device_trace.h — device-side intersection and traversal for gfx950.
//
// Reproduces, decision for decision, the reference's closest-hit and any-hit walks
// (all paths relative to /root/reference/BHRayTracer):
//   recursive()                     Main.cpp:389-413          scene-graph DFS, later-wins ties
//   Node::ToNodeCoords              Scenes/scene.h:490-496    per-level transform chain (never composed)
//   Sphere/Plane::IntersectRay      Objects/Sphere/Sphere.cpp:8-75, Objects/Plane/Plane.cpp:8-77
//   Box::IntersectRay               Objects/Box/Box.cpp:3-46  slab test by DIVISION
//   TriObj::IntersectRay/TraceBVHNode/IntersectTriangle   Objects/TriObj/TriObj.cpp:17-39,68-270
//   GenLight::Shadow/ShadowRayRecursive                   Lights/GenLight.cpp:10-69
//   TriObj::ShadowRecursive/TraceBVHShadow                TriObj.cpp:41-66,272-307
// but restructured for a GPU lane: no recursion, no per-lane stack.  The BVH walk is a
// state machine over (node, depth, two trail bits per level) with parent links; only the
// ray parameter, node, triangle id and side are produced here (hit attributes are recomputed
// by the shading kernel from this compact record).
// Must be compiled with -ffp-contract=off.
#pragma once
#include "device_types.h"
#include "vecmath.h"

namespace bhrt {

#define BHRT_PERP 0.001745f     /* TriObj.cpp:12 */
#define BHRT_TRI_BIAS 0.0001f   /* TriObj.cpp:9 */
#define BHRT_SHADOW_BIAS 0.00001f /* GenLight.cpp:5 */

struct Hit {
    float t;
    int node, prim, front;
};

#ifdef BHRT_DEBUG_STREAM
__device__ unsigned long long g_walk_dbg[16];
// adds v to counter i once per wave (called from divergent code: the first active lane adds)
__device__ inline void walk_dbg(int i, unsigned long long v) { if (__lane_id() == (uint32_t)__ffsll((long long)__ballot(true)) - 1u) atomicAdd(&g_walk_dbg[i], v); }
#define WALK_DBG(i, v) walk_dbg(i, v)
#else
#define WALK_DBG(i, v)
#endif
__device__ inline V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

// Node::ToNodeCoords (scene.h:490-496): p' = itm*(p-pos); d' = itm*((p+d)-pos) - p'
__device__ inline void to_node(const bhrt_xform &t, V3 &p, V3 &d)
{
    V3 pos = ld3(t.pos);
    V3 np = mat_mul(t.itm, p - pos);
    V3 nd = mat_mul(t.itm, (p + d) - pos) - np;
    p = np;
    d = nd;
}
// the same with rootNode's identity transformation (ShadowRayRecursive starts AT the root: GenLight.cpp:17)
__device__ inline void to_node_identity(V3 &p, V3 &d)
{
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    V3 z = v3(0, 0, 0);
    V3 np = mat_mul(I, p - z);
    V3 nd = mat_mul(I, (p + d) - z) - np;
    p = np;
    d = nd;
}
// ray of node n's object space, through the chain of its ancestors (root excluded)
__device__ inline void local_ray(const DevScene &S, int n, V3 &p, V3 &d)
{
    const int depth = S.nodes[n].depth;
    const int32_t *ch = S.chain + (size_t)n * BHRT_MAX_NODE_DEPTH;
    for (int k = 0; k < depth; k++) to_node(S.nodes[ch[k]].xf, p, d);
}

#define BHRT_FAST_REL 1.9073486328125e-06f /* 2^-19 */
#define BHRT_FAST_ABS 7.888609052210118e-31f /* 2^-100 */

// local_ray for a camera ray: the origin's part of every level comes from DevScene::cam_chain (18 of the 42 operations)
__device__ inline void local_ray_camera(const DevScene &S, int n, V3 &p, V3 &d)
{
    const int depth = S.nodes[n].depth;
    const int32_t *ch = S.chain + (size_t)n * BHRT_MAX_NODE_DEPTH;
    const float *cp = S.cam_chain + (size_t)n * BHRT_MAX_NODE_DEPTH * 3;
    for (int k = 0; k < depth; k++) {
        const bhrt_xform &t = S.nodes[ch[k]].xf;
        const V3 np = ld3(cp + 3 * k);
        d = mat_mul(t.itm, (p + d) - ld3(t.pos)) - np;
        p = np;
    }
}

// Box::IntersectRay (Box.cpp:3-46).  The reference forms n.Dot(v) with axis vectors; the zero products only
// affect the sign of a zero, which no comparison below can see, so the axis components are used directly.
__device__ inline bool box_hit(const float *b, V3 o, V3 d, float t_max, float &t_min)
{
    float tz1 = (d.z != 0) ? (b[2] - o.z) / d.z : BHRT_BIGFLOAT;
    float tz2 = (d.z != 0) ? (b[5] - o.z) / d.z : -BHRT_BIGFLOAT;
    float ty1 = (d.y != 0) ? (b[1] - o.y) / d.y : BHRT_BIGFLOAT;
    float ty2 = (d.y != 0) ? (b[4] - o.y) / d.y : -BHRT_BIGFLOAT;
    float tx1 = (d.x != 0) ? (b[0] - o.x) / d.x : BHRT_BIGFLOAT;
    float tx2 = (d.x != 0) ? (b[3] - o.x) / d.x : -BHRT_BIGFLOAT;
    float tMin = fmax_cy(fmax_cy(fmin_cy(tx1, tx2), fmin_cy(ty1, ty2)), fmin_cy(tz1, tz2));
    float tMax = fmin_cy(fmin_cy(fmax_cy(tx1, tx2), fmax_cy(ty1, ty2)), fmax_cy(tz1, tz2));
    if (tMin <= tMax && tMin < t_max) { t_min = tMin; return true; }
    return false;
}

// tMin of Box::IntersectRay alone, for a ray without a zero direction component: on every axis the smaller of the two quotients is the one of
// the plane the ray meets first — (b_lo - o) / d <= (b_hi - o) / d for d > 0 and the other way round for d < 0, because subtraction and
// division round monotonically (equal quotients: either) — so three IEEE divisions give the same float as the six with their minima.
__device__ inline float box_near_exact(const float *b, V3 o, V3 d)
{
    const float tx = ((d.x > 0 ? b[0] : b[3]) - o.x) / d.x, ty = ((d.y > 0 ? b[1] : b[4]) - o.y) / d.y, tz = ((d.z > 0 ? b[2] : b[5]) - o.z) / d.z;
    return fmax_cy(fmax_cy(tx, ty), tz);
}

// The same slab test for many boxes along ONE ray: the six IEEE float divisions per box become
// float(double(a) * rd) with rd = 1.0 / double(d) computed once per ray and axis.  This is bit-identical to the
// float division a / d: the quotient of two binary32 numbers is never closer than 2^-49 (relative) to a rounding
// boundary of binary32 (a midpoint between neighbours, or the overflow threshold), while double(a)*rd differs from a/d by
// at most 2^-52 (relative), so both round to the same float — also to the same infinity.  The argument needs 24 bits of
// quotient: a subnormal quotient takes the plain division (and so does an exact zero, which the one cheap test below cannot
// tell from it), and so does a ray with a zero direction component, for which the reference does not divide at all.
// (Per-quotient exponent checks instead of the one min-of-magnitudes test: 13 % more time in k_trace_mesh.)
struct RayRcp {
    double rx, ry, rz;
    bool slow; // a zero direction component: the reference replaces those quotients by +-BIGFLOAT (Box.cpp), plain form
};
__device__ inline RayRcp ray_rcp(V3 d)
{
    RayRcp r;
    r.rx = 1.0 / (double)d.x; r.ry = 1.0 / (double)d.y; r.rz = 1.0 / (double)d.z;
    r.slow = d.x == 0 || d.y == 0 || d.z == 0;
    return r;
}
// v_min_f32 / v_max_f32 / v_min3_f32 / v_max3_f32 as they are: fminf / fmaxf first quiet each operand (`v_max_f32 x, x`,
// six extra instructions per box) because the compiler cannot know that no signalling NaN arrives here.
__device__ inline float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ inline float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ inline float hw_min_abs(float a, float b) { float r; asm("v_min_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
// kAsm = false: the same with fminf / fmaxf — the any-hit loop (mesh_shadow) is 25 % SLOWER with the asm form (the opaque
// instructions change how its single loop is scheduled), the closest-hit loop 2.5 % faster.
template <bool kAsm = true>
__device__ inline bool box_hit_rcp(const float *b, V3 o, V3 d, const RayRcp &r, float t_max, float &t_min)
{
    const float tz1 = (float)((double)(b[2] - o.z) * r.rz), tz2 = (float)((double)(b[5] - o.z) * r.rz);
    const float ty1 = (float)((double)(b[1] - o.y) * r.ry), ty2 = (float)((double)(b[4] - o.y) * r.ry);
    const float tx1 = (float)((double)(b[0] - o.x) * r.rx), tx2 = (float)((double)(b[3] - o.x) * r.rx);
    // a zero or subnormal quotient (double rounding could differ there) sends the whole box to the plain divisions; one test
    const float smallest = kAsm ? hw_min3(hw_min_abs(tz1, tz2), hw_min_abs(ty1, ty2), hw_min_abs(tx1, tx2))
                                : fminf(fminf(fminf(fabsf(tz1), fabsf(tz2)), fminf(fabsf(ty1), fabsf(ty2))), fminf(fabsf(tx1), fabsf(tx2)));
    if (r.slow || !(smallest >= 1.17549435e-38f)) return box_hit(b, o, d, t_max, t_min);
    // no NaN and no zero among the six from here on: cyMin / cyMax (`a <= b ? a : b`, compare + select) and the hardware's
    // min / max instructions give the same bits (they differ only on NaNs and on the sign of a zero)
    const float tMin = kAsm ? hw_max3(hw_min(tx1, tx2), hw_min(ty1, ty2), hw_min(tz1, tz2)) : fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    const float tMax = kAsm ? hw_min3(hw_max(tx1, tx2), hw_max(ty1, ty2), hw_max(tz1, tz2)) : fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    if (tMin <= tMax && tMin < t_max) { t_min = tMin; return true; }
    return false;
}

// Approximate-first slab test.  Every use of the slab test's results in the closest-hit traversal is a comparison
// (tMin <= tMax, tMin < t_max, tmin of child 1 < tmin of child 2), so the six correctly rounded quotients are only needed when
// a comparison is close.  q' = RN(a * RN(1/d)) is within 3 * 2^-24 (relative) of the rounded quotient RN(a/d) for normal
// results; min / max keep that bound (two-sided, relative to the larger magnitude), so the approximate tMin / tMax are within
// 2^-21 * |own value| (+ 2^-125 for subnormal quotients) of the exact ones.  A comparison is taken from the approximations
// only when the two sides differ by more than 2^-19 * (|a| + |b|) + 2^-100 — 8x the bound, which also covers the rounding of
// the difference and of the tolerance themselves; otherwise (and for rays with a zero direction component, whose quotients
// the reference replaces by +-BIGFLOAT) the box is evaluated exactly (box_hit_rcp).  Infinite or NaN approximations make
// |diff| > tol false, i.e. indecisive.  3 instructions per quotient become 1.
struct RayRcpF {
    float rx, ry, rz;
    bool slow;
};
__device__ inline RayRcpF ray_rcp_f(V3 d)
{
    RayRcpF r;
    r.rx = 1.0f / d.x; r.ry = 1.0f / d.y; r.rz = 1.0f / d.z;
    // zero components (see above), and components whose reciprocal would not be a normal float with full precision
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    r.slow = !(fminf(fminf(ax, ay), az) >= BHRT_FAST_ABS && fmaxf(fmaxf(ax, ay), az) <= 1.2676506e30f /* 2^100 */);
    return r;
}
// 1 = hit, 0 = miss, -1 = too close to call.  One comparison decides both conditions: with u = min(tMax', t_max),
// u - tMin' > tol means tMin < tMax and tMin < t_max for certain (tMax' >= u, and the error of tMax' relative to |u| is no
// larger than relative to itself on the side that matters), u - tMin' < -tol means one of them fails for certain.
__device__ inline int box_fast(const float *b, V3 o, const RayRcpF &r, float t_max, float &t_min_approx, float &t_max_approx /* of the slabs alone, without t_max */)
{
    const float tz1 = (b[2] - o.z) * r.rz, tz2 = (b[5] - o.z) * r.rz;
    const float ty1 = (b[1] - o.y) * r.ry, ty2 = (b[4] - o.y) * r.ry;
    const float tx1 = (b[0] - o.x) * r.rx, tx2 = (b[3] - o.x) * r.rx;
    const float tMin = hw_max3(hw_min(tx1, tx2), hw_min(ty1, ty2), hw_min(tz1, tz2));
    const float tMax = hw_min3(hw_max(tx1, tx2), hw_max(ty1, ty2), hw_max(tz1, tz2));
    const float u = hw_min(tMax, t_max);
    const float diff = u - tMin, tol = fmaf(BHRT_FAST_REL, fabsf(tMin) + fabsf(u), BHRT_FAST_ABS);
    t_min_approx = tMin;
    t_max_approx = tMax;
    return diff > tol ? 1 : (diff < -tol ? 0 : -1);
}
__device__ inline int box_fast(const float *b, V3 o, const RayRcpF &r, float t_max, float &t_min_approx)
{
    float t_max_approx;
    return box_fast(b, o, r, t_max, t_min_approx, t_max_approx);
}
// the same for the any-hit loop, with fminf / fmaxf (see box_hit_rcp<false>: that loop schedules worse around the asm forms)
__device__ inline int box_fast_f(const float *b, V3 o, const RayRcpF &r, float t_max, float &t_min_approx, float &t_max_approx)
{
    const float tz1 = (b[2] - o.z) * r.rz, tz2 = (b[5] - o.z) * r.rz;
    const float ty1 = (b[1] - o.y) * r.ry, ty2 = (b[4] - o.y) * r.ry;
    const float tx1 = (b[0] - o.x) * r.rx, tx2 = (b[3] - o.x) * r.rx;
    const float tMin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    const float tMax = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    const float u = fminf(tMax, t_max);
    const float diff = u - tMin, tol = fmaf(BHRT_FAST_REL, fabsf(tMin) + fabsf(u), BHRT_FAST_ABS);
    t_min_approx = tMin;
    t_max_approx = tMax;
    return diff > tol ? 1 : (diff < -tol ? 0 : -1);
}
__device__ inline int box_fast_f(const float *b, V3 o, const RayRcpF &r, float t_max)
{
    float a, c;
    return box_fast_f(b, o, r, t_max, a, c);
}
// tmin1 < tmin2 from the approximations: 1 / 0, or -1 = too close to call
__device__ inline int order_fast(float tmin1, float tmin2)
{
    const float d = tmin2 - tmin1, tol = fmaf(BHRT_FAST_REL, fabsf(tmin1) + fabsf(tmin2), BHRT_FAST_ABS);
    return fabsf(d) > tol ? (d > 0 ? 1 : 0) : -1;
}
// Box::IntersectRay's decision for ONE box (a mesh's root box: the gate of TriObj::IntersectRay / ShadowRecursive), approximate first: the
// three double reciprocals of the exact form (ray_rcp, ~100 instructions) are only formed when the approximation cannot tell.  t_min: the
// approximate entry distance in that case (the parking kernel's sort key is made of it: an ordering hint), the exact one otherwise.
template <bool kAsm = true>
__device__ inline bool box_hit_lazy(const float *b, V3 o, V3 d, const RayRcpF &rf, float t_max, float &t_min)
{
    int f = -1;
    if (!rf.slow) f = kAsm ? box_fast(b, o, rf, t_max, t_min) : box_fast_f(b, o, rf, t_max);
    if (f < 0) return box_hit_rcp<kAsm>(b, o, d, ray_rcp(d), t_max, t_min);
    return f == 1;
}

// barycentric part of IntersectTriangle (TriObj.cpp:105-168), shared with the attribute recomputation; the vertices come
// projected (bhrt_tri), only the hit point is projected here
__device__ inline bool tri_areas(const bhrt_tri &tr, V3 vX, float &a0, float &a1, float &a2)
{
    const uint32_t axis = tr.face_axis >> 30;
    float pXx = axis == 0 ? vX.y : vX.x, pXy = axis == 2 ? vX.y : vX.z;
    if (axis == 3) { pXx = 0; pXy = 0; }
    const float p0x = tr.p0[0], p0y = tr.p0[1], p1x = tr.p1[0], p1y = tr.p1[1], p2x = tr.p2[0], p2y = tr.p2[1];
    // Vec2::Cross: (-y)*p.x + x*p.y (cyVector.h:260-262)
    float e1x = p1x - pXx, e1y = p1y - pXy, e2x = p2x - pXx, e2y = p2y - pXy, e0x = p0x - pXx, e0y = p0y - pXy;
    a0 = ((-e1y) * e2x + e1x * e2y) / 2.f;
    a1 = ((-e2y) * e0x + e2x * e0y) / 2.f;
    a2 = ((-e0y) * e1x + e0x * e1y) / 2.f;
    if ((a2 < 0 || a1 < 0 || a0 < 0) && !(a0 < 0 && a1 < 0 && a2 < 0)) return false;
    return true;
}

// TriObj::IntersectTriangle (TriObj.cpp:68-189) up to the accept decision, in two parts: the plane part (TriObj.cpp:79-103: needs the first
// 20 bytes of the record) and the barycentric part (the rest of the record, read only when the plane hit lies in range).
// vN, |vN| and vN.v0 come precomputed with the triangle (same float operations as TriObj.cpp:79,85,89); dlen = ray.dir.Length().
struct TriPlane { float nx, ny, nz, nd, len; };
__device__ inline TriPlane tri_plane_ld(const bhrt_tri *tr)
{
    const float4 a = *(const float4 *)tr->vN; // vN, vN_dot_v0: one 16-byte load
    TriPlane P;
    P.nx = a.x; P.ny = a.y; P.nz = a.z; P.nd = a.w; P.len = tr->vN_len;
    return P;
}
// kRange = false: without the range test `t > hInfo.z` (TriObj.cpp:92) — for a caller that tests the triangles of a leaf side by side and applies
// that comparison itself, in triangle order, against the bound the earlier triangles have left
template <bool kRange = true>
__device__ inline bool tri_plane_test(const TriPlane &P, V3 o, V3 d, float dlen, int side, float t_cur, float &t, bool &hitFront)
{
    // The reference returns at each failed test; here the cheap tests are evaluated straight through and AND-ed (a wave
    // of 64 rays almost always has a lane that passes each of them, so the early returns only cost branches), with one
    // branch left in front of the barycentric part.  Divisions by zero only feed predicates that are already false.
    const V3 vN = v3(P.nx, P.ny, P.nz);
    const float t_divisor = dot(vN, d);
    bool ok = t_divisor != 0;
    // grazing test `abs(t_divisor / (|vN| * |d|)) < 0.001745` (TriObj.cpp:85-88): the quotient is only compared, so an
    // approximate one (hardware reciprocal: within 2^-21 of the rounded quotient) decides unless it lies within 2^-19 of the
    // threshold; a zero or subnormal denominator gives an infinite approximation = "not grazing", like the division does
    const float den = P.len * dlen;
    const float aperp = fabsf(t_divisor * __builtin_amdgcn_rcpf(den));
    bool grazing = aperp < BHRT_PERP;
    if (fabsf(aperp - BHRT_PERP) <= BHRT_PERP * BHRT_FAST_REL || aperp != aperp) {
        WALK_DBG(12, 1);
        const float perp = t_divisor / den;
        grazing = perp > -BHRT_PERP && perp < BHRT_PERP;
    }
    ok = ok && !grazing;
    t = (P.nd - dot(vN, o)) / t_divisor;
    ok = ok && !(t <= 0 || (kRange && t > t_cur));
    hitFront = t_divisor < 0;
    return ok && !(!hitFront && side == BHRT_HIT_FRONT) && !(hitFront && side == BHRT_HIT_BACK);
}
__device__ inline bool tri_hit(const bhrt_tri &tr, V3 o, V3 d, float dlen, int side, float t_cur, float &t_out, int &front_out)
{
    float t;
    bool hitFront;
    if (!tri_plane_test(tri_plane_ld(&tr), o, d, dlen, side, t_cur, t, hitFront)) return false;
    V3 vX = o + t * d;
    float a0, a1, a2;
    if (!tri_areas(tr, vX, a0, a1, a2)) return false;
    t_out = t;
    front_out = hitFront ? 1 : 0;
    return true;
}
// The triangles [off, off + count) of a leaf in order (TriObj.cpp:231-246: every hit lowers the bound the next triangle is tested against).
// The plane parts of four triangles are fetched TOGETHER: fetched one by one inside the loop, every triangle costs the wave two to three
// memory round trips in a row (plane part, then the vertices) — a leaf of four was 8-12 dependent waits long, the longest chain of a round.
#ifndef BHRT_LEAF_BATCH
#define BHRT_LEAF_BATCH 1
#endif
__device__ inline bool leaf_hits(const bhrt_tri *tris, uint32_t off, uint32_t count, V3 o, V3 d, float dlen, int side, float &ht, int &hprim, int &hfront)
{
    bool r = false;
#if BHRT_LEAF_BATCH
    for (uint32_t base = 0; base < count; base += 4) { // a leaf holds at most 8
        TriPlane P[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) P[k] = tri_plane_ld(tris + off + min(base + k, count - 1)); // past the end: the last one again (not tested)
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (base + k < count) {
                WALK_DBG(8, 1); WALK_DBG(9, __popcll(__ballot(true)));
                const bhrt_tri &tr = tris[off + base + k];
                float t;
                bool hitFront;
                if (tri_plane_test(P[k], o, d, dlen, side, ht, t, hitFront)) {
                    WALK_DBG(10, 1); WALK_DBG(11, __popcll(__ballot(true)));
                    float a0, a1, a2;
                    if (tri_areas(tr, o + t * d, a0, a1, a2)) { ht = t; hprim = (int)(tr.face_axis & 0x3fffffffu); hfront = hitFront ? 1 : 0; r = true; }
                }
            }
        }
    }
#else
    for (uint32_t i = 0; i < count; i++) {
        const bhrt_tri &tr = tris[off + i];
        float t;
        int fr;
        if (tri_hit(tr, o, d, dlen, side, ht, t, fr)) { ht = t; hprim = (int)(tr.face_axis & 0x3fffffffu); hfront = fr; r = true; }
    }
#endif
    return r;
}

struct MeshRef {
    const bhrt_bvh_node *bvh;  // breadth-first numbered tree (bhrt_mesh::off_dbvh)
    const bhrt_tri *ltris;     // leaf order: ltris[off + i] is the i-th triangle of the leaf with element offset `off`
    const bhrt_bvh_node *lds;  // the first `n_lds` nodes (= the top levels) staged in LDS by the workgroup, or nullptr
    uint32_t n_lds;
    bool nested;               // bhrt_mesh::bvh_nested: a box-missed inner sibling cannot produce a hit (see skip_missed below)
    const uint32_t *dpar;      // parent links of the breadth-first copy (in that copy a LEAF's `parent` word is its packed normal cone: leaf_skip)
    float k0, k1, big, omax;   // bhrt_mesh::skip_*: the mesh-wide constants of leaf_skip (omax = 0: no leaf qualifies)
};
__device__ inline MeshRef mesh_ref(const DevScene &S, int mi)
{
    const bhrt_mesh &m = S.meshes[mi];
    MeshRef r;
    r.bvh = (const bhrt_bvh_node *)(S.blob + m.off_dbvh);
    r.ltris = (const bhrt_tri *)(S.blob + m.off_leaf_tris);
    r.lds = nullptr;
    r.n_lds = 0;
    r.nested = m.bvh_nested != 0;
    r.dpar = (const uint32_t *)(S.blob + m.off_dparent);
    r.k0 = m.skip_k0; r.k1 = m.skip_k1; r.big = m.skip_big; r.omax = m.skip_omax;
    return r;
}
// TraceBVHNode visits the sibling of a child that returned nothing even when the sibling's box was missed (TriObj.cpp:245-248,
// 263-266).  When that sibling is an INNER node the visit consists of the two box tests of ITS children and ends
// there: Box::IntersectRay is a chain of monotone float operations of the box bounds (b - o, the division by d, min / max: each
// rounds monotonically; an axis with d == 0 drops out of parent and child alike; a NaN from the ray sits in the same places for
// both), the child boxes are nested in the missed box float for float (bhrt_mesh::bvh_nested, checked at load), and the ray and
// t_max are the ones the missed test used (the first child returned no hit, so HitInfo is unchanged) — hence tMin(child) >=
// tMin(box) and tMax(child) <= tMax(box), both children miss, `return false`.  After a hit in the first child the sibling is
// only visited when `hit.z > tmin_sibling`, and tmin of a missed box stays BIGFLOAT.  So a level whose second-visited child is
// a box-missed inner node is complete once the first child returns: it is marked done when it is opened.  A box-missed LEAF
// sibling is still visited — its triangles are tested without a box test in the reference, and nothing ties their arithmetic
// to the slab test's.  One third of all node visits on the 100 k-triangle mesh are such siblings (87 % of them inner nodes).
#ifndef BHRT_SKIP_MISSED
#define BHRT_SKIP_MISSED 1
#endif
__device__ inline bool skip_missed(const MeshRef &M, bool box_hit, uint32_t data) { return BHRT_SKIP_MISSED && M.nested && !box_hit && !(data & 0x80000000u); }
// A box-missed LEAF sibling (36 % of the closed room's leaf visits, never a hit in 10^9 of them) is visited by the reference all the same, and its
// triangles are tested without a box test: nothing in IntersectTriangle (TriObj.cpp:68-189) knows the box, and from far away the test's three
// signed areas are rounding noise.  The visit may be left out when it PROVABLY accepts nothing (the proof, and what the host computes per
// leaf and per mesh for it, is in scene_host.cpp::ComputeLeafSkip; DESIGN.md 4 "leaf skip").  The ray must
//   * be tame: max |o_i| <= M.omax (and no zero / tiny / huge direction component: the approximate slab test was decisive);
//   * MISS the leaf box inflated by m = k0 + k1 max|o_i| as a LINE (not merely lie beyond the current hit): with tMin / tMax the approximate
//     slab values of the plain box, inflating by m moves every quotient by at most m max|1/d_i|, so  tMin - tMax > 2 m max|1/d_i| + tol  does it;
//   * pass through the box inflated by `big` (a NEAR miss: the hit point with a triangle's plane then stays close enough for the sign test to
//     be exact):  tMin - tMax < 2 big min|1/d_i| - tol;
//   * not graze any triangle of the leaf: |d . A| >= |d| with A the leaf's packed cone (three 10-bit components, 2-bit exponent; 0 = never).
// aux = the leaf's `parent` word in the breadth-first copy.  tol: twice the slab test's own (the bounds here are about real arithmetic).
// Compiled into the walks of a render with bhrt_opts::leaf_skip only (template parameter kLS of walk_round / mesh_shadow_stack and the kernels that
// hold them): its mere presence in the default kernels — behind a wave-uniform branch never taken — cost them 3-11 % (registers, code layout).
__device__ inline bool leaf_skip(const MeshRef &M, uint32_t aux, float tmin, float tmax, V3 o, V3 d, const RayRcpF &rf, float dlen)
{
    if (aux == 0u) return false;
    const float omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const float rmax = fmaxf(fmaxf(fabsf(rf.rx), fabsf(rf.ry)), fabsf(rf.rz)), rmin = fminf(fminf(fabsf(rf.rx), fabsf(rf.ry)), fabsf(rf.rz));
    const float gap = tmin - tmax, tol = fmaf(2.f * BHRT_FAST_REL, fabsf(tmin) + fabsf(tmax), BHRT_FAST_ABS);
    const float m = fmaf(M.k1, omax, M.k0);
    bool ok = omax <= M.omax && gap > fmaf(2.000002f * m, rmax, tol) && gap < fmaf(1.999998f * M.big, rmin, -tol);
    const int qx = (int)(aux << 22) >> 22, qy = (int)(aux << 12) >> 22, qz = (int)(aux << 2) >> 22, e = (int)(aux >> 30);
    const float dq = d.x * (float)qx + d.y * (float)qy + d.z * (float)qz;
    return ok && fabsf(dq) >= ldexpf(dlen, 8 - e) * 1.000244140625f; // (1 + 2^-12): the rounding of the three products and of |d|
}
// one BVH node (32 B) from the LDS nodelet when it is one of the staged top levels, else from global memory
struct NodeRec {
    float b[6];
    uint32_t data, parent;
};
__device__ inline NodeRec node_at(const MeshRef &M, uint32_t i)
{
    NodeRec n;
    if (i < M.n_lds) {
        const float4 *p = (const float4 *)(M.lds + i);
        const float4 a = p[0], c = p[1];
        n.b[0] = a.x; n.b[1] = a.y; n.b[2] = a.z; n.b[3] = a.w; n.b[4] = c.x; n.b[5] = c.y;
        n.data = __float_as_uint(c.z); n.parent = __float_as_uint(c.w);
    } else {
        const float4 *p = (const float4 *)(M.bvh + i);
        const float4 a = p[0], c = p[1];
        n.b[0] = a.x; n.b[1] = a.y; n.b[2] = a.z; n.b[3] = a.w; n.b[4] = c.x; n.b[5] = c.y;
        n.data = __float_as_uint(c.z); n.parent = __float_as_uint(c.w);
    }
    return n;
}
// the two children of an inner node (adjacent, first id even): both in the LDS nodelet or both in global memory
// (BHRT_LDS_NODES is even), so one branch and four 16-byte loads fetch the pair
__device__ inline void node_pair_at(const MeshRef &M, uint32_t c1, NodeRec &n1, NodeRec &n2)
{
    float4 a, b, c, d;
    if (c1 < M.n_lds) {
        const float4 *p = (const float4 *)(M.lds + c1);
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
    } else {
        const float4 *p = (const float4 *)(M.bvh + c1);
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
    }
    n1.b[0] = a.x; n1.b[1] = a.y; n1.b[2] = a.z; n1.b[3] = a.w; n1.b[4] = b.x; n1.b[5] = b.y;
    n1.data = __float_as_uint(b.z); n1.parent = __float_as_uint(b.w);
    n2.b[0] = c.x; n2.b[1] = c.y; n2.b[2] = c.z; n2.b[3] = c.w; n2.b[4] = d.x; n2.b[5] = d.y;
    n2.data = __float_as_uint(d.z); n2.parent = __float_as_uint(d.w);
}
__device__ inline uint32_t node_parent(const MeshRef &M, uint32_t i) { return M.dpar[i]; }
__device__ inline uint32_t node_data(const MeshRef &M, uint32_t i) { return i < M.n_lds ? M.lds[i].data : M.bvh[i].data; }

// the same from global memory only (mesh_closest_vote: with a nodelet the pointer select turns every node fetch into a flat
// load; the top levels stay in L1/L2 anyway: 50.8 -> 49.5 ms on C3 without)
__device__ inline NodeRec node_at_g(const MeshRef &M, uint32_t i)
{
    NodeRec n;
    const float4 *p = (const float4 *)(M.bvh + i);
    const float4 a = p[0], c = p[1];
    n.b[0] = a.x; n.b[1] = a.y; n.b[2] = a.z; n.b[3] = a.w; n.b[4] = c.x; n.b[5] = c.y;
    n.data = __float_as_uint(c.z); n.parent = __float_as_uint(c.w);
    return n;
}
__device__ inline void node_pair_at_g(const MeshRef &M, uint32_t c1, NodeRec &n1, NodeRec &n2)
{
    const float4 *p = (const float4 *)(M.bvh + c1);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    n1.b[0] = a.x; n1.b[1] = a.y; n1.b[2] = a.z; n1.b[3] = a.w; n1.b[4] = b.x; n1.b[5] = b.y;
    n1.data = __float_as_uint(b.z); n1.parent = __float_as_uint(b.w);
    n2.b[0] = c.x; n2.b[1] = c.y; n2.b[2] = c.z; n2.b[3] = c.w; n2.b[4] = d.x; n2.b[5] = d.y;
    n2.data = __float_as_uint(d.z); n2.parent = __float_as_uint(d.w);
}

// TriObj::IntersectRay + TraceBVHNode (TriObj.cpp:17-39,192-270) as a stackless state machine.
//   descending into a node: leaf -> test its <=4 triangles in order; inner -> test both child boxes with
//   t_max = current hit, none hit -> "return false", else go to the nearer child (child1 iff tmin1 < tmin2).
//   ascending from a child with its return value r:
//     from the FIRST-visited child: r true  -> the sibling is visited only if its box is hit with
//                                              t_max = current hit (== `tempHitInfo.z > tmin_far`, TriObj.cpp:235,255);
//                                    r false -> the sibling is visited unconditionally, even if its box
//                                              was missed (TriObj.cpp:245-248,263-266)
//     from the SECOND-visited child: return (first child's r) ? true : r
//   children are adjacent and the first child's id is even (cyBVH.h:281-291), so sibling = id ^ 1.
// Trail: bit (depth-1) of inFar / nearHit per level; depth <= 64 is checked at upload.
// One ray, one wave (k_trace_slow): the summaries of up to 64 subtrees that the lanes have walked side by side (mesh_closest_coop below).
struct CoopLds {
    uint8_t slot[2304];   // node id -> index of its summary, 0xff = none
    uint32_t node[64];
    uint32_t r[64];       // TraceBVHNode's return value for the subtree ...
    float ht[64];         // ... and the hit it leaves behind when that is true
    int32_t prim[64], front[64];
    uint32_t n, ht_in;    // how many; bits of the HitInfo::z every one of them started from
    uint32_t queue[128];  // scratch of the cut
};
// TraceBVHNode(start) — the walk below `start`, whose own box the caller has dealt with.  active = false: the lane takes no part.
// frontier != nullptr: a node that has a summary is not walked: the summary stands in for it when the hit distance is still the one it was
// computed from (nothing else of the state enters a subtree's decisions), else the subtree is walked now.
template <bool kStitch = false>
__device__ inline bool mesh_closest_from(const MeshRef &M, uint32_t start, V3 o, V3 d, int side, float &ht, int &hprim, int &hfront, bool active = true,
                                         CoopLds *frontier = nullptr)
{
    auto has_summary = [&](uint32_t id) { return kStitch && id < 2304u && frontier->slot[id] != 0xff; };
    int st = 3; // phase the lane waits for: 0 step down through an inner node, 1 leaf, 2 step up, 3 done, 4 a node with a summary
    uint32_t data = active ? node_data(M, start) : 0u;
    if (active) st = has_summary(start) ? 4 : ((data & 0x80000000u) ? 1 : 0);
    const RayRcpF rf = ray_rcp_f(d);
    const float dlen = length(d);
    uint32_t cur = start;
    int depth = 0;
    uint64_t inFar = 0, nearHit = 0;
    bool r = false, any = false;
    // Every round the wave runs ONE phase body, the one most of its lanes wait for (see mesh_closest_vote, which also keeps
    // the path in LDS; this form walks parent links and has no depth limit below the 64 trail bits).
    while (true) {
        const int nD = __popcll(__ballot(st == 0)), nL = __popcll(__ballot(st == 1)), nC = __popcll(__ballot(st == 2)), nF = kStitch ? __popcll(__ballot(st == 4)) : 0;
        if (nD + nL + nC + nF == 0) break;
        if (kStitch && nF > 0) { // every lane is in the same state when stitching
            if (st == 4) {
                if (__float_as_uint(ht) != frontier->ht_in) {
                    // a hit has moved the bound since the summaries were made: every lane walks its subtree again from the new one
                    // (the recursion meets a handful of ever closer hits along a ray, so this happens a handful of times)
                    const uint32_t lane = threadIdx.x & 63u, n = frontier->n;
                    float sh = ht;
                    int sp = hprim, sf = hfront;
                    const bool sr = mesh_closest_from<false>(M, lane < n ? frontier->node[lane] : 1u, o, d, side, sh, sp, sf, lane < n);
                    __syncthreads();
                    if (lane < n) { frontier->r[lane] = sr ? 1u : 0u; frontier->ht[lane] = sh; frontier->prim[lane] = sp; frontier->front[lane] = sf; }
                    if (lane == 0) frontier->ht_in = __float_as_uint(ht);
                    __syncthreads();
                }
                const uint32_t k = frontier->slot[cur];
                r = frontier->r[k] != 0;
                if (r) { ht = frontier->ht[k]; hprim = frontier->prim[k]; hfront = frontier->front[k]; }
                any |= r;
                st = 2;
            }
        } else if (nD >= nL && nD >= nC) {
            if (st == 0) {
                const uint32_t c1 = data & 0x7fffffffu;
                float tmin1 = BHRT_BIGFLOAT, tmin2 = BHRT_BIGFLOAT;
                NodeRec n1, n2;
                node_pair_at(M, c1, n1, n2);
                const uint32_t d1 = n1.data, d2 = n2.data;
                const int f1 = rf.slow ? -1 : box_fast(n1.b, o, rf, ht, tmin1), f2 = rf.slow ? -1 : box_fast(n2.b, o, rf, ht, tmin2);
                bool b1 = f1 == 1, b2 = f2 == 1;
                int ord = (b1 && b2) ? order_fast(tmin1, tmin2) : (b1 ? 1 : 0);
                if (f1 < 0 || f2 < 0 || ord < 0) { // a comparison too close to call (or a zero direction component): the exact test
                    const RayRcp rr = ray_rcp(d);
                    tmin1 = BHRT_BIGFLOAT; tmin2 = BHRT_BIGFLOAT;
                    b1 = box_hit_rcp(n1.b, o, d, rr, ht, tmin1);
                    b2 = box_hit_rcp(n2.b, o, d, rr, ht, tmin2);
                    ord = tmin1 < tmin2 ? 1 : 0;
                }
                if (!b1 && !b2) { r = false; st = 2; }
                else {
                    depth++;
                    const uint64_t bit = 1ull << (depth - 1);
                    const bool first1 = ord == 1;
                    inFar = skip_missed(M, first1 ? b2 : b1, first1 ? d2 : d1) ? (inFar | bit) : (inFar & ~bit);
                    nearHit &= ~bit;
                    cur = first1 ? c1 : c1 + 1;
                    data = first1 ? d1 : d2;
                    st = has_summary(cur) ? 4 : ((data & 0x80000000u) ? 1 : 0);
                }
            }
        } else if (nL >= nC) {
            if (st == 1) {
                const uint32_t count = ((data >> 28) & 7u) + 1, off = data & 0x0fffffffu;
                r = leaf_hits(M.ltris, off, count, o, d, dlen, side, ht, hprim, hfront);
                any |= r;
                st = 2;
            }
        } else {
            if (st == 2) { // one level up
                if (depth == 0) st = 3; // the call on `start` returned
                else {
                    const uint64_t bit = 1ull << (depth - 1);
                    const uint32_t sib = cur ^ 1u;
                    const bool sib_sum = has_summary(sib);
                    if (!(inFar & bit)) {
                        if (r) {
                            nearHit |= bit;
                            float tmf;
                            const NodeRec ns = node_at(M, sib);
                            const uint32_t ds = ns.data;
                            int fs = rf.slow ? -1 : box_fast(ns.b, o, rf, ht, tmf);
                            if (fs < 0) fs = box_hit_rcp(ns.b, o, d, ray_rcp(d), ht, tmf) ? 1 : 0;
                            if (fs) { inFar |= bit; cur = sib; data = ds; st = sib_sum ? 4 : ((ds & 0x80000000u) ? 1 : 0); }
                            else { cur = node_parent(M, cur); depth--; /* r stays true */ }
                        } else {
                            inFar |= bit;
                            cur = sib;
                            data = node_data(M, sib);
                            st = sib_sum ? 4 : ((data & 0x80000000u) ? 1 : 0);
                        }
                    } else {
                        r = (nearHit & bit) ? true : r;
                        cur = node_parent(M, cur);
                        depth--;
                    }
                }
            }
        }
    }
    return any;
}
// TriObj::IntersectRay (TriObj.cpp:17-39): the root box gate, then TraceBVHNode(root)
__device__ inline bool mesh_closest(const MeshRef &M, V3 o, V3 d, int side, float &ht, int &hprim, int &hfront)
{
    float tm;
    const bool enter = box_hit_rcp(node_at(M, 1).b, o, d, ray_rcp(d), ht, tm);
    return mesh_closest_from(M, 1, o, d, side, ht, hprim, hfront, enter);
}
// The same for ONE ray held by all 64 lanes of a wave (k_trace_slow: a ray parallel to a coordinate axis walks up to the whole tree).
// TraceBVHNode is a sequential recursion, but a subtree's part in it depends on the state it is entered with only through the hit
// distance.  So: the tree's top is cut at up to 64 nodes; every lane walks the subtree of one of them from the hit distance the mesh is
// entered with and keeps what TraceBVHNode would return (`r`) and leave behind (the hit); then the recursion is run from the root with
// those summaries in the subtrees' place — valid as long as the hit distance is still the one they started from; when a hit has changed
// it, the lanes walk their subtrees again from the new distance before the recursion goes on (a handful of times per ray: every time a
// closer hit turns up).  The same boxes, triangles and comparisons in the same order as the lone walk: the same result, in a few dozen
// parallel walks of 1/64 of the tree instead of ~10^5 dependent rounds.
__device__ inline bool mesh_closest_coop(const MeshRef &M, CoopLds &L, V3 o, V3 d, int side, float &ht, int &hprim, int &hfront)
{
    float tm;
    if (!box_hit_rcp(node_at(M, 1).b, o, d, ray_rcp(d), ht, tm)) return false; // uniform: every lane holds the same ray
    const uint32_t lane = threadIdx.x & 63u;
    __syncthreads();
    for (uint32_t k = lane; k < 2304u; k += 64) L.slot[k] = 0xff;
    if (lane == 0) { // the cut, breadth first: inner nodes are replaced by their two children, level by level, until there are 64 nodes (or only leaves)
        uint32_t nf = 0, head = 0, tail = 1;
        L.queue[0] = 1;
        while (head < tail && nf + (tail - head) < 64) {
            const uint32_t id = L.queue[head++], dat = M.bvh[id].data, c1 = dat & 0x7fffffffu;
            if ((dat & 0x80000000u) || c1 + 1 >= 2304u) L.node[nf++] = id;
            else { L.queue[tail++] = c1; L.queue[tail++] = c1 + 1; }
        }
        while (head < tail) L.node[nf++] = L.queue[head++];
        L.n = nf;
        L.ht_in = __float_as_uint(ht);
    }
    __syncthreads();
    const uint32_t n = L.n;
    if (lane < n) L.slot[L.node[lane]] = (uint8_t)lane;
    float sh = ht;
    int sp = hprim, sf = hfront;
    const bool sr = mesh_closest_from(M, lane < n ? L.node[lane] : 1u, o, d, side, sh, sp, sf, lane < n);
    if (lane < n) { L.r[lane] = sr ? 1u : 0u; L.ht[lane] = sh; L.prim[lane] = sp; L.front[lane] = sf; }
    __syncthreads();
    const bool any = mesh_closest_from<true>(M, 1, o, d, side, ht, hprim, hfront, true, &L);
    __syncthreads();
    return any;
}

// The traversal with the path kept in LDS (mesh_closest_vote below): stack[level * stride] holds the pair index (id >> 1) of the level's
// first-visited child, `side` its low id bit.  Returning from a second-visited child only passes the result up, so the climb
// jumps straight to the deepest level still waiting in its first-visited child (a bit scan over inFar) instead of walking
// parent links one dependent load at a time.  Needs depth <= 32 (the caller checks the meshes) and a path element wide enough for the pair indices.
// ... and with the wave running, in every round, the ONE phase most of its lanes wait for (descend step / leaf /
// climb step: a wave-uniform choice from three ballots) instead of the three phase loops in turn, each until its last lane
// is through: same per-ray operation sequence, ~1.2x the lanes per instruction (C3 trace 61 -> 51 ms).
#ifndef BHRT_FUSED_CLIMB
#define BHRT_FUSED_CLIMB 1
#endif
#ifndef BHRT_VOTE_D
#define BHRT_VOTE_D 1 /* the round is a descend step when lanes waiting for one x BHRT_VOTE_D >= lanes waiting for a leaf x BHRT_VOTE_L */
#define BHRT_VOTE_L 1
#endif
// One lane's walk, cut into rounds so that a wave can take new rays in between (kernels.hip::k_trace_mesh_stream): the state a round
// reads and writes.  st: 0 descend step, 1 leaf, 2 climb step, 3 done.
struct MeshWalk {
    V3 o, d; // the ray in the mesh's space
    RayRcpF rf;
    float dlen;
    uint32_t data, inFar, nearHit, sides;
    int depth, st;
    bool r, any;
};
// the root box gate of TriObj::IntersectRay (TriObj.cpp:17-39)
__device__ inline void walk_begin(const MeshRef &M, MeshWalk &W, V3 o, V3 d, float ht)
{
    float tm;
    const NodeRec root = node_at_g(M, 1);
    W.o = o; W.d = d;
    W.st = 3;
    W.data = root.data;
    W.rf = ray_rcp_f(d);
    if (box_hit_lazy(root.b, o, d, W.rf, ht, tm)) W.st = (W.data & 0x80000000u) ? 1 : 0;
    W.dlen = length(d);
    W.depth = 0;
    W.inFar = 0; W.nearHit = 0; W.sides = 0;
    W.r = false; W.any = false;
}
// One round: the wave runs the phase most of its lanes wait for (nD / nL / nC = lanes waiting for a descend step / a leaf / a climb step).
template <class PathT, bool kLS = false> // PathT: uint16_t (pair indices < 2^16: meshes below 2^17 nodes) or uint32_t; kLS: bhrt_opts::leaf_skip
__device__ inline void walk_round(const MeshRef &M, MeshWalk &W, int side, float &ht, int &hprim, int &hfront, PathT *stack, uint32_t stride, int nD, int nL, int nC)
{
    const V3 o = W.o, d = W.d;
#if BHRT_FUSED_CLIMB
    // the climb step rides at the end of EVERY round (lanes that have just left a leaf or missed both boxes included): a lane never waits for a
    // round of its own to climb, and the vote is between two phases
    if (nD * BHRT_VOTE_D >= nL * BHRT_VOTE_L && nD > 0) {
#else
    if (nD >= nL && nD >= nC) {
#endif
        if (W.st == 0) {
            const uint32_t c1 = W.data & 0x7fffffffu;
            float tmin1 = BHRT_BIGFLOAT, tmin2 = BHRT_BIGFLOAT;
            NodeRec n1, n2;
            node_pair_at_g(M, c1, n1, n2);
            const uint32_t d1 = n1.data, d2 = n2.data;
            float tmx1 = 0.f, tmx2 = 0.f;
            const int f1 = W.rf.slow ? -1 : box_fast(n1.b, o, W.rf, ht, tmin1, tmx1), f2 = W.rf.slow ? -1 : box_fast(n2.b, o, W.rf, ht, tmin2, tmx2);
            const float at1 = tmin1, at2 = tmin2; // the approximate slab values (leaf_skip)
            bool b1 = f1 == 1, b2 = f2 == 1;
            int ord = (b1 && b2) ? order_fast(tmin1, tmin2) : (b1 ? 1 : 0);
            WALK_DBG(0, 1); WALK_DBG(1, __popcll(__ballot(true)));
            if (f1 == 1 && f2 == 1 && ord < 0) {
                // Both boxes hit for certain, only their order is open — 98.5 % of the undecided steps (closed room), nearly all of them children
                // the ray enters through a face they share: the entry distances are then the same quotient.  Box::IntersectRay's tMin alone, by its
                // own divisions (box_near_exact): 6 quotients instead of the 24 + 3 double reciprocals of the two full boxes.
                WALK_DBG(2, 1); WALK_DBG(3, __popcll(__ballot(true)));
                ord = box_near_exact(n1.b, o, d) < box_near_exact(n2.b, o, d) ? 1 : 0;
            } else if (f1 < 0 || f2 < 0 || ord < 0) {
                WALK_DBG(13, 1);
                const RayRcp rr = ray_rcp(d);
                tmin1 = BHRT_BIGFLOAT; tmin2 = BHRT_BIGFLOAT;
                b1 = box_hit_rcp(n1.b, o, d, rr, ht, tmin1);
                b2 = box_hit_rcp(n2.b, o, d, rr, ht, tmin2);
                ord = tmin1 < tmin2 ? 1 : 0;
            }
            if (!b1 && !b2) { W.r = false; W.st = 2; }
            else {
                W.depth++;
                const uint32_t bit = 1u << (W.depth - 1);
                const bool first1 = ord == 1;
                bool sib_done = skip_missed(M, first1 ? b2 : b1, first1 ? d2 : d1);
                // a box-missed LEAF sibling whose visit provably accepts nothing (the approximate test itself must have said "missed")
                if (kLS && M.omax > 0.f && !sib_done && ((first1 ? d2 : d1) & 0x80000000u) && (first1 ? f2 : f1) == 0 && !(first1 ? b2 : b1)) {
                    sib_done = leaf_skip(M, first1 ? n2.parent : n1.parent, first1 ? at2 : at1, first1 ? tmx2 : tmx1, o, d, W.rf, W.dlen);
                    WALK_DBG(14, 1); WALK_DBG(15, sib_done ? 1 : 0);
                }
                W.inFar = sib_done ? (W.inFar | bit) : (W.inFar & ~bit);
                W.nearHit &= ~bit;
                W.data = first1 ? d1 : d2;
                W.sides = first1 ? (W.sides & ~bit) : (W.sides | bit);
                stack[(uint32_t)W.depth * stride] = (PathT)(c1 >> 1);
                W.st = (W.data & 0x80000000u) ? 1 : 0;
            }
        }
#if BHRT_FUSED_CLIMB
    } else if (nL > 0) {
#else
    } else if (nL >= nC) {
#endif
        if (W.st == 1) {
            const uint32_t count = ((W.data >> 28) & 7u) + 1, off = W.data & 0x0fffffffu;
            W.r = leaf_hits(M.ltris, off, count, o, d, W.dlen, side, ht, hprim, hfront);
            W.any |= W.r;
            W.st = 2;
        }
#if BHRT_FUSED_CLIMB
    }
    { // (a climb step only once 6 / 12 / 20 lanes wait for one: closed room +0.4 / +1.3 / +3 %)
#else
    } else {
#endif
        if (W.st == 2) {
            WALK_DBG(4, 1); WALK_DBG(5, __popcll(__ballot(true)));
            const int depth = W.depth;
            const uint32_t below = depth >= 32 ? 0xffffffffu : (depth > 0 ? ((1u << depth) - 1u) : 0u);
            const uint32_t waiting = ~W.inFar & below;
            if (!waiting) { W.depth = 0; W.st = 3; }
            else {
                const int l = 32 - __clz((int)waiting);
                const uint32_t upto = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
                W.r = W.r || (W.nearHit & below & ~upto) != 0;
                W.depth = l;
                const uint32_t bit = 1u << (l - 1);
                const uint32_t sib = ((uint32_t)stack[(uint32_t)l * stride] << 1) | (((W.sides >> (l - 1)) & 1u) ^ 1u);
                if (W.r) {
                    WALK_DBG(6, 1); WALK_DBG(7, __popcll(__ballot(true)));
                    W.nearHit |= bit;
                    float tmf;
                    const NodeRec ns = node_at_g(M, sib);
                    const uint32_t ds = ns.data;
                    int fs = W.rf.slow ? -1 : box_fast(ns.b, o, W.rf, ht, tmf);
                    if (fs < 0) fs = box_hit_rcp(ns.b, o, d, ray_rcp(d), ht, tmf) ? 1 : 0;
                    if (fs) { W.inFar |= bit; W.data = ds; W.st = (ds & 0x80000000u) ? 1 : 0; }
                    else W.depth--; // stays in climb
                } else {
                    W.inFar |= bit;
                    W.data = M.bvh[sib].data;
                    W.st = (W.data & 0x80000000u) ? 1 : 0;
                }
            }
        }
    }
}
template <class PathT, bool kLS = false>
__device__ inline bool mesh_closest_vote(const MeshRef &M, V3 o, V3 d, int side, float &ht, int &hprim, int &hfront, PathT *stack, uint32_t stride)
{
    MeshWalk W;
    walk_begin(M, W, o, d, ht);
    while (true) {
        const int nD = __popcll(__ballot(W.st == 0)), nL = __popcll(__ballot(W.st == 1)), nC = __popcll(__ballot(W.st == 2));
        if (nD + nL + nC == 0) break;
        walk_round<PathT, kLS>(M, W, side, ht, hprim, hfront, stack, stride, nD, nL, nC);
    }
    return W.any;
}

// TriObj::ShadowRecursive + TraceBVHShadow (TriObj.cpp:41-66,272-307): pre-order walk (child1 then child2 — the
// evaluation order of `A | B` in the g++ build of the reference), both children visited unless NEITHER box is hit,
// stops at the first leaf that reports a front-face hit; the range test is applied to that hit only.
__device__ inline bool mesh_shadow(const MeshRef &M, V3 o, V3 d, float t_max)
{
    float tm;
    const RayRcp rr = ray_rcp(d);
    if (!box_hit_rcp<false>(node_at(M, 1).b, o, d, rr, BHRT_BIGFLOAT, tm)) return false;
    const float dlen = length(d);
    const RayRcpF rf = ray_rcp_f(d);
    uint32_t cur = 1;
    int depth = 0;
    bool desc = true, found = false;
    float t_min = BHRT_BIGFLOAT;
    while (true) { // single-loop state machine (measured faster than the phased form for any-hit rays)
        if (desc) {
            const uint32_t data = node_data(M, cur);
            if (data & 0x80000000u) {
                const uint32_t count = ((data >> 28) & 7u) + 1, off = data & 0x0fffffffu;
                float ht = BHRT_BIGFLOAT; // fresh HitInfo per leaf (TriObj.cpp:280)
                int hp, hf;
                if (leaf_hits(M.ltris, off, count, o, d, dlen, BHRT_HIT_FRONT, ht, hp, hf)) { found = true; t_min = ht; } // ht = the last accepted t
                if (found) break;
                desc = false;
            } else {
                const uint32_t c1 = data & 0x7fffffffu;
                float t1, t2;
                const NodeRec n1 = node_at(M, c1), n2 = node_at(M, c1 + 1);
                int f1 = rf.slow ? -1 : box_fast_f(n1.b, o, rf, BHRT_BIGFLOAT), f2 = rf.slow ? -1 : box_fast_f(n2.b, o, rf, BHRT_BIGFLOAT);
                if (f1 < 0) f1 = box_hit_rcp<false>(n1.b, o, d, rr, BHRT_BIGFLOAT, t1) ? 1 : 0;
                if (f2 < 0) f2 = box_hit_rcp<false>(n2.b, o, d, rr, BHRT_BIGFLOAT, t2) ? 1 : 0;
                const bool b1 = f1 == 1, b2 = f2 == 1;
                if (!b1 && !b2) desc = false;
                else { depth++; cur = c1; }
            }
        } else {
            if (depth == 0) break;
            if ((cur & 1u) == 0) { cur = cur | 1u; desc = true; } // child1 done -> child2
            else { cur = node_parent(M, cur); depth--; }
        }
    }
    return found && t_min > BHRT_TRI_BIAS && t_min < t_max;
}

// mesh_shadow with the path in LDS (see mesh_closest_vote): after a subtree the walk continues at the second child of the
// deepest level still in its first child, found by a bit scan; no parent links are read.
template <class PathT, bool kLS = false>
__device__ inline bool mesh_shadow_stack(const MeshRef &M, V3 o, V3 d, float t_max, PathT *stack, uint32_t stride)
{
    float tm;
    const RayRcpF rf = ray_rcp_f(d);
    const NodeRec root = node_at(M, 1);
    int st = 3; // 0 inner node, 1 leaf, 2 climb, 3 done — the wave runs the phase most of its lanes wait for (see mesh_closest_vote)
    uint32_t data = root.data;
    if (box_hit_lazy<false>(root.b, o, d, rf, BHRT_BIGFLOAT, tm)) st = (data & 0x80000000u) ? 1 : 0;
    const float dlen = length(d);
    uint32_t inSecond = 0;
    int depth = 0;
    bool found = false;
    float t_min = BHRT_BIGFLOAT;
    while (true) {
        const int nI = __popcll(__ballot(st == 0)), nL = __popcll(__ballot(st == 1)), nC = __popcll(__ballot(st == 2));
        if (nI + nL + nC == 0) break;
        if (nI >= nL && nI >= nC) {
            if (st == 0) {
                const uint32_t c1 = data & 0x7fffffffu;
                float t1, t2;
                const NodeRec n1 = node_at(M, c1), n2 = node_at(M, c1 + 1);
                float a1 = 0.f, x1 = 0.f, a2 = 0.f, x2 = 0.f;
                int f1 = rf.slow ? -1 : box_fast_f(n1.b, o, rf, BHRT_BIGFLOAT, a1, x1), f2 = rf.slow ? -1 : box_fast_f(n2.b, o, rf, BHRT_BIGFLOAT, a2, x2);
                const bool m1 = f1 == 0, m2 = f2 == 0; // missed, by the approximate test's own verdict (leaf_skip)
                if (f1 < 0 || f2 < 0) { // the exact form: its double reciprocals only here (kept across the loop they cost twelve registers)
                    const RayRcp rr = ray_rcp(d);
                    if (f1 < 0) f1 = box_hit_rcp<false>(n1.b, o, d, rr, BHRT_BIGFLOAT, t1) ? 1 : 0;
                    if (f2 < 0) f2 = box_hit_rcp<false>(n2.b, o, d, rr, BHRT_BIGFLOAT, t2) ? 1 : 0;
                }
                if (f1 != 1 && f2 != 1) st = 2;
                else {
                    // TraceBVHShadow visits both children once either box is hit; a box-missed inner child ends at its own two box
                    // tests (skip_missed): child 1 is stepped over, child 2 marked done.  A box-missed LEAF child likewise when its visit
                    // provably accepts nothing (leaf_skip).
                    depth++;
                    const uint32_t bit = 1u << (depth - 1);
                    const bool over1 = skip_missed(M, f1 == 1, n1.data) || (kLS && M.omax > 0.f && m1 && (n1.data & 0x80000000u) && leaf_skip(M, n1.parent, a1, x1, o, d, rf, dlen));
                    const bool over2 = skip_missed(M, f2 == 1, n2.data) || (kLS && M.omax > 0.f && m2 && (n2.data & 0x80000000u) && leaf_skip(M, n2.parent, a2, x2, o, d, rf, dlen));
                    inSecond = (over1 || over2) ? (inSecond | bit) : (inSecond & ~bit);
                    stack[(uint32_t)depth * stride] = (PathT)(c1 >> 1);
                    data = over1 ? n2.data : n1.data;
                    st = (data & 0x80000000u) ? 1 : 0;
                }
            }
        } else if (nL >= nC) {
            if (st == 1) {
                const uint32_t count = ((data >> 28) & 7u) + 1, off = data & 0x0fffffffu;
                float ht = BHRT_BIGFLOAT; // fresh HitInfo per leaf (TriObj.cpp:280)
                int hp, hf;
                if (leaf_hits(M.ltris, off, count, o, d, dlen, BHRT_HIT_FRONT, ht, hp, hf)) { found = true; t_min = ht; } // ht = the last accepted t
                st = found ? 3 : 2;
            }
        } else {
            if (st == 2) {
                const uint32_t below = depth >= 32 ? 0xffffffffu : (depth > 0 ? ((1u << depth) - 1u) : 0u);
                const uint32_t waiting = ~inSecond & below;
                if (!waiting) st = 3;
                else {
                    depth = 32 - __clz((int)waiting);
                    inSecond |= 1u << (depth - 1);
                    data = node_data(M, ((uint32_t)stack[(uint32_t)depth * stride] << 1) | 1u);
                    st = (data & 0x80000000u) ? 1 : 0;
                }
            }
        }
    }
    return found && t_min > BHRT_TRI_BIAS && t_min < t_max;
}

// Sphere::IntersectRay accept decision (Sphere.cpp:8-50)
__device__ inline bool sphere_hit(V3 oc, V3 dir, int side, float t_cur, float &t_out, int &front_out)
{
    float A = dot(dir, dir);
    float B = 2 * dot(dir, oc);
    float C = dot(oc, oc) - 1;
    float DD = B * B - 4 * A * C;
    if (!(DD > 0)) return false;
    float sq = sqrtf(DD);
    float t1 = (-B + sq) / (2 * A);
    float t2 = (-B - sq) / (2 * A);
    float t = BHRT_BIGFLOAT;
    bool hitFront = true;
    if (t1 < 0 && t2 < 0) return false;
    else if (t1 * t2 <= 0) {
        if (side == BHRT_HIT_FRONT) return false;
        t = t1;
        hitFront = false;
    } else if (t1 > 0 && t2 > 0) {
        if (side == BHRT_HIT_FRONT || side == BHRT_HIT_FRONT_AND_BACK) { t = t2; hitFront = true; }
        else if (side == BHRT_HIT_BACK) { t = t1; hitFront = false; }
    }
    if (t_cur < t || t <= 0) return false;
    t_out = t;
    front_out = hitFront ? 1 : 0;
    return true;
}
// Plane::IntersectRay accept decision (Plane.cpp:8-34)
__device__ inline bool plane_hit(V3 p, V3 d, int side, float t_cur, float &t_out, int &front_out)
{
    if (d.z == 0.0f) return false;
    float t = -p.z / d.z;
    if (t <= 0 || t > t_cur) return false;
    V3 x = p + t * d;
    if (x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1) return false;
    bool hitFront = (dot(-d, v3(0, 0, 1)) > 0);
    if (!hitFront && side == BHRT_HIT_FRONT) return false;
    else if (hitFront && side == BHRT_HIT_BACK) return false;
    t_out = t;
    front_out = hitFront ? 1 : 0;
    return true;
}

// recursive(&rootNode, ...) (Main.cpp:389-413): nodes in DFS pre-order, "t > hit" rejects so a later
// equal-t candidate wins.
// LDS nodelet: the workgroup copies the first BHRT_LDS_NODES nodes (top levels, breadth-first numbering) of the mesh it is
// about to traverse into LDS.  Uniform control flow is required (every thread of the block reaches the barriers): the
// scene-graph loop below is uniform — `active` only predicates the per-ray work.
#ifndef BHRT_LDS_NODES
#define BHRT_LDS_NODES 512 /* 16 KB per workgroup: 9 full levels */
#endif
__device__ inline void stage_nodelet(const DevScene &S, int mesh, bhrt_bvh_node *lds, MeshRef &M, uint32_t lds_nodes = BHRT_LDS_NODES /* even */)
{
    M = mesh_ref(S, mesh);
    const uint32_t nn = S.meshes[mesh].n_bvh_nodes;
    const uint32_t cnt = nn < lds_nodes ? nn : lds_nodes;
    __syncthreads(); // previous mesh's nodelet no longer in use
    const float4 *src = (const float4 *)M.bvh;
    float4 *dst = (float4 *)lds;
    for (uint32_t k = threadIdx.x; k < cnt * 2; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
    M.lds = lds;
    M.n_lds = cnt;
}

// recursive() (Main.cpp:389-413) over the flattened scene graph.
// lds == nullptr: no staging (any thread may call it alone); otherwise ALL threads of the block must call it together.
// start > 0: resume at scene node `start` with the hit so far in h (k_trace_mesh).  park = true: stop at the first mesh
// whose root box the ray hits and return that node's index (the caller parks the ray there); -1 = ran to the end.
// park_key (park only): coherence key of the parked ray = Morton cell of its entry point into the mesh's box + direction octant.
__device__ inline uint32_t park_spread(uint32_t v) // low bits -> every third bit
{
    uint32_t r = 0;
    for (int k = 0; k < BHRT_PARK_CELL_BITS; k++) r |= ((v >> k) & 1u) << (3 * k);
    return r;
}
// kMeshes = false: the scene has no mesh node; the BVH code is not compiled in (half the registers: 8 waves per SIMD instead of
// 5, C2 trace 1.13 -> 0.90 ms).
// path != nullptr: this lane's column of the LDS path stack (stride path_stride, 33 rows) -> mesh_closest_vote; the caller has
// checked that every mesh qualifies.  lds_nodes: size of the nodelet buffer behind `lds`.
// park_slow (park only): set when the parked ray has a zero direction component in the mesh's space.  Box::IntersectRay leaves such an
// axis out (Box.cpp:13-28), so a ray parallel to one coordinate axis "hits" every box whose extent on that axis it
// crosses — the reference walks (nearly) the whole tree for it, and so does the traversal here: ~10^5 sequential rounds for the
// 100 k-triangle mesh.  The samplers produce such rays at a rate of ~1.5e-8 per diffuse GI ray off an axis-aligned wall (a 31-bit draw
// below 2^-25 makes theta exactly 0 in GetSampleInSemiSphere, MtlBlinn.cpp:697-716: the ray leaves along the normal).
template <bool kMeshes = true, class PathT = uint16_t, bool kLS = false>
__device__ inline int trace_closest(const DevScene &S, V3 o, V3 d, int side, Hit &h, bool active = true, bhrt_bvh_node *lds = nullptr, int start = 0,
                                    bool park = false, uint32_t *park_key = nullptr, PathT *path = nullptr, uint32_t path_stride = 0,
                                    uint32_t lds_nodes = BHRT_LDS_NODES, bool camera = false /* o = the camera position */, bool *park_slow = nullptr)
{
    if (start == 0) { h.t = BHRT_BIGFLOAT; h.node = -1; h.prim = -1; h.front = 1; }
    int parked = -1;
    for (int n = 0; n < S.n_nodes; n++) {
        const int type = S.nodes[n].obj_type;
        if (type == BHRT_OBJ_NONE) continue;
        if (!kMeshes && type == BHRT_OBJ_MESH) continue; // cannot happen: the host picks kMeshes from the scene
        MeshRef M;
        if (kMeshes && type == BHRT_OBJ_MESH) {
            if (lds) stage_nodelet(S, S.nodes[n].mesh, lds, M, lds_nodes);
            else M = mesh_ref(S, S.nodes[n].mesh);
        }
        if (!active || n < start || parked >= 0) continue;
        V3 lp = o, ld = d;
        if (camera) local_ray_camera(S, n, lp, ld);
        else local_ray(S, n, lp, ld);
        float t;
        int fr;
        if (type == BHRT_OBJ_SPHERE) {
            if (sphere_hit(lp, ld, side, h.t, t, fr)) { h.t = t; h.node = n; h.prim = -1; h.front = fr; }
        } else if (type == BHRT_OBJ_PLANE) {
            if (plane_hit(lp, ld, side, h.t, t, fr)) { h.t = t; h.node = n; h.prim = -1; h.front = fr; }
        } else if (!kMeshes) {
        } else if (park) {
            float tm; // the root box gate of TriObj::IntersectRay (TriObj.cpp:17-39), repeated by mesh_closest on resume
            const NodeRec root = node_at(M, 1);
            if (box_hit_lazy(root.b, lp, ld, ray_rcp_f(ld), h.t, tm)) {
                parked = n;
                if (park_slow) *park_slow = ld.x == 0 || ld.y == 0 || ld.z == 0;
                if (park_key) { // ordering hint only: any value is correct
                    const V3 e = tm > 0 ? lp + tm * ld : lp;
                    const float nn = (float)(1 << BHRT_PARK_CELL_BITS), m = nn - 1.f;
                    const float cx = fminf(fmaxf((e.x - root.b[0]) / (root.b[3] - root.b[0]) * nn, 0.f), m);
                    const float cy = fminf(fmaxf((e.y - root.b[1]) / (root.b[4] - root.b[1]) * nn, 0.f), m);
                    const float cz = fminf(fmaxf((e.z - root.b[2]) / (root.b[5] - root.b[2]) * nn, 0.f), m);
                    const uint32_t oct = (ld.x < 0 ? 1u : 0u) | (ld.y < 0 ? 2u : 0u) | (ld.z < 0 ? 4u : 0u);
                    *park_key = (oct << (3 * BHRT_PARK_CELL_BITS)) | park_spread((uint32_t)cx) | (park_spread((uint32_t)cy) << 1) | (park_spread((uint32_t)cz) << 2);
                }
            }
        } else {
            if (path ? mesh_closest_vote<PathT, kLS>(M, lp, ld, side, h.t, h.prim, h.front, path, path_stride) : mesh_closest(M, lp, ld, side, h.t, h.prim, h.front)) h.node = n;
        }
    }
    return parked;
}

// recursive() for ONE ray held by every lane of a wave: meshes through mesh_closest_coop (k_trace_slow)
__device__ inline void trace_closest_coop(const DevScene &S, CoopLds &L, V3 o, V3 d, int side, Hit &h)
{
    h.t = BHRT_BIGFLOAT; h.node = -1; h.prim = -1; h.front = 1;
    for (int n = 0; n < S.n_nodes; n++) {
        const int type = S.nodes[n].obj_type;
        if (type == BHRT_OBJ_NONE) continue;
        V3 lp = o, ld = d;
        local_ray(S, n, lp, ld);
        float t;
        int fr;
        if (type == BHRT_OBJ_SPHERE) {
            if (sphere_hit(lp, ld, side, h.t, t, fr)) { h.t = t; h.node = n; h.prim = -1; h.front = fr; }
        } else if (type == BHRT_OBJ_PLANE) {
            if (plane_hit(lp, ld, side, h.t, t, fr)) { h.t = t; h.node = n; h.prim = -1; h.front = fr; }
        } else if (mesh_closest_coop(mesh_ref(S, S.nodes[n].mesh), L, lp, ld, side, h.t, h.prim, h.front)) h.node = n;
    }
}

// GenLight::Shadow (GenLight.cpp:10-69).  The result is an OR over per-node tests that do not influence each
// other, so nodes are tested in index order; each test is the reference's (including its quirks Q1-Q3).
// kMode 3: the scene has no mesh (BVH code not compiled in).  kMode 0: everything.  kMode 1: spheres and planes only; returns 2.f instead of 1.f when the ray also hits the root
// box of a mesh (the caller parks it for k_shadow_mesh).  kMode 2: the meshes only.
template <int kMode, class PathT = uint16_t, bool kLS = false>
__device__ inline float trace_shadow_t(const DevScene &S, V3 o, V3 d, float t_max, PathT *path = nullptr, uint32_t path_stride = 0)
{
    V3 rp = o, rd = d;
    to_node_identity(rp, rd); // rootNode's own ToNodeCoords
    bool wants_mesh = false;
    for (int n = 0; n < S.n_nodes; n++) {
        const int type = S.nodes[n].obj_type;
        if (type == BHRT_OBJ_NONE) continue;
        if (kMode == 2 && type != BHRT_OBJ_MESH) continue;
        if (kMode == 3 && type == BHRT_OBJ_MESH) continue; // cannot happen
        const int depth = S.nodes[n].depth;
        const int32_t *ch = S.chain + (size_t)n * BHRT_MAX_NODE_DEPTH;
        V3 pp = rp, pd = rd; // ray in the PARENT's space (used by the plane test, Q1)
        for (int k = 0; k + 1 < depth; k++) to_node(S.nodes[ch[k]].xf, pp, pd);
        V3 lp = pp, ld = pd;
        to_node(S.nodes[n].xf, lp, ld);
        if (type == BHRT_OBJ_SPHERE) {
            float A = dot(ld, ld);
            float B = 2 * dot(ld, lp);
            float C = dot(lp, lp) - 1;
            float DD = B * B - 4 * A * C;
            if (DD > 0) {
                float sq = sqrtf(DD);
                float t1 = (-B + sq) / (2 * A);
                float t2 = (-B - sq) / (2 * A);
                float t = fmin_cy(t1, t2);
                if (!(t < 0) && t < t_max && t > BHRT_SHADOW_BIAS) return 0.f;
            }
        } else if (type == BHRT_OBJ_PLANE) {
            float t = -lp.z / ld.z;
            if (!(t < 0)) {
                V3 x = pp + t * pd; // un-transformed ray
                if (!(x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1))
                    if (t < t_max && t > BHRT_SHADOW_BIAS) return 0.f;
            }
        } else if (kMode == 3) {
        } else if (kMode == 1) {
            float tm; // the root box gate of TriObj::ShadowRecursive (TriObj.cpp:41-54), repeated by mesh_shadow
            if (!wants_mesh && box_hit_lazy<false>(node_at(mesh_ref(S, S.nodes[n].mesh), 1).b, lp, ld, ray_rcp_f(ld), BHRT_BIGFLOAT, tm)) wants_mesh = true;
        } else {
            if (path ? mesh_shadow_stack<PathT, kLS>(mesh_ref(S, S.nodes[n].mesh), lp, ld, t_max, path, path_stride) : mesh_shadow(mesh_ref(S, S.nodes[n].mesh), lp, ld, t_max)) return 0.f;
        }
    }
    return wants_mesh ? 2.f : 1.f;
}
__device__ inline float trace_shadow(const DevScene &S, V3 o, V3 d, float t_max) { return trace_shadow_t<0>(S, o, d, t_max); }

} // namespace bhrt
