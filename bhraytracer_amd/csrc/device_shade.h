// device_shade.h — device-side hit-attribute reconstruction, textures, samplers and the Blinn
// integrator pieces used by the wavefront shading kernel.
//
// Follows (paths relative to /root/reference/BHRayTracer):
//   hit attributes        Objects/Sphere/Sphere.cpp:51-72, Objects/Plane/Plane.cpp:36-74,
//                         Objects/TriObj/TriObj.cpp:157-186, Scenes/scene.h:497-501, Main.cpp:401,407-412
//   textures              Scenes/scene.h:318-337,344-354,364-422, Textures/Texture.cpp:97-136
//   samplers, Rnd01       Materials/Blinn/MtlBlinn.cpp:42-49,591-716
//   lights                Lights/PointLight.cpp:7-18, Lights/lights.h:29-87
// Transcendentals come from bhrt_detmath.h so that the CPU oracle (device-math mode) is bit-comparable.
#pragma once
#include "bhrt_detmath.h"
#include "bhrt_rng.h"
#include "device_trace.h"

namespace bhrt {

#define BHRT_PI_D 3.14159265 /* the reference's PI macro (a double) */
#define BHRT_BIAS 0.0001f    /* MtlBlinn.cpp:10 */
#define BHRT_EULER 2.7182818f
#define BHRT_MAXLOOP (1 << 20) /* cap on the reference's unbounded rejection loops (same cap in the oracle) */

struct Attr {
    V3 p, N, uvw, du, dv;
};

__device__ inline void from_node(const bhrt_xform &t, V3 &p, V3 &N) // Node::FromNodeCoords, scene.h:497-501
{
    p = mat_mul(t.tm, p) + ld3(t.pos);
    N = normalized(mat_tmul(t.itm, N));
}
__device__ inline void from_node_identity(V3 &p, V3 &N)
{
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    p = mat_mul(I, p) + v3(0, 0, 0);
    N = normalized(mat_tmul(I, N));
}

// Rebuild the HitInfo the reference's intersectors fill, from the compact record (t, node, prim).
// need_uv = false skips texture coordinates and ray differentials (only textures read them).
__device__ inline void hit_attrs(const DevScene &S, V3 o, V3 d, float t, int node, int prim, bool need_uv, Attr &a)
{
    V3 lp = o, ld = d;
    local_ray(S, node, lp, ld);
    const bhrt_node &nd = S.nodes[node];
    a.uvw = v3(0.5f, 0.5f, 0.5f);
    a.du = v3(0, 0, 0);
    a.dv = v3(0, 0, 0);
    if (nd.obj_type == BHRT_OBJ_SPHERE) {
        a.p = lp + t * ld;
        a.N = a.p;
        if (need_uv) {
            V3 dn = normalized(a.N);
            a.uvw.x = (float)(0.5f + dm::atan2f_(dn.y, dn.x) / (2 * BHRT_PI_D));
            a.uvw.y = (float)(0.5f - dm::asinf_(dn.z) / (BHRT_PI_D));
            a.uvw.z = 0;
        }
    } else if (nd.obj_type == BHRT_OBJ_PLANE) {
        a.p = lp + t * ld;
        a.N = v3(0, 0, 1);
        if (need_uv) {
            a.uvw = v3((1 + a.p.x) / 2.f, (1 + a.p.y) / 2.f, 0);
            // ray differentials from the CAMERA's dd_x / dd_y (Plane.cpp:51-70)
            V3 ddx = ld3(S.cam.dd_x), ddy = ld3(S.cam.dd_y);
            V3 nd_ = normalized(ld);
            float scaled_t = length(t * ld);
            float nn = dot(nd_, nd_);
            float pw = dm::powf_(nn, 1.5f);
            V3 dDX = (nn * ddx - dot(nd_, ddx) * nd_) / pw;
            V3 dDY = (nn * ddy - dot(nd_, ddy) * nd_) / pw;
            float d_t_x = -(0 + scaled_t * dot(dDX, a.N) / dot(nd_, a.N));
            float d_t_y = -(0 + scaled_t * dot(dDY, a.N) / dot(nd_, a.N));
            V3 hx = (scaled_t * dDX + v3(0, 0, 0)) + d_t_x * nd_;
            V3 hy = (scaled_t * dDY + v3(0, 0, 0)) + d_t_y * nd_;
            a.du = hx / 2.f;
            a.dv = hy / 2.f;
        }
    } else { // triangle: TriObj.cpp:157-186
        const bhrt_mesh &m = S.meshes[nd.mesh];
        const bhrt_tri &tr = ((const bhrt_tri *)(S.blob + m.off_tris))[prim];
        V3 vX = lp + t * ld;
        float a0, a1, a2;
        tri_areas(tr, vX, a0, a1, a2);
        float asum = a0 + a1 + a2;
        float bx = a0 / asum, by = a1 / asum, bz = a2 / asum;
        const float *vn = (const float *)(S.blob + m.off_vn);
        const uint32_t *fn = (const uint32_t *)(S.blob + m.off_fn) + 3 * (size_t)prim;
        a.N = ld3(vn + 3 * (size_t)fn[0]) * bx + ld3(vn + 3 * (size_t)fn[1]) * by + ld3(vn + 3 * (size_t)fn[2]) * bz; // cyTriMesh.h:191
        a.p = vX;
        if (need_uv) {
            const float *vt = (const float *)(S.blob + m.off_vt);
            const uint32_t *ft = (const uint32_t *)(S.blob + m.off_ft) + 3 * (size_t)prim;
            a.uvw = ld3(vt + 3 * (size_t)ft[0]) * bx + ld3(vt + 3 * (size_t)ft[1]) * by + ld3(vt + 3 * (size_t)ft[2]) * bz;
        }
    }
    // Main.cpp:401 (the node's own transform) then Main.cpp:407-412 (its parent's, or rootNode's identity)
    from_node(nd.xf, a.p, a.N);
    if (nd.parent >= 0) from_node(S.nodes[nd.parent].xf, a.p, a.N);
    else from_node_identity(a.p, a.N);
}

// ---------------------------------------------------------------------------------------------- textures
__device__ inline V3 tile_clamp(V3 uvw) // scene.h:344-354
{
    V3 u;
    u.x = uvw.x - (int)uvw.x; u.y = uvw.y - (int)uvw.y; u.z = uvw.z - (int)uvw.z;
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}
__device__ inline V3 texel(const uint8_t *data, int width, int yy, int xx)
{
    const uint8_t *p = data + 3 * ((size_t)yy * width + xx);
    return v3(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f); // Color24::ToColor
}
__device__ inline V3 tex_sample(const DevScene &S, const bhrt_texture &t, V3 uvw)
{
    if (t.type == BHRT_TEX_CHECKER) { // Texture.cpp:127-136
        V3 u = tile_clamp(uvw);
        V3 c1 = ld3(t.color1), c2 = ld3(t.color2);
        if (u.x <= 0.5f) return u.y <= 0.5f ? c1 : c2;
        return u.y <= 0.5f ? c2 : c1;
    }
    int width = t.width, height = t.height; // Texture.cpp:97-123
    if (width + height == 0) return v3(0, 0, 0);
    const uint8_t *data = S.blob + t.off_data;
    V3 u = tile_clamp(uvw);
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
    return texel(data, width, iy, ix) * ((1 - fx) * (1 - fy)) + texel(data, width, iy, ixp) * (fx * (1 - fy)) +
           texel(data, width, iyp, ix) * ((1 - fx) * fy) + texel(data, width, iyp, ixp) * (fx * fy);
}
__device__ inline V3 xform_to(const bhrt_xform &t, V3 p) { return mat_mul(t.itm, p - ld3(t.pos)); } // scene.h:220
// TexturedColor::Sample(uvw) (scene.h:410 -> :371)
__device__ inline V3 tc_sample(const DevScene &S, const bhrt_texcolor &tc, V3 uvw)
{
    V3 c = ld3(tc.color);
    if (tc.map < 0) return c;
    const bhrt_texmap &m = S.texmaps[tc.map];
    if (m.texture < 0) return c * v3(0, 0, 0);
    return c * tex_sample(S, S.textures[m.texture], xform_to(m.xf, uvw));
}
// Texture::Sample's 31 footprint taps (scene.h:318-337) of a CHECKER texture, one coordinate: do they all fall into the half-tile of tap 0?
// A tap's coordinate is fl(fl(x + fl(tx * a)) + fl(ty * b)) with |tx|, |ty| <= 0.5 (r = sqrt(halton) / 2 times a sine / cosine: the table of
// DevScene::tapx / tapy, checked at upload), i.e. within e = (|a| + |b|) / 2 of x plus two roundings of at most 2^-24 (|x| + e) each;
// TextureChecker::Sample (Texture.cpp:127-136) then takes the fractional part (TileClamp: exact below 2^23, + 1 for negatives with another
// 2^-24) and compares it with 0.5.  With w = e (1 + 2^-20) + 2^-21 (|x| + e) + 2^-22 — every term above with room to spare, and the
// rounding of this very expression — all taps lie strictly inside (x - w, x + w); when that interval holds no multiple of 0.5 every tap
// compares like tap 0.  NaN, infinities and huge coordinates answer false (the taps are then evaluated one by one).
__device__ inline bool checker_taps_in_one_cell(float x, float a, float b)
{
    const float e = 0.5f * fabsf(a) + 0.5f * fabsf(b);
    const float w = e * 1.00000095367431640625f + 4.76837158203125e-07f * (fabsf(x) + e) + 2.384185791015625e-07f;
    const float lo = x - w, hi = x + w;
    return fabsf(x) + w < 4194304.f && floorf(2.f * lo) == floorf(2.f * hi) && 2.f * lo != floorf(2.f * lo);
}
// TexturedColor::Sample(uvw, duvw) (scene.h:411 -> :372-380 -> :318-337)
__device__ inline V3 tc_sample_d(const DevScene &S, const bhrt_texcolor &tc, V3 uvw, V3 du, V3 dv)
{
    V3 c = ld3(tc.color);
    if (tc.map < 0) return c;
    const bhrt_texmap &m = S.texmaps[tc.map];
    if (m.texture < 0) return c * v3(0, 0, 0);
    const bhrt_texture &t = S.textures[m.texture];
    V3 u = xform_to(m.xf, uvw);
    V3 d0 = xform_to(m.xf, du + uvw) - u;
    V3 d1 = xform_to(m.xf, dv + uvw) - u;
    V3 s = tex_sample(S, t, u);
    if (length_sq(d0) + length_sq(d1) == 0) return c * s;
    if (t.type == BHRT_TEX_CHECKER && checker_taps_in_one_cell(u.x, d0.x, d1.x) && checker_taps_in_one_cell(u.y, d0.y, d1.y)) {
        const V3 s0 = s; // every tap returns the colour of tap 0: the same 31 float additions, without the 31 tap positions and lookups
        for (int i = 1; i < 32; i++) s = s + s0;
        return c * (s / float(32));
    }
    for (int i = 1; i < 32; i++) s = s + tex_sample(S, t, u + S.tapx[i] * d0 + S.tapy[i] * d1);
    return c * (s / float(32));
}
__device__ inline V3 sample_environment(const DevScene &S, const bhrt_texcolor &tc, V3 dir) // scene.h:414-420
{
    float z = dm::asinf_(-dir.z) / float(M_PI) + 0.5f;
    float x = dir.x / (fabsf(dir.x) + fabsf(dir.y));
    float y = dir.y / (fabsf(dir.x) + fabsf(dir.y));
    return tc_sample(S, tc, v3(0.5f, 0.5f, 0.0f) + z * (x * v3(0.5f, 0.5f, 0) + y * v3(-0.5f, 0.5f, 0)));
}

// ---------------------------------------------------------------------------------------------- RNG + samplers
struct DRng {
    uint32_t key, ctr;
    // wrap: the bits of the counter that advance.  All of them for a (pixel, sample, path, section) stream; the low 16 for a photon emission, which
    // owns a 2^16-draw window of its key (bhrt_photon_stream): a degenerate path whose rejection loops run to BHRT_MAXLOOP then re-reads its OWN
    // window instead of running through the windows of the emissions behind it — two emissions never read the same (key, counter) pair.
    uint32_t wrap = 0xffffffffu;
    __device__ int rand() { const uint32_t c = ctr; ctr = (c & ~wrap) | ((c + 1u) & wrap); return bhrt_rand31(key, c); }
    __device__ float rnd01() // MtlBlinn.cpp:42-49
    {
        float r = dm::rand_to_unit(rand());
        int guard = 0;
        while ((r == 0.0f || r == 1.0f) && guard++ < BHRT_MAXLOOP) r = dm::rand_to_unit(rand());
        return r;
    }
};
__device__ inline V3 clamp_white(V3 c) // ClampColorToWhite, MtlBlinn.cpp:79-83
{
    if (c.x > 1) c.x = 1.f;
    if (c.y > 1) c.y = 1.f;
    if (c.z > 1) c.z = 1.f;
    return c;
}
__device__ inline bool isnan_f(float x) { return x != x; }
__device__ inline float acos_safe(float v) { return dm::acosf_(fmin_cy(1.f, fmax_cy(-1.f, v))); } // cyCore.h:191-193

__device__ inline V3 random_crossing_vector(DRng &g, V3 V) // MtlBlinn.cpp:591-600
{
    V3 r = v3(0, 0, 1);
    int guard = 0;
    while (is_zero(cross(V, r)) && guard++ < BHRT_MAXLOOP) {
        float c = g.rnd01(), b = g.rnd01(), a = g.rnd01(); // g++ evaluates Vec3f(Rnd01(),Rnd01(),Rnd01()) right to left
        r = v3(a, b, c);
    }
    return r;
}
__device__ inline V3 sample_along_normal(DRng &g, V3 N, float R) // MtlBlinn.cpp:602-617
{
    float r = g.rnd01();
    r = sqrtf(r) * R;
    float theta = (float)(g.rnd01() * 2 * BHRT_PI_D);
    float x = r * dm::cosf_(theta), y = r * dm::sinf_(theta);
    V3 axis1 = cross(random_crossing_vector(g, N), N);
    V3 axis2 = cross(axis1, N);
    return N + normalized(axis1) * x + normalized(axis2) * y;
}
__device__ inline V3 sample_along_light_direction(DRng &g, V3 N, float glossiness, float &o_theta) // MtlBlinn.cpp:619-635
{
    float u = g.rnd01();
    float weightTheta = acos_safe(dm::powf_(u, 1.f / (glossiness + 1.f)));
    o_theta = weightTheta;
    float R = dm::tanf_(weightTheta);
    float phi = (float)(g.rnd01() * 2 * BHRT_PI_D);
    float x = R * dm::cosf_(phi), y = R * dm::sinf_(phi);
    V3 axis1 = cross(random_crossing_vector(g, N), N);
    V3 axis2 = cross(axis1, N);
    return N + normalized(axis1) * x + normalized(axis2) * y;
}
__device__ inline V3 sample_in_semi_sphere(DRng &g, V3 N, float &o_theta) // MtlBlinn.cpp:697-716
{
    for (int guard = 0; guard < BHRT_MAXLOOP; guard++) {
        V3 axisY = normalized(cross(N, random_crossing_vector(g, N)));
        V3 axisX = cross(N, axisY);
        float phi = (float)(g.rnd01() * 2 * BHRT_PI_D);
        float rnd = g.rnd01();
        float theta = 0.5f * acos_safe(1 - 2 * rnd);
        o_theta = theta;
        float sinTheta = dm::sinf_(theta);
        V3 ret = (sinTheta * dm::cosf_(phi)) * axisX + (sinTheta * dm::sinf_(phi)) * axisY + dm::cosf_(theta) * N;
        if (dot(N, ret) <= 0) continue;
        return ret;
    }
    return N;
}
// The draws and the lobe angle of GetSampleAlongLightDirection without the direction itself (see gi_direction)
struct LobeDraw { float theta, phi; V3 rc; };
__device__ inline LobeDraw draw_along_light_direction(DRng &g, V3 N, float glossiness)
{
    LobeDraw d;
    const float u = g.rnd01();
    d.theta = acos_safe(dm::powf_(u, 1.f / (glossiness + 1.f)));
    d.phi = (float)(g.rnd01() * 2 * BHRT_PI_D);
    d.rc = random_crossing_vector(g, N);
    return d;
}
__device__ inline V3 build_along_light_direction(V3 N, const LobeDraw &d)
{
    const float R = dm::tanf_(d.theta);
    const float x = R * dm::cosf_(d.phi), y = R * dm::sinf_(d.phi);
    const V3 axis1 = cross(d.rc, N);
    const V3 axis2 = cross(axis1, N);
    return N + normalized(axis1) * x + normalized(axis2) * y;
}
// GIUseSpecularDirOrDiffuseDir, MtlBlinn.cpp:354-378.  The reference builds the diffuse and the specular candidate and keeps
// one; the choice depends only on the two lobe angles and one more draw.  All draws are made in the reference's order, the
// specular direction (tan, sin, cos, two normalisations) is only built when it is the one kept, the diffuse one (built anyway:
// its rejection loop tests it) only normalised then: a wave of diffuse-surface hits skips ~300 instructions.
__device__ inline V3 gi_direction(DRng &g, bool &useSpecular, V3 vN, V3 vV, float kd, float ks, float glossiness)
{
    float diffuseTheta = 0;
    const V3 diffuseRaw = sample_in_semi_sphere(g, vN, diffuseTheta);
    float p_diffuseTheta = dm::sinf_(2 * diffuseTheta);
    float cosvVvN = dot(vN, vV);
    V3 vR = (2 * cosvVvN) * vN - vV;
    const LobeDraw lobe = draw_along_light_direction(g, vR, glossiness);
    float p_specularTheta = dm::powf_(dm::cosf_(lobe.theta), glossiness);
    float P_Diffuse = kd * p_diffuseTheta;
    float P_sum = P_Diffuse + ks * p_specularTheta;
    float P_Diffuse_Norm = P_Diffuse / P_sum;
    float rnd = g.rnd01();
    useSpecular = rnd >= P_Diffuse_Norm;
    if (useSpecular) return build_along_light_direction(vR, lobe);
    return normalized(diffuseRaw);
}
__device__ inline float tc_max(const float *c) { return fmax_cy(fmax_cy(c[0], c[1]), c[2]); } // GetKD/GetKS, MtlBlinn.cpp:68-69
__device__ inline float gray3(const float *c) { return (c[0] + c[1] + c[2]) / 3.0f; }
// GetSampleInLight, MtlBlinn.cpp:637-695.  The reference builds both candidate directions (Phong lobe around L, disk of the
// light) and keeps one.  Which one is kept depends only on scalars (the lobe angle, the disk radius, one more draw), so the
// draws and scalars of both are evaluated in the reference's order and ONE direction is then constructed from the
// selected (axis, x, y, crossing vector): the same operations on the kept candidate, ~200 instructions less per call.
__device__ inline V3 sample_in_light(DRng &g, const float *diffuseColor, const float *specularColor, const bhrt_light &light, V3 hitP, float glossiness)
{
    if (light.type == BHRT_LIGHT_POINT) {
        float kd = tc_max(diffuseColor), ks = tc_max(specularColor);
        V3 vL = ld3(light.vec) - hitP;
        // ---- candidate 1: GetSampleAlongLightDirection(L normalised, glossiness) — draws u, phi, (crossing vector)
        const V3 nL = normalized(vL);
        const float u = g.rnd01();
        const float diffuseTheta = acos_safe(dm::powf_(u, 1.f / (glossiness + 1.f)));
        const float Rd = dm::tanf_(diffuseTheta);
        const float phi_d = (float)(g.rnd01() * 2 * BHRT_PI_D);
        const V3 rc_d = random_crossing_vector(g, nL);
        bool useSpecular = false;
        float Rs = 0, theta_s = 0;
        V3 rc_s = v3(0, 0, 1);
        if (!(ks == 0 && kd != 0)) {
            // ---- candidate 2: point on the light's disk — draws r, theta, (crossing vector)
            const float r = g.rnd01();
            Rs = sqrtf(r) * (int)light.size; // GetSize() truncates to int (lights.h:76)
            theta_s = (float)(g.rnd01() * 2 * BHRT_PI_D);
            rc_s = random_crossing_vector(g, vL);
            const float p_specular = 2 * r / (Rs * Rs);
            if (ks != 0 && kd == 0) useSpecular = true;
            else {
                const float p_diffuse = dm::powf_(dm::cosf_(diffuseTheta), glossiness); // no draws: evaluated only where the choice needs it
                const float P_Diffuse = kd * p_diffuse;
                const float P_Specular = ks * p_specular;
                const float P_sum = P_Diffuse + P_Specular;
                const float P_Diffuse_Norm = P_Diffuse / P_sum;
                const float rnd = g.rnd01();
                useSpecular = rnd >= P_Diffuse_Norm;
            }
        }
        // ---- the kept candidate: axis + unit(axis1) * x + unit(axis2) * y, normalised
        const V3 axis = useSpecular ? vL : nL, rc = useSpecular ? rc_s : rc_d;
        const float R = useSpecular ? Rs : Rd, ang = useSpecular ? theta_s : phi_d;
        const float x = R * dm::cosf_(ang), y = R * dm::sinf_(ang);
        const V3 axis1 = cross(rc, axis);
        const V3 axis2 = cross(axis1, axis);
        return normalized(axis + normalized(axis1) * x + normalized(axis2) * y);
    }
    V3 dd = light.type == BHRT_LIGHT_DIRECT ? ld3(light.vec) : v3(0, 0, 0);
    return -(normalized(dd));
}

} // namespace bhrt
