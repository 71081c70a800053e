// device_photon.h — caustic photon map on the device.
//
// Follows (paths relative to /root/reference/BHRayTracer):
//   BuildCausticPhotonMap / TraceCausticPhotonRay        Main.cpp:319-386
//   PointLight::RandomPhoton, GetProbability             Lights/PointLight.cpp:20-34, Lights/lights.h:81
//   MtlBlinn::RandomPhotonBounceForCaustic               Materials/Blinn/MtlBlinn.cpp:203-303
//   PhotonMap::Photon (24-byte record), Set/GetDirection, SetPower   DataStructure/cyPhotonMap.h:72-90,172-214
//   PhotonMap::EstimateIrradiance<1000> / LocatePhotons  cyPhotonMap.h:332-382,421-498
//
// Emission: one lane per emission index (its own RNG stream, include/bhrt_rng.h), the lane follows its photon to
// the end and keeps the photons it would AddPhoton() in a private slot list; a stable compaction in emission order
// then reproduces "the first N AddPhoton calls" of the reference's sequential loop, independent of batch size.
// Gather: the reference's estimate depends on the ORDER in which the kd-tree walk meets the photons (the first
// heap replacement discards the current maximum unconditionally, and the float sums run in heap-array order), so
// each lane replays the exact sequential walk — stackless, because the balanced tree is in heap order
// (children 2i, 2i+1) — with its candidate list in a global scratch column.
#pragma once
#include "device_shade.h"

namespace bhrt {

#pragma pack(push, 1)
struct DPhoton { // cyPhotonMap.h:72-90 — also the record of Resource/causticPhotonMap.dat (Main.cpp:383-385)
    float pos[3];
    float power;
    uint8_t color[3];
    uint8_t planeAndDirZ;
    int16_t dirX, dirY;
};
#pragma pack(pop)
static_assert(sizeof(DPhoton) == 24, "photon record must be 24 bytes");

#define BHRT_PHOTON_ABSORB 0.3f /* Photon_AbsorbChance, MtlBlinn.cpp:27 */
#define BHRT_PHOTON_K 1000      /* MAX_PhotonCountInArea, MtlBlinn.cpp:28 */

BHRT_FN uint8_t float_to_byte(float r) // Color24::FloatToByte, cyColor.h:271-272
{
    int v = int(r * 255 + 0.5f);
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
BHRT_FN DPhoton make_photon(V3 pos, V3 dir, V3 power) // AddPhoton + SetDirection + SetPower, cyPhotonMap.h:172-190,218-232
{
    DPhoton p;
    p.pos[0] = pos.x; p.pos[1] = pos.y; p.pos[2] = pos.z;
    p.planeAndDirZ = 0; // uninitialised low bits in the reference
    p.dirX = (int16_t)(int)(dir.x * 0x7FFF);
    p.dirY = (int16_t)(int)(dir.y * 0x7FFF);
    if (!(dir.z > 0)) p.planeAndDirZ = 0x8;
    p.power = power.x;
    if (p.power < power.y) p.power = power.y;
    if (p.power < power.z) p.power = power.z;
    V3 q = power / p.power;
    p.color[0] = float_to_byte(q.x); p.color[1] = float_to_byte(q.y); p.color[2] = float_to_byte(q.z);
    return p;
}
BHRT_FN V3 photon_direction(const DPhoton &p) // GetDirection incl. the dropped dirY^2, cyPhotonMap.h:192-214
{
    V3 dir;
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
BHRT_FN V3 photon_power(const DPhoton &p) { return v3(p.color[0] / 255.0f, p.color[1] / 255.0f, p.color[2] / 255.0f) * p.power; }

#if defined(__HIPCC__)
// HandleRayWhenRefractionRayOut (MtlBlinn.cpp:543-589): true = leaves the medium (ro, rd set), false = internal reflection
__device__ inline bool refraction_out(DRng &g, V3 inDir, V3 hitP, V3 hitN, float ior, float refrGloss, V3 &ro, V3 &rd)
{
    V3 vN = hitN;
    V3 vV = -inDir;
    float cosPhi1 = dot(vV, -vN);
    float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
    float sinPhi2 = ior * sinPhi1;
    if (sinPhi2 <= 1) {
        float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
        V3 vTn = vN * cosPhi2;
        V3 vNxV = cross(vN, vV);
        V3 vTp = normalized(cross(vN, vNxV)) * sinPhi2;
        V3 vT = vTn + vTp;
        V3 vT_sampled = normalized(vT);
        if (refrGloss > 0) {
            float dotSign = 0;
            int guard = 0;
            while (dotSign <= 0 && guard++ < BHRT_MAXLOOP) {
                float theta = 0;
                vT_sampled = sample_along_light_direction(g, vT, refrGloss, theta);
                dotSign = dot(vT_sampled, vN);
            }
        }
        rd = normalized(vT_sampled);
        ro = hitP + vN * BHRT_BIAS;
        return true;
    }
    rd = ((-2 * cosPhi1) * vN - vV);
    ro = hitP - vN * BHRT_BIAS;
    return false;
}

// One emitted photon followed to its end.  Writes at most `cap` records to out[0..), returns how many the
// reference would have passed to AddPhoton (may exceed cap -> caller flags overflow).
__device__ inline uint32_t emit_photon_path(const DevScene &S, uint32_t seed, uint64_t emission, const int32_t *plights, int n_plights,
                                            float sum_intensity, DPhoton *out, uint32_t cap)
{
    DRng g;
    g.key = bhrt_photon_key(seed, emission);
    g.ctr = 0;
    // light choice, Main.cpp:365-371
    float rnd = g.rnd01();
    int li = 0;
    while (rnd > gray3(S.lights[plights[li]].intensity) * S.lights[plights[li]].size / sum_intensity && li < n_plights - 1) li++;
    const bhrt_light &light = S.lights[plights[li]];
    // PointLight::RandomPhoton, PointLight.cpp:20-34
    float phi = (float)(g.rnd01() * 2 * BHRT_PI_D);
    float theta = acos_safe(1 - 2 * g.rnd01());
    V3 axisZ = v3(0, 0, 1), axisX = v3(1, 0, 0), axisY = v3(0, 1, 0);
    V3 d = dm::sinf_(theta) * (axisX * dm::cosf_(phi) + axisY * dm::sinf_(phi)) + axisZ * dm::cosf_(theta);
    V3 o = ld3(light.vec);
    V3 intensity = ld3(light.intensity);
    uint32_t stored = 0;
    bool first = true;
    for (int guard = 0; guard < BHRT_MAXLOOP; guard++) { // TraceCausticPhotonRay, Main.cpp:319-340
        Hit h;
        trace_closest(S, o, d, BHRT_HIT_FRONT, h);
        if (h.node < 0) break;
        const int mi = S.nodes[h.node].material;
        if (mi < 0) break;
        const bhrt_material &m = S.materials[mi];
        Attr a;
        hit_attrs(S, o, d, h.t, h.node, h.prim, false, a);
        if (!first && gray3(m.diffuse.color) > 0) { // IsPhotonSurface, materials.h:47
            if (stored < cap) out[stored] = make_photon(a.p, normalized(d), intensity);
            stored++;
        }
        if (m.kind != BHRT_MTL_BLINN) break;
        // RandomPhotonBounceForCaustic, MtlBlinn.cpp:203-303
        float r01 = g.rnd01();
        V3 vN = normalized(a.N);
        V3 vV = -(normalized(d));
        if (gray3(m.refraction.color) > 0) {
            float cosPhi1 = dot(vN, vV);
            float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
            float sinPhi2 = sinPhi1 / m.ior;
            float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
            V3 vTn = (-cosPhi2) * vN;
            V3 vNxV = cross(vN, vV);
            V3 vTp = normalized(cross(vN, vNxV)) * sinPhi2;
            V3 vT = vTn + vTp;
            V3 io = a.p - vN * BHRT_BIAS;
            Hit hb;
            trace_closest(S, io, vT, BHRT_HIT_BACK, hb);
            if (hb.node < 0) break;
            Attr ab;
            hit_attrs(S, io, vT, hb.t, hb.node, hb.prim, false, ab);
            V3 no, ndir;
            if (!refraction_out(g, vT, ab.p, ab.N, m.ior, m.refraction_glossiness, no, ndir)) break;
            o = no;
            d = ndir;
        } else {
            if (r01 < BHRT_PHOTON_ABSORB) break;
            float diffuseTheta = 0;
            (void)normalized(sample_in_semi_sphere(g, vN, diffuseTheta));
            float p_diffuseTheta = dm::sinf_(2 * diffuseTheta);
            float specularTheta = 0;
            float cosvVvN = dot(vN, vV);
            V3 vR = (2 * cosvVvN) * vN - vV;
            V3 specRayDir = sample_along_light_direction(g, vR, m.glossiness, specularTheta);
            float p_specularTheta = dm::powf_(dm::cosf_(specularTheta), m.glossiness);
            float P_Diffuse = tc_max(m.diffuse.color) * p_diffuseTheta;
            float P_sum = P_Diffuse + tc_max(m.specular.color) * p_specularTheta;
            float p_Diff = (P_Diffuse / P_sum) * (1 - BHRT_PHOTON_ABSORB) + BHRT_PHOTON_ABSORB;
            float p_Spec = (1 - p_Diff) * (1 - BHRT_PHOTON_ABSORB) + BHRT_PHOTON_ABSORB;
            if (!(r01 >= p_Diff)) break; // diffuse bounce: a caustic photon stops
            intensity = intensity * (ld3(m.specular.color) / p_Spec);
            d = specRayDir;
            o = a.p + a.N * BHRT_BIAS;
        }
        first = false;
    }
    return stored;
}

// Exact replay of EstimateIrradiance<1000> + LocatePhotons for one query.  photons[1..n] balanced (heap order),
// half = n/2 - 1 (cyPhotonMap.h:257, Q11).  cd2 / cidx: this lane's scratch columns, element k at [k * stride].
__device__ inline bool photon_estimate(const DPhoton *photons, int n, int half, V3 pos, V3 normal, float radius, float *cd2, uint32_t *cidx,
                                       size_t stride, V3 &irrad, V3 &direction)
{
    irrad = v3(0, 0, 0);
    direction = v3(0, 0, 0);
    if (n <= 0) return false;
    float d2max = radius * radius; // np.dist2[0]
    int found = 0;
    bool heap = false;
    V3 sumI = v3(0, 0, 0), sumD = v3(0, 0, 0); // running sums, valid while the list is still in insertion order
    int idx = 1;
    bool desc = true;
    int from = 0;
    while (true) {
        int node_to_process = 0;
        if (desc) {
            if (idx < half) {
                const DPhoton &p = photons[idx];
                const int axis = p.planeAndDirZ & 0x3;
                const float dist = (axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z)) - p.pos[axis];
                idx = dist > 0 ? 2 * idx + 1 : 2 * idx;
                continue;
            }
            node_to_process = idx;
        } else {
            if (from == 1) break;
            const int par = from >> 1;
            const DPhoton &p = photons[par];
            const int axis = p.planeAndDirZ & 0x3;
            const float dist = (axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z)) - p.pos[axis];
            const int firstc = dist > 0 ? 2 * par + 1 : 2 * par;
            if (from == firstc && dist * dist < d2max) {
                idx = firstc ^ 1;
                desc = true;
                continue;
            }
            node_to_process = par;
        }
        { // the node's own photon, cyPhotonMap.h:439-497
            const DPhoton p = photons[node_to_process];
            V3 dif = ld3(p.pos) - pos;
            float dist2 = length_sq(dif);
            if (dist2 < d2max) {
                V3 pd = photon_direction(p);
                if (!(dot(pd, normal) >= 0)) {
                    if (found < BHRT_PHOTON_K) {
                        found++;
                        cd2[(size_t)found * stride] = dist2;
                        cidx[(size_t)found * stride] = (uint32_t)node_to_process;
                        sumI = sumI + 1.f * photon_power(p);
                        sumD = sumD + pd * (1.f * p.power);
                        if (found == BHRT_PHOTON_K) { // build the max-heap
                            heap = true;
                            const int half_found = found >> 1;
                            for (int k = half_found; k >= 1; k--) {
                                int parent = k;
                                const uint32_t tp = cidx[(size_t)k * stride];
                                const float td2 = cd2[(size_t)k * stride];
                                while (parent <= half_found) {
                                    int j = parent + parent;
                                    if (j < found && cd2[(size_t)j * stride] < cd2[(size_t)(j + 1) * stride]) j++;
                                    if (td2 >= cd2[(size_t)j * stride]) break;
                                    cd2[(size_t)parent * stride] = cd2[(size_t)j * stride];
                                    cidx[(size_t)parent * stride] = cidx[(size_t)j * stride];
                                    parent = j;
                                }
                                cidx[(size_t)parent * stride] = tp;
                                cd2[(size_t)parent * stride] = td2;
                            }
                        }
                    } else {
                        int parent = 1, j = 2;
                        while (j <= found) {
                            if (j < found && cd2[(size_t)j * stride] < cd2[(size_t)(j + 1) * stride]) j++;
                            if (dist2 > cd2[(size_t)j * stride]) break;
                            cd2[(size_t)parent * stride] = cd2[(size_t)j * stride];
                            cidx[(size_t)parent * stride] = cidx[(size_t)j * stride];
                            parent = j;
                            j <<= 1;
                        }
                        cidx[(size_t)parent * stride] = (uint32_t)node_to_process;
                        cd2[(size_t)parent * stride] = dist2;
                        d2max = cd2[(size_t)1 * stride];
                    }
                }
            }
        }
        from = node_to_process;
        desc = false;
    }
    if (found == 0) return false;
    if (heap) { // the list was reordered: sum in heap-array order like cyPhotonMap.h:353-365
        sumI = v3(0, 0, 0);
        sumD = v3(0, 0, 0);
        for (int i = 1; i <= found; i++) {
            const DPhoton p = photons[cidx[(size_t)i * stride]];
            sumI = sumI + 1.f * photon_power(p);
            sumD = sumD + photon_direction(p) * (1.f * p.power);
        }
    }
    const float area = (float)M_PI * d2max;
    if (area > 0) {
        const float one_over_area = 1.0f / area;
        sumI = sumI * one_over_area;
    }
    irrad = sumI;
    direction = sumD / length(sumD);
    return true;
}
#endif // __HIPCC__

} // namespace bhrt
