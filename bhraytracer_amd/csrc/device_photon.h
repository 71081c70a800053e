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
#define BHRT_HEAP_COLUMN 1002   /* entries of one query's candidate column: slot 0 unused, 1..1000, one pad (16-byte pairs) */

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
// kGlobal = false: the caustic map (TraceCausticPhotonRay + RandomPhotonBounceForCaustic, Main.cpp:319-340, MtlBlinn.cpp:203-303);
// kGlobal = true: the global map (TracePhotonRay + RandomPhotonBounce, Main.cpp:296-317, MtlBlinn.cpp:140-202): a transmissive
// surface ends the photon, a diffuse bounce continues it with the power rescaled by kd / p_Diff.
template <bool kGlobal>
__device__ inline uint32_t emit_photon_path(const DevScene &S, uint32_t seed, uint64_t emission, const int32_t *plights, int n_plights,
                                            float sum_intensity, DPhoton *out, uint32_t cap)
{
    DRng g;
    bhrt_photon_stream(seed, emission, &g.key, &g.ctr);
    g.wrap = BHRT_PHOTON_WINDOW_MASK;
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
            if (kGlobal) break; // RandomPhotonBounce returns false for a transmissive material (MtlBlinn.cpp:149-151)
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
            const V3 diffuseRayDir = normalized(sample_in_semi_sphere(g, vN, diffuseTheta));
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
            const bool useSpecular = r01 >= p_Diff;
            if (!kGlobal && !useSpecular) break; // diffuse bounce: a caustic photon stops
            if (kGlobal) {
                const V3 kdf = ld3(m.diffuse.color) / p_Diff, ksf = ld3(m.specular.color) / p_Spec;
                intensity = intensity * (useSpecular ? ksf : kdf);
                d = useSpecular ? specRayDir : diffuseRayDir;
            } else {
                intensity = intensity * (ld3(m.specular.color) / p_Spec);
                d = specRayDir;
            }
            o = a.p + a.N * BHRT_BIAS;
        }
        first = false;
    }
    return stored;
}

// ---- gather ------------------------------------------------------------------------------------------------------
// Device copy of the balanced map, split by use: the walk reads only `hot` (position + split axis, one 16-byte load
// per visited node); an accepted photon adds two more 16-byte loads from `cold` (direction, power), both decoded
// once by k_photon_expand with the reference's own arithmetic (GetDirection's integer square root, Color24::ToColor).
struct PhotonMapDev {
    const float4 *hot;  // [n+1]: pos.xyz, w = bits of (planeAndDirZ & 3)
    const float4 *cold; // [2*(n+1)]: {dir.xyz, power}, {power * color.rgb, 0}
    const float4 *dbox; // [2*n_dbox]: {lo.xyz}, {hi.xyz}: bounds of the DIRECTIONS of the photons LocatePhotons reaches below node i (itself included), for the tree's top levels
    int n_dbox;         // nodes [1, n_dbox) have one (BHRT_DBOX_LEVELS levels)
    int n, half;        // photons[1..n] in heap order; half = n/2 - 1 (cyPhotonMap.h:257)
    float lo[3], hi[3]; // bounds of the photon positions
};

// Stackless walk over the balanced kd-tree in the order of LocatePhotons (cyPhotonMap.h:421-498), as ONE uniform step
// per loop iteration: every lane loads one node and either moves on (to the near child, to the far child, or back up)
// or reports that the node's own photon is due.  All decisions are selects — a lane-per-query walk written with nested
// loops ran with 7 of 64 lanes active (each lane in a different loop); in this form only the "photon due" part is a
// divergent block.
struct PhotonWalk {
    int cur = 1, from = 0; // node to handle next; the child we came back from (0 = arriving from the parent)
    bool done = false;
};
// One step.  Returns true when the own photon of node `node` (hot record `rec`) is due.  d2max = CURRENT squared radius.
__device__ inline bool photon_walk_step(const PhotonMapDev &M, PhotonWalk &w, V3 pos, float d2max, int &node, float4 &rec)
{
    const int cur = w.cur;
    const float4 h = M.hot[cur];
    const int axis = (int)__float_as_uint(h.w);
    const float dist = (axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z)) - (axis == 0 ? h.x : (axis == 1 ? h.y : h.z));
    const int nearc = dist > 0 ? 2 * cur + 1 : 2 * cur;
    const bool arriving = w.from == 0;
    const bool down_near = arriving && cur < M.half;                            // LocatePhotons recurses only below `half` (Q11)
    const bool down_far = !arriving && w.from == nearc && dist * dist < d2max;  // back from the near side: far side if the plane is within the radius
    const bool due = !(down_near || down_far);                                  // leaf on arrival, or both sides done
    node = cur;
    rec = h;
    // transitions
    w.done = due && cur == 1;
    w.cur = down_near ? nearc : (down_far ? (nearc ^ 1) : (cur >> 1));
    w.from = due ? cur : 0;
    return due;
}
__device__ inline bool photon_outside_bounds(const PhotonMapDev &M, V3 pos, float radius)
{
    const float r = radius * 1.001f + 1e-6f; // a query farther than this from the photons' box along one axis cannot accept any photon
    return pos.x < M.lo[0] - r || pos.x > M.hi[0] + r || pos.y < M.lo[1] - r || pos.y > M.hi[1] + r || pos.z < M.lo[2] - r || pos.z > M.hi[2] + r;
}
__device__ inline void photon_finish(V3 sumI, V3 sumD, float d2max, V3 &irrad, V3 &direction) // cyPhotonMap.h:366-381
{
    const float area = (float)M_PI * d2max;
    if (area > 0) {
        const float one_over_area = 1.0f / area;
        sumI = sumI * one_over_area;
    }
    irrad = sumI;
    direction = sumD / length(sumD);
}

// EstimateIrradiance<1000> for a query that meets fewer than 1000 photons: the candidate list never becomes a heap, so
// the sums run in walk order and no list has to be kept.  Returns 0 = no photon, 1 = done, 2 = the 1000th photon was
// met: the caller redoes this query with photon_estimate_heap (the sums then run in heap-array order); 3 = more than
// `budget` photons visited without an answer: the caller hands the query to a whole wave (k_photon_gather_select), so
// that one lane's long walk (a dense cluster inside the radius, most of it rejected) does not hold up its launch.
// found_out (statistics instantiation only, knob "gather_stats": the extra live value costs the default kernel 7 % — measured): the photons accepted
// (inside the radius, on the right side) — what the estimate is made of; bhrt_stats.photon_found
template <bool kFound = false>
__device__ inline int photon_estimate_fast(const PhotonMapDev &M, V3 pos, V3 normal, float radius, int budget, V3 &irrad, V3 &direction, uint32_t &visited, int *found_out = nullptr)
{
    irrad = v3(0, 0, 0);
    direction = v3(0, 0, 0);
    if (kFound) *found_out = 0;
    if (M.n <= 0 || photon_outside_bounds(M, pos, radius)) return 0;
    const float d2max = radius * radius;
    int found = 0;
    V3 sumI = v3(0, 0, 0), sumD = v3(0, 0, 0);
    PhotonWalk w;
    int node;
    float4 h;
    while (!w.done) {
        if (!photon_walk_step(M, w, pos, d2max, node, h)) continue;
        visited++;
        if (--budget < 0) return 3;
        const float dist2 = length_sq(v3(h.x, h.y, h.z) - pos);
        if (dist2 < d2max) {
            const float4 c0 = M.cold[2 * (size_t)node];
            const V3 pd = v3(c0.x, c0.y, c0.z);
            if (!(dot(pd, normal) >= 0)) {
                if (++found == BHRT_PHOTON_K) return 2;
                const float4 c1 = M.cold[2 * (size_t)node + 1];
                sumI = sumI + 1.f * v3(c1.x, c1.y, c1.z);
                sumD = sumD + pd * (1.f * c0.w);
            }
        }
    }
    if (kFound) *found_out = found;
    if (found == 0) return 0;
    photon_finish(sumI, sumD, d2max, irrad, direction);
    return 1;
}

// ---- heavy queries (>= 1000 acceptable photons inside the radius): the same SET as LocatePhotons, found by one wave --------
// What the reference's heap replay (cyPhotonMap.h:439-497) ends up with.  Let A be the acceptable photons inside the radius (dist2 <
// r^2, direction against the normal), in the order the walk meets them.  The first 1000, F, fill the list; the list becomes a max-heap;
// the 1001st REPLACES THE ROOT UNCONDITIONALLY (np.dist2[0] is still r^2 at that moment), i.e. m = the farthest photon of F leaves whatever
// the newcomer's distance; from then on np.dist2[0] is the heap's maximum and every further photon replaces the maximum iff it is
// nearer — a plain streaming selection.  So the final list is the 1000 nearest of A - {m}, and its np.dist2[0] their largest distance.
// With T = the 1001 nearest of A:  m is in T  <=>  every photon of F is in T  <=>  no photon of A outside T comes, in walk order, before
// the second-to-last member of T.  Then the list is T - {m}; otherwise it is simply the 1000 nearest of A.
//
// One wave per query: the walk's SET of nodes does not depend on the visiting order, so the wave expands it 64 * BHRT_SEL_NPL nodes
// per round from an LDS stack, with the shrinking bound of a k-nearest search (the 1001st nearest so far); candidates (dist2, photon,
// path) are appended to a per-wave buffer and selected from registers — the reference's per-query candidate heap is 8 KB of dependent
// global-memory accesses per replacement, which is what bounds the lane-per-query replay.  Walk order is recovered from the path alone: `sides` holds, most significant bit first, one bit
// per level (1 = the far side was taken); with the node's depth that is its rank key (photon_walk_rank's, without the loads).
// The float sums then run in candidate-buffer order, not in the reference's heap-array order: same photons, same area, the
// irradiance equal to a few ulp (north_star's bar is 1e-4); bhrt_opts.photon_exact = 1 keeps the exact replay for every query.
// The same kernel takes the LONG walks of the lane pass (a dense cluster inside the radius that the normal test rejects: up to 4 * 10^5
// nodes for a point on the glass sphere above the focus): when the walk ends with fewer than 1000 candidates the list never became a heap,
// the reference summed in walk order, and so does this — the candidates are sorted by their walk keys (bitonic, in the LDS the stack no
// longer needs) and added up in that order: identical bits.
// Returns 0 = no photon, 1 = done, 4 = undecided (a bound-pruned walk that saw no photon of A - T early enough in walk order; a full
// stack spill; exact_only and 1000 candidates): the caller hands the query to the exact replay.
#ifndef BHRT_SEL_CAP
#define BHRT_SEL_CAP 2048   /* candidates kept per query: 1001 + room between two compactions (a multiple of 64) */
#endif
#ifndef BHRT_SEL_STACK
#define BHRT_SEL_STACK 768  /* 9 KB of LDS per wave: sixteen waves per CU */
#endif
#ifndef BHRT_SEL_SLACK
#define BHRT_SEL_SLACK 64   /* a shed keeps between 1001 and 1001 + slack candidates */
#endif
#ifndef BHRT_SEL_NPL
#define BHRT_SEL_NPL 1      /* nodes per lane and round (2: half the rounds, but twice the stack and registers: slower at equal occupancy) */
#endif
#ifndef BHRT_SEL_TRIGGER
#define BHRT_SEL_TRIGGER 1280 /* candidates at which the far end is shed and the bound tightened: early and often — a compaction costs about
                                 three rounds of the walk, a loose bound costs hundreds (everything inside the full radius is visited) */
#endif
#define BHRT_SEL_T (BHRT_PHOTON_K + 1)
#define BHRT_SEL_SLOTS (BHRT_SEL_CAP / 64)
// The walk's stack lives in LDS (8 KB per wave: occupancy is what hides the two dependent loads of a round; with the candidates in LDS as
// well — 40 KB — a CU held four waves and the pass ran at a sixth of the lane pass's rate per node).  The candidates live in a scratch
// of the wave in global memory: appended with coalesced stores, read back into registers — lane l holds entries l, l + 64, ... — when the
// buffer is full, where the selection runs on registers only.
// LocatePhotons drops a photon whose direction has a non-negative dot product with the query's normal (cyPhotonMap.h:443-446).  With the
// bounds of the directions below a node, min over the box of d . n = sum_i min(lo_i n_i, hi_i n_i) is a lower bound of every such product;
// when it is clearly positive (2^-20 |n|_1: the products are sums of three terms of magnitude <= |n_i| with a rounding error below
// 2^-22 |n|_1 each way) every photon below the node is dropped by that test and the node's subtree holds nothing for this query —
// whatever its distance.  A point on the glass sphere right above the caustic has 4 * 10^5 photons in reach, all of them travelling
// the wrong way: the long walks of the gather were walks through such subtrees.
#define BHRT_DBOX_LEVELS 16
__device__ inline bool photon_subtree_rejected(const PhotonMapDev &M, uint32_t node, V3 n)
{
    if ((int)node >= M.n_dbox) return false;
    const float4 lo = M.dbox[2 * (size_t)node], hi = M.dbox[2 * (size_t)node + 1];
    const float bound = fminf(lo.x * n.x, hi.x * n.x) + fminf(lo.y * n.y, hi.y * n.y) + fminf(lo.z * n.z, hi.z * n.z);
    return bound >= 9.5367431640625e-07f * (fabsf(n.x) + fabsf(n.y) + fabsf(n.z)); // 2^-20; NaN compares false
}
struct SelectLds {
    uint32_t st_node[BHRT_SEL_STACK], st_sides[BHRT_SEL_STACK];
    float st_plane[BHRT_SEL_STACK]; // squared distance to the splitting plane that justified a far-side entry (0 for near-side entries)
};
#define BHRT_SEL_SPILL 4096 /* stack entries a wave can park in global memory when its LDS stack is full */
#define BHRT_SEL_SPILL_CHUNK 256
#define BHRT_SEL_SCRATCH_WORDS (3 * BHRT_SEL_CAP + 3 * BHRT_SEL_SPILL)
struct SelectScratch { // BHRT_SEL_CAP entries each; sp_*: BHRT_SEL_SPILL entries each
    uint32_t *d2, *idx, *sides;
    uint32_t *sp_node, *sp_sides;
    float *sp_plane;
};
// The LDS stack is full: its OLDEST BHRT_SEL_SPILL_CHUNK entries (far sides of the first levels: the least urgent) go to the wave's
// spill area in global memory, the rest moves down.  false = the spill area is full as well.
__device__ inline bool sel_spill(SelectLds &L, const SelectScratch &C, uint32_t &top, uint32_t &spilled, uint32_t lane)
{
    if (spilled + BHRT_SEL_SPILL_CHUNK > BHRT_SEL_SPILL || top < BHRT_SEL_SPILL_CHUNK) return false;
    for (uint32_t k = lane; k < BHRT_SEL_SPILL_CHUNK; k += 64) {
        C.sp_node[spilled + k] = L.st_node[k]; C.sp_sides[spilled + k] = L.st_sides[k]; C.sp_plane[spilled + k] = L.st_plane[k];
    }
    __syncthreads();
    for (uint32_t b = 0; b + BHRT_SEL_SPILL_CHUNK < top; b += 64) { // chunk after chunk: each is read whole before it is written lower down
        const uint32_t k = b + lane;
        const bool in = k + BHRT_SEL_SPILL_CHUNK < top;
        const uint32_t a = in ? L.st_node[k + BHRT_SEL_SPILL_CHUNK] : 0u, c = in ? L.st_sides[k + BHRT_SEL_SPILL_CHUNK] : 0u;
        const float p = in ? L.st_plane[k + BHRT_SEL_SPILL_CHUNK] : 0.f;
        __syncthreads();
        if (in) { L.st_node[k] = a; L.st_sides[k] = c; L.st_plane[k] = p; }
        __syncthreads();
    }
    top -= BHRT_SEL_SPILL_CHUNK;
    spilled += BHRT_SEL_SPILL_CHUNK;
    return true;
}
__device__ inline unsigned long long sel_key(uint32_t sides, uint32_t node)
{
    unsigned long long x = sides;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    const int depth = 31 - __clz((int)node);
    return x | (2ull << (62 - 2 * depth));
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }
__device__ inline unsigned long long wave_min_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
// the wave's candidate distances in registers: r[k] = entry lane + 64 k, or all ones beyond n (bit patterns of non-negative floats order like the floats)
struct SelRegs { uint32_t r[BHRT_SEL_SLOTS]; };
__device__ inline void sel_load(SelRegs &R, const SelectScratch &C, uint32_t n, uint32_t lane)
{
#pragma unroll
    for (int k = 0; k < BHRT_SEL_SLOTS; k++) { const uint32_t j = lane + 64u * k; R.r[k] = j < n ? C.d2[j] : 0xffffffffu; }
}
__device__ inline uint32_t sel_count_below(const SelRegs &R, uint32_t pivot)
{
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < BHRT_SEL_SLOTS; k++) c += R.r[k] < pivot ? 1u : 0u;
    return wave_sum_u32(c);
}
// keeps the entries with d2 < pivot, and up to `ties` of those == pivot, in buffer order; folds the walk keys of the others into min_out.
// In place: chunk k is read (a whole-wave load the stores of the chunk depend on) before anything at or below it is overwritten.
__device__ inline uint32_t sel_compact(const SelRegs &R, const SelectScratch &C, uint32_t n, uint32_t pivot, uint32_t ties, unsigned long long &min_out, uint32_t lane)
{
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t w = 0;
    unsigned long long mk = ~0ull;
#pragma unroll
    for (int k = 0; k < BHRT_SEL_SLOTS; k++) {
        if (64u * k >= n) break; // uniform
        const uint32_t j = lane + 64u * k;
        const bool in = j < n;
        const uint32_t d = R.r[k], ix = in ? C.idx[j] : 0u, sd = in ? C.sides[j] : 0u;
        const uint64_t mt = __ballot(in && d == pivot);
        const bool keep = in && (d < pivot || (d == pivot && (uint32_t)__popcll(mt & lt) < ties));
        const uint64_t mkp = __ballot(keep);
        ties -= (uint32_t)__popcll(__ballot(keep && d == pivot));
        if (in && !keep) { const unsigned long long key = sel_key(sd, ix); mk = key < mk ? key : mk; }
        if (keep) { const uint32_t o = w + (uint32_t)__popcll(mkp & lt); C.d2[o] = d; C.idx[o] = ix; C.sides[o] = sd; }
        w += (uint32_t)__popcll(mkp);
    }
    mk = wave_min_u64(mk);
    min_out = mk < min_out ? mk : min_out;
    return w;
}
__device__ inline int photon_estimate_select(const PhotonMapDev &M, SelectLds &L, const SelectScratch &C, V3 pos, V3 normal, float radius, V3 &irrad, V3 &direction,
                                             uint32_t &visited, uint32_t *knn_out /* optional: BHRT_PHOTON_K + 2 words: count, bits of np.dist2[0], the photons */,
                                             uint32_t *dbg /* rounds, compactions (per wave, lane-uniform) */, bool exact_only /* bhrt_opts.photon_exact */)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t lt = (1ull << lane) - 1ull;
    const float r2 = radius * radius;
    irrad = v3(0, 0, 0);
    direction = v3(0, 0, 0);
    if (M.n <= 0) return 0;
    float bound = r2;
    uint32_t top = 1, n = 0, spilled = 0;
    bool pruned = false; // the bound has left r^2: the buffer no longer holds all of A
    unsigned long long min_out = ~0ull; // smallest walk key among the photons of A seen and dropped
    if (lane == 0) { L.st_node[0] = 1; L.st_sides[0] = 0; L.st_plane[0] = 0.f; }
    __syncthreads();
    while (top > 0 || spilled > 0) {
        if (top == 0) { // the LDS stack ran empty: the last parked chunk comes back
            spilled -= BHRT_SEL_SPILL_CHUNK;
            for (uint32_t k = lane; k < BHRT_SEL_SPILL_CHUNK; k += 64) {
                L.st_node[k] = C.sp_node[spilled + k]; L.st_sides[k] = C.sp_sides[spilled + k]; L.st_plane[k] = C.sp_plane[spilled + k];
            }
            top = BHRT_SEL_SPILL_CHUNK;
            __syncthreads();
        }
        const uint32_t take = top < 64u * BHRT_SEL_NPL ? top : 64u * BHRT_SEL_NPL;
        const uint32_t base = top - take;
        top = base;
        dbg[0]++;
        uint32_t node[BHRT_SEL_NPL], sides[BHRT_SEL_NPL];
        bool valid[BHRT_SEL_NPL];
        float4 h[BHRT_SEL_NPL], c0[BHRT_SEL_NPL];
#pragma unroll
        for (int u = 0; u < BHRT_SEL_NPL; u++) {
            const uint32_t e = base + lane + 64u * u;
            valid[u] = e < base + take && L.st_plane[e] < bound; // a far side entered under a looser bound may be out of reach by now
            node[u] = valid[u] ? L.st_node[e] : 1u;
            sides[u] = valid[u] ? L.st_sides[e] : 0u;
            if (valid[u] && photon_subtree_rejected(M, node[u], normal)) valid[u] = false; // nothing below it passes the normal test
        }
        __syncthreads(); // every lane has read its stack slots
#pragma unroll
        for (int u = 0; u < BHRT_SEL_NPL; u++) { h[u] = M.hot[node[u]]; c0[u] = M.cold[2 * (size_t)node[u]]; } // all loads of the round in flight together
#pragma unroll
        for (int u = 0; u < BHRT_SEL_NPL; u++) {
            const uint32_t me = node[u];
            bool push_near = false, push_far = false;
            uint32_t nearc = 0;
            float plane2 = 0.f;
            if (valid[u] && (int)me < M.half) {
                const int axis = (int)__float_as_uint(h[u].w);
                const float dist = (axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z)) - (axis == 0 ? h[u].x : (axis == 1 ? h[u].y : h[u].z));
                nearc = dist > 0 ? 2 * me + 1 : 2 * me;
                push_near = true;
                plane2 = dist * dist;
                push_far = plane2 < bound;
            }
            visited += (uint32_t)__popcll(__ballot(valid[u])) * (lane == 0 ? 1u : 0u);
            const float dist2 = length_sq(v3(h[u].x, h[u].y, h[u].z) - pos);
            const bool accept = valid[u] && dist2 < bound && !(dot(v3(c0[u].x, c0[u].y, c0[u].z), normal) >= 0);
            const uint64_t mn = __ballot(push_near), mf = __ballot(push_far), ma = __ballot(accept);
            const uint32_t cn = (uint32_t)__popcll(mn), cf = (uint32_t)__popcll(mf);
            if (top + cn + cf > BHRT_SEL_STACK && !sel_spill(L, C, top, spilled, lane)) return 4;
            const uint32_t far_bit = 1u << __clz((int)me); // bit 31 - depth(me): the level decided at `me`, most significant bit first
            // far sides below near sides: the near sides are popped first (the bound tightens before the far sides are looked at)
            if (push_far) { const uint32_t o = top + (uint32_t)__popcll(mf & lt); L.st_node[o] = nearc ^ 1u; L.st_sides[o] = sides[u] | far_bit; L.st_plane[o] = plane2; }
            if (push_near) { const uint32_t o = top + cf + (uint32_t)__popcll(mn & lt); L.st_node[o] = nearc; L.st_sides[o] = sides[u]; L.st_plane[o] = 0.f; }
            top += cn + cf;
            if (accept) { const uint32_t o = n + (uint32_t)__popcll(ma & lt); C.d2[o] = __float_as_uint(dist2); C.idx[o] = me; C.sides[o] = sides[u]; }
            n += (uint32_t)__popcll(ma);
        }
        __syncthreads();
        if (exact_only && n >= BHRT_PHOTON_K) return 4;
        if (n >= BHRT_SEL_TRIGGER) {
            static_assert(BHRT_SEL_TRIGGER + 64 * BHRT_SEL_NPL <= BHRT_SEL_CAP && BHRT_SEL_T + BHRT_SEL_SLACK < BHRT_SEL_TRIGGER, "selection buffer sizes");
            // shed the far end: a pivot with 1001 <= #(d2 < pivot) <= 1001 + slack, by bisection on the bit patterns; the pivot is the new bound
            SelRegs R;
            sel_load(R, C, n, lane);
            uint32_t lo = 0, hi = __float_as_uint(bound), pivot = hi; // #(d2 < hi) = n >= 1001 throughout
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                const uint32_t c = sel_count_below(R, mid);
                if (c < BHRT_SEL_T) lo = mid;
                else { hi = mid; pivot = mid; if (c <= BHRT_SEL_T + BHRT_SEL_SLACK) break; }
            }
            n = sel_compact(R, C, n, pivot, 0u, min_out, lane);
            dbg[1]++;
            if (n >= BHRT_SEL_TRIGGER) return 4; // a few hundred candidates at exactly the same distance
            bound = __uint_as_float(pivot);
            pruned = true;
        }
    }
    if (n == 0) return 0;
    if (n < BHRT_PHOTON_K) {
        // the list never became a heap: sums in walk order (cyPhotonMap.h:353-365 over np.photon[1..found] as LocatePhotons filled it)
        static_assert(sizeof(SelectLds) >= 1024 * sizeof(unsigned long long), "the sort reuses the stack's LDS");
        if (M.n >= (1 << 26)) return 4; // the candidate's position rides in the key's low 10 bits: free below 2^26 photons (depth <= 25)
        unsigned long long *keys = reinterpret_cast<unsigned long long *>(&L);
        uint32_t n_sort = 64;
        while (n_sort < n) n_sort <<= 1;
        __syncthreads();
        for (uint32_t i = lane; i < n_sort; i += 64) keys[i] = i < n ? (sel_key(C.sides[i], C.idx[i]) | i) : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= n_sort; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = lane; t < n_sort / 2; t += 64) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), q = i | j; // pair (i, i + j), ascending where (i & k) == 0
                    const unsigned long long a = keys[i], b = keys[q];
                    if ((a > b) == ((i & k) == 0)) { keys[i] = b; keys[q] = a; }
                }
                __syncthreads();
            }
        V3 sumI = v3(0, 0, 0), sumD = v3(0, 0, 0);
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t m = n - base < 64u ? n - base : 64u;
            float4 c0 = make_float4(0, 0, 0, 0), c1 = c0;
            if (lane < m) { const size_t k = C.idx[(uint32_t)keys[base + lane] & 1023u]; c0 = M.cold[2 * k]; c1 = M.cold[2 * k + 1]; }
            for (uint32_t i = 0; i < m; i++) { // every lane adds the same values in the same order
                sumI = sumI + 1.f * v3(__shfl(c1.x, (int)i), __shfl(c1.y, (int)i), __shfl(c1.z, (int)i));
                sumD = sumD + v3(__shfl(c0.x, (int)i), __shfl(c0.y, (int)i), __shfl(c0.z, (int)i)) * (1.f * __shfl(c0.w, (int)i));
            }
        }
        __syncthreads();
        photon_finish(sumI, sumD, r2, irrad, direction);
        return 1;
    }
    float d2max = r2; // np.dist2[0]: stays r^2 until the 1001st photon arrives
    uint32_t drop = 0xffffffffu;
    if (n >= BHRT_SEL_T) {
        // T = the 1001 nearest: the 1001st smallest value is the largest lo with #(d2 < lo) < 1001
        SelRegs R;
        sel_load(R, C, n, lane);
        uint32_t lo = 0, hi = __float_as_uint(bound); // #(d2 < hi) = n >= 1001; #(d2 < lo) < 1001
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (sel_count_below(R, mid) < BHRT_SEL_T) lo = mid; else hi = mid;
        }
        const uint32_t below = sel_count_below(R, lo);
        n = sel_compact(R, C, n, lo, BHRT_SEL_T - below, min_out, lane); // now exactly 1001
        // the two largest walk keys of T, the farthest member, and the farthest member apart from the last one in walk order
        unsigned long long k1 = 0, far_all = 0;
        for (uint32_t j = lane; j < n; j += 64) {
            const unsigned long long k = sel_key(C.sides[j], C.idx[j]);
            k1 = k > k1 ? k : k1;
            const unsigned long long f = ((unsigned long long)C.d2[j] << 32) | j;
            far_all = f > far_all ? f : far_all;
        }
        k1 = wave_max_u64(k1);
        far_all = wave_max_u64(far_all);
        unsigned long long k2 = 0, far_f = 0;
        for (uint32_t j = lane; j < n; j += 64) {
            const unsigned long long k = sel_key(C.sides[j], C.idx[j]);
            if (k != k1) {
                k2 = k > k2 ? k : k2;
                const unsigned long long f = ((unsigned long long)C.d2[j] << 32) | j;
                far_f = f > far_f ? f : far_f;
            }
        }
        k2 = wave_max_u64(k2);
        far_f = wave_max_u64(far_f);
        if (min_out < k2) drop = (uint32_t)far_all;  // a photon outside T is among the first 1000 of the walk: the 1000 nearest
        else if (!pruned) drop = (uint32_t)far_f;    // all of A was seen: F = T without its last member in walk order, m = F's farthest
        else return 4;
        float dm = 0;
        for (uint32_t j = lane; j < n; j += 64) if (j != drop) dm = fmaxf(dm, __uint_as_float(C.d2[j]));
        for (int off = 32; off > 0; off >>= 1) dm = fmaxf(dm, __shfl_xor(dm, off));
        d2max = dm;
    }
    // sums over the list, candidate-buffer order: lane-strided partial sums, then a fixed butterfly
    V3 sumI = v3(0, 0, 0), sumD = v3(0, 0, 0);
    for (uint32_t j = lane; j < n; j += 64) {
        if (j == drop) continue;
        const size_t k = C.idx[j];
        const float4 c0 = M.cold[2 * k], c1 = M.cold[2 * k + 1];
        sumI = sumI + 1.f * v3(c1.x, c1.y, c1.z);
        sumD = sumD + v3(c0.x, c0.y, c0.z) * (1.f * c0.w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sumI = sumI + v3(__shfl_xor(sumI.x, off), __shfl_xor(sumI.y, off), __shfl_xor(sumI.z, off));
        sumD = sumD + v3(__shfl_xor(sumD.x, off), __shfl_xor(sumD.y, off), __shfl_xor(sumD.z, off));
    }
    if (knn_out) {
        if (lane == 0) { knn_out[0] = n - (drop != 0xffffffffu ? 1u : 0u); knn_out[1] = __float_as_uint(d2max); }
        for (uint32_t j = lane; j < n; j += 64)
            if (j != drop) knn_out[2 + (j > drop && drop != 0xffffffffu ? j - 1 : j)] = C.idx[j];
    }
    photon_finish(sumI, sumD, d2max, irrad, direction);
    return 1;
}

// The full replay with the 1000-entry candidate heap (cyPhotonMap.h:439-497,353-365).  cand: this lane's scratch column,
// element k at [k * stride]; one 8-byte entry = (bits of dist2) << 32 | photon index, so a sift step moves one word
// (dist2 >= 0: the float order is the order of the bit patterns, but the comparisons below stay float comparisons).
__device__ inline float cand_d2(unsigned long long e) { return __uint_as_float((uint32_t)(e >> 32)); }
__device__ inline unsigned long long make_cand(float d2, uint32_t idx) { return ((unsigned long long)__float_as_uint(d2) << 32) | idx; }
// child pair (2j, 2j+1) of the candidate heap in one 16-byte load (the column is 16-byte aligned, 2j is even)
__device__ inline void cand_pair(const unsigned long long *cand, int j, unsigned long long &c0, unsigned long long &c1)
{
    const ulonglong2 v = *(const ulonglong2 *)(cand + j);
    c0 = v.x; c1 = v.y;
}
__device__ inline bool photon_estimate_heap(const PhotonMapDev &M, V3 pos, V3 normal, float radius, unsigned long long *cand, size_t /*stride = 1*/,
                                            V3 &irrad, V3 &direction, uint32_t &visited)
{
    // The walk advances by uniform steps (photon_walk_step); the sift-downs stay inner loops.  (Tried and slower: every sift
    // LEVEL as a step of one flat loop, +27 %; a two-phase loop "all lanes walk to their next candidate, then sift
    // together", +9 %.)
    irrad = v3(0, 0, 0);
    direction = v3(0, 0, 0);
    if (M.n <= 0) return false;
    float d2max = radius * radius; // np.dist2[0]
    int found = 0;
    constexpr int kHalf = BHRT_PHOTON_K >> 1;
    PhotonWalk w;
    while (!w.done) {
        int node;
        float4 h;
        if (!photon_walk_step(M, w, pos, d2max, node, h)) continue;
        visited++;
        const float dist2 = length_sq(v3(h.x, h.y, h.z) - pos);
        if (!(dist2 < d2max)) continue;
        const float4 c0 = M.cold[2 * (size_t)node];
        if (dot(v3(c0.x, c0.y, c0.z), normal) >= 0) continue;
        if (found < BHRT_PHOTON_K) {
            found++;
            cand[found] = make_cand(dist2, (uint32_t)node);
            if (found == BHRT_PHOTON_K) { // build the max-heap (cyPhotonMap.h:458-476)
                for (int k = kHalf; k >= 1; k--) {
                    int parent = k;
                    const unsigned long long t = cand[k];
                    const float td2 = cand_d2(t);
                    while (parent <= kHalf) {
                        int j = parent + parent;
                        unsigned long long cj, cj1;
                        cand_pair(cand, j, cj, cj1);
                        if (j < BHRT_PHOTON_K && cand_d2(cj) < cand_d2(cj1)) { j++; cj = cj1; }
                        if (td2 >= cand_d2(cj)) break;
                        cand[parent] = cj;
                        parent = j;
                    }
                    cand[parent] = t;
                }
            }
        } else { // replace the maximum (cyPhotonMap.h:478-495)
            int parent = 1;
            while (parent <= kHalf) {
                int j = parent + parent;
                unsigned long long cj, cj1;
                cand_pair(cand, j, cj, cj1);
                if (j < BHRT_PHOTON_K && cand_d2(cj) < cand_d2(cj1)) { j++; cj = cj1; }
                if (dist2 > cand_d2(cj)) break;
                cand[parent] = cj;
                parent = j;
            }
            cand[parent] = make_cand(dist2, (uint32_t)node);
            d2max = cand_d2(cand[1]);
        }
    }
    if (found == 0) return false;
    V3 sumI = v3(0, 0, 0), sumD = v3(0, 0, 0); // list order: insertion order below 1000 photons, heap-array order from there on
    for (int i = 1; i <= found; i++) {
        const size_t k = (uint32_t)cand[i];
        const float4 c0 = M.cold[2 * k], c1 = M.cold[2 * k + 1];
        sumI = sumI + 1.f * v3(c1.x, c1.y, c1.z);
        sumD = sumD + v3(c0.x, c0.y, c0.z) * (1.f * c0.w);
    }
    photon_finish(sumI, sumD, d2max, irrad, direction);
    return true;
}
#endif // __HIPCC__

} // namespace bhrt
