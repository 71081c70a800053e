// device_types.h — data the HIP kernels see: the scene view over the HBM-resident flat blob and
// the SoA wavefront buffers (ray queues, compact hits, shade frames).
#pragma once
#include <stdint.h>

#include "bhrt_flat.h"

namespace bhrt {

// View over the uploaded flat scene (include/bhrt_flat.h); passed to kernels by value.
struct DevScene {
    const uint8_t *blob;
    const bhrt_node *nodes;
    const bhrt_mesh *meshes;
    const bhrt_material *materials;
    const bhrt_light *lights;
    const bhrt_texmap *texmaps;
    const bhrt_texture *textures;
    const int32_t *chain; // n_nodes x BHRT_MAX_NODE_DEPTH: ancestors of node n from depth 1 down to n itself
    // per-material / per-light constants of Shade() that the reference recomputes on every call, formed once at upload with
    // the same float operations: Schlick R0 (MtlBlinn.cpp:107) and the light-choice thresholds Gray(I) / allLightIntensity
    // (MtlBlinn.cpp:311-318)
    const float *mat_r0;     // [n_materials]
    const float *light_pick; // [n_lights]
    // the camera position carried through every node's ancestor chain (Node::ToNodeCoords, scene.h:490-496: p' = itm*(p-pos)),
    // formed at upload with the same float operations: every camera ray shares its origin, so the kernels of the first wave
    // step only transform the direction.  [n_nodes][BHRT_MAX_NODE_DEPTH][3]: entry k = the position after chain level k
    const float *cam_chain;
    int32_t n_nodes, n_lights;
    float all_light_intensity;
    bhrt_camera cam;
    bhrt_texcolor background, environment;
    float tapx[32], tapy[32]; // the 31 elliptic Halton taps of Texture::Sample (scene.h:322-329)
};

// continuation kinds of a closest-hit ray
enum : uint32_t {
    RK_CAMERA = 0,   // Main.cpp:153-158      frame = sample slot
    RK_GI = 1,       // MtlBlinn.cpp:392-396  frame = shading frame that asked
    RK_REFR_IN = 2,  // MtlBlinn.cpp:478-483
    RK_REFR_OUT = 3, // MtlBlinn.cpp:524-526
    RK_DEAD = 15     // padding slot (pixel of an edge tile outside the image): never traced or shaded
};
// how a frame's result reaches its parent
enum : uint32_t { FH_ROOT = 0, FH_GI = 1, FH_REFR_FRONT = 2, FH_REFR_OUT = 3 };
// direct-light term state of a frame
enum : uint32_t { DM_NONE = 0, DM_AMBIENT = 1, DM_DIRECT = 2, DM_POINT = 3, DM_POINT_ZERO = 4 };
// frame flags
enum : uint32_t { FF_CONST = 1u /* value in refr */, FF_HAS_REFR = 2u /* a refraction ray was emitted: refr + refr_color valid */, FF_HAS_GI = 4u };

// Closest-hit ray queue (SoA, one array per field -> coalesced)
struct RayQueue {
    float *ox, *oy, *oz, *dx, *dy, *dz;
    uint32_t *frame;   // owner (sample slot for RK_CAMERA)
    uint32_t *meta;    // kind | side << 4 | bounce << 8
    uint32_t *rng_ctr; // draw counter of the owner's refraction-section stream (glossy refraction chains)
};
struct HitBuf {
    float *t;
    int32_t *node, *prim, *front;
};
struct ShadowQueue {
    float *ox, *oy, *oz, *dx, *dy, *dz, *tmax;
    uint32_t *frame;
};
// One MtlBlinn::Shade invocation (MtlBlinn.cpp:89-138) awaiting its sub-terms
struct Frames {
    uint32_t *parent;  // parent frame, or sample slot when how == FH_ROOT
    uint32_t *info;    // how | dmode << 3 | light << 8 | flags << 16 | material << 20 (12 bits)
    uint32_t *info2;   // (gi + 64) | bounce << 8
    uint32_t *skey;    // sample key (include/bhrt_rng.h)
    uint64_t *code;    // shade-call path code
    float *mult;       // 3: factor applied when delivering to the parent (GI: kd/ks sample; REFR_OUT: refraction*absorption)
    float *refr;       // 3: refraction term (MtlBlinn.cpp:117)
    float *gi;         // 3: GI term (:124)
    float *gi_mult;    // 3: (useSpecular ? specular : diffuse).Sample at this frame's hit (:406,417)
    float *brdf;       // 3: brdfXCosTheta (:325)
    float *refr_color; // 3: (1-F)*refraction (:117)
    float *rr;         // squared distance to the point light (PointLight.cpp:10-11)
    float *vis;        // Shadow() result (GenLight.cpp:10-13)
    float *caustic;    // 3: photon-map term brdf*irrad (:329-342), zero when off
    // inputs of the caustic term, only allocated when the photon map is on: hit p, hit N, vV, kd sample, ks sample
    float *ph_p, *ph_n, *ph_v, *ph_kd, *ph_ks;
};

#define BHRT_ORDER_SHARDS 32

struct Counters {
    // One counter per 128-byte line: atomics to different words of ONE line still serialise in L2
    // (measured: 196 k atomics on one line -> +1.7 ms).  The host reads back the first four lines after a wave step.
    struct alignas(128) Line { uint32_t v; uint32_t pad[31]; };
    Line n_next;    // rays pushed to the next closest-hit queue   (zeroed by k_order_prefix before each k_shade)
    Line n_shadow;  // rays pushed to the shadow queue             (same)
    Line n_frames;  // frames allocated so far in this pass
    Line overflow;  // set when a capacity was exceeded
    // traced rays sorted for shading: class x shard element counts (see RayOrder)
    Line cls[4][BHRT_ORDER_SHARDS];
    Line n_slow;    // rays set aside in the slow queue (kernels.hip::SlowQueue), this pass
    Line shade_done; // workgroups of the step's k_shade that have finished (the last one publishes the step's counters; it leaves 0 behind)
    Line mesh_cursor; // next entry of the parked list to hand out (k_trace_mesh_stream; zeroed by k_mesh_prefix)
};
#define BHRT_COUNTERS_HOST_BYTES (4 * 128)

// Shading order.  k_trace_closest files every traced ray under one of three classes, so that a k_shade workgroup
// only holds rays of one class (GI rays are incoherent: without this, nearly every wave mixes a few expensive
// Shade() entries with many trivial misses and pays for the longest path):
//   0 heavy   the hit opens a new Shade() frame (camera / GI / refraction-front / refraction-out hits)
//   1 medium  refraction ray hit a back face: HandleRayWhenRefractionRayOut, emits one ray
//   2 light   misses and the |z| <= Bias GI case: background / environment / constants
// One atomic per wave and class, on counters sharded BHRT_ORDER_SHARDS ways by wave index, each on its own
// 128-byte line (one line takes only ~88 atomics/us; a workgroup-level barrier here would hold the whole
// workgroup until its slowest ray is done).
//
// A fourth list, filled by the first trace kernel of a step when the scene has meshes: rays that reached a mesh whose
// root box they hit.  They are parked there (hit so far + the scene node to resume at, in the hit buffer) and finished
// by k_trace_mesh in dense waves — in the mixed launch a wave pays a full BVH traversal for every few mesh rays it holds.
//   3 mesh    parked at a mesh node (not a shading class: k_trace_mesh files the finished ray under 0..2)
enum : uint32_t { RC_HEAVY = 0, RC_MEDIUM = 1, RC_LIGHT = 2, RC_MESH = 3, RC_NONE = 4 };
struct RayOrder {
    uint32_t *idx;      // [4][BHRT_ORDER_SHARDS][shard_cap] ray indices
    uint32_t shard_cap;
    // written by k_order_prefix after the trace kernel: first k_shade workgroup of every (class, shard) segment
    // (+ one end marker) and the segment's element count, both compact so that a workgroup finds its segment
    // with two loads per lane instead of walking 96 counter lines
    uint32_t *seg_start; // [3 * BHRT_ORDER_SHARDS + 1]
    uint32_t *seg_count; // [3 * BHRT_ORDER_SHARDS]
    // the same for the parked mesh rays (k_mesh_prefix): first k_trace_mesh workgroup / element count per shard
    uint32_t *mesh_start; // [BHRT_ORDER_SHARDS + 1]
    uint32_t *mesh_count; // [BHRT_ORDER_SHARDS + 1]: per shard, then the total
    // coherence sort of the parked rays (counting sort by BHRT_PARK_KEY_BITS-bit key: entry cell in the mesh's box + direction octant)
    uint32_t *park_key;   // [cap_rays] key of ray i, written when it is parked
    uint32_t *park_sorted; // [cap_rays] the parked ray indices in key order
    uint32_t *park_bucket; // [1 << BHRT_PARK_KEY_BITS] + tile sums behind it
    uint32_t *frame_base;  // [3 * BHRT_ORDER_SHARDS] first Shade() frame of a segment's rays (heavy class; k_order_prefix)
    uint32_t *park_rank;   // [cap_rays] rank of ray i inside its key's bucket (what the histogram's atomic returned)
};
#define BHRT_PARK_CELL_BITS 6
#define BHRT_PARK_KEY_BITS (3 * BHRT_PARK_CELL_BITS + 3)

} // namespace bhrt
