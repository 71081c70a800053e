// kernels.hip — the MI355X (gfx950) render path: wavefront path tracing over SoA buffers in HBM.
//
// One launch of each kernel processes one "wave step" of all paths in flight:
//   camera_ray()    Main.cpp:132-153,179-192   the camera ray of a (pixel, sample) slot, computed inside the first step's kernels
//   k_trace_closest Main.cpp:389-413 + Objects/* ordered scene-graph closest hit -> compact hits; files every ray under its
//                                              shading class; with meshes in the scene it parks the rays that enter a mesh
//   k_trace_mesh    TriObj.cpp:17-39,192-270   the parked rays, key-sorted, in dense workgroups: BVH traversal + rest of the scene graph
//   k_shade         MtlBlinn.cpp:89-589        one lane per traced ray, in class-sorted order: rebuilds the HitInfo, evaluates one
//                                              Shade() entry or one refraction-chain step, emits <=2 closest rays,
//                                              <=1 shadow ray, <=1 new shading frame (wave ballot + prefix-sum
//                                              compaction, one atomic per workgroup and queue)
//   k_trace_shadow  GenLight.cpp:10-69         any-hit -> visibility into the owning frame (+ _park / k_shadow_mesh with meshes; there on a second
//                                              stream, started with the NEXT step's mesh walk: nothing reads a visibility before k_combine)
//   k_combine       MtlBlinn.cpp:117-137,343,431,470,511,539  folds finished frames into their parents, deepest
//                                              wave step first (the per-level clamps forbid a running throughput)
//   k_resolve       Main.cpp:170,220-230       in-order sample sum, /spp, gamma, Color24
//   k_photon_*      Main.cpp:319-386, cyPhotonMap.h   caustic photon map: emission, and the k-NN gather in three passes
//   k_tiles_*       (no counterpart)           multi-GPU framebuffer exchange: pack / unpack of a rank's tiles
// The recursion of the reference becomes: ray kinds (continuations) + a tree of shading frames.
// No CPU fallback: every entry point fails when there is no usable HIP device.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <chrono>
#include <string>
#include <vector>

#include "bhrt.h"
#include "device_photon.h"
#include "photon_host.h"
#include "scene_internal.h"

namespace bhrt {

#define HIP_CHECK(expr)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            SetError(std::string(#expr) + ": " + hipGetErrorString(e_));                                       \
            return BHRT_ERR_HIP;                                                                               \
        }                                                                                                      \
    } while (0)

constexpr int kBlock = 256;

// Exact unsigned 32-bit division by a divisor that is fixed for a launch (Granlund & Montgomery, "Division by invariant
// integers using multiplication"): the slot -> (pixel, sample) -> tile arithmetic is four div/mod pairs per lane, ~20
// instructions each as runtime divisions, 5 as this.  Holds for every 32-bit dividend.
struct FastDiv {
    uint32_t d, m, s1, s2;
};
static FastDiv MakeFastDiv(uint32_t d)
{
    FastDiv f;
    f.d = d ? d : 1;
    uint32_t l = 0;
    while ((1ull << l) < f.d) l++; // ceil(log2 d)
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - f.d)) / f.d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ inline uint32_t fdiv(uint32_t n, const FastDiv &f)
{
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}
__device__ inline void fdivmod(uint32_t n, const FastDiv &f, uint32_t &q, uint32_t &r)
{
    q = fdiv(n, f);
    r = n - q * f.d;
}

struct PassInfo {
    int32_t W, H, tile, tiles_x, tiles_y, rank, world;
    uint32_t q0;       // first owned-pixel index of this pass
    uint32_t n_pixels; // owned pixels in this pass (incl. out-of-image pixels of edge tiles)
    int32_t spp;
    uint32_t seed;
    int32_t jitter, gamma;
    // per-frame constants of RandomPositionInPixel (Main.cpp:132-139), formed once on the host with the same float
    // operations the reference performs per sample: unit dd_x, unit dd_y, |dd_x|
    float jx[3], jy[3], pixel_len;
    FastDiv by_spp, by_tile_px, by_tiles_x, by_tile; // divisions by spp, tile * tile, tiles_x, tile
};
struct RenderParams {
    int32_t internal_bounces, gi_bounces;
    uint32_t cap_rays, cap_shadow, cap_frames;
    int32_t photon;
};

// owned-pixel index q -> image coordinates; tiles are dealt round-robin to ranks
__device__ inline bool pixel_of(const PassInfo &P, uint32_t q, int &i, int &j)
{
    uint32_t k, within, ty, tx, wy, wx;
    fdivmod(q, P.by_tile_px, k, within);
    const uint32_t tile_id = (uint32_t)P.rank + k * (uint32_t)P.world;
    fdivmod(tile_id, P.by_tiles_x, ty, tx);
    fdivmod(within, P.by_tile, wy, wx);
    i = (int)(tx * P.tile + wx);
    j = (int)(ty * P.tile + wy);
    return ty < (uint32_t)P.tiles_y && i < P.W && j < P.H;
}

// Queue compaction: wave64 ballot + prefix popcount inside each wave, wave totals combined through LDS,
// ONE atomic per workgroup and queue.  (One atomic per wave was the bottleneck of the first version: a
// single word takes ~88 atomics/us on MI355X, and a 16 M-lane launch has 262 k waves.)
// All threads of the block must call it (convergent call sites only).  kShadeBlock threads per block.
// 512 threads: two workgroups fit a CU at k_shade's 126 VGPRs, so one's load / barrier / store phases overlap the
// other's arithmetic (1024 = the whole CU in lockstep: +18 % kernel time); 256 is no faster and doubles the queue atomics.
#ifndef BHRT_SHADE_BLOCK
#define BHRT_SHADE_BLOCK 512
#endif
constexpr int kShadeBlock = BHRT_SHADE_BLOCK;
constexpr int kShadeWaves = kShadeBlock / 64;

struct BlockAllocLds {
    uint32_t total[3][kShadeWaves];
    uint32_t base[3][kShadeWaves];
};

// per-lane request of `cnt` (0..2) slots from up to three counters at once; returns this lane's first slot
__device__ inline void block_alloc3(BlockAllocLds &L, uint32_t *c0, uint32_t n0a, uint32_t n0b, uint32_t *c1, uint32_t n1, uint32_t *c2, uint32_t n2,
                                    uint32_t &s0a, uint32_t &s0b, uint32_t &s1, uint32_t &s2)
{
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t ma = __ballot(n0a != 0), mb = __ballot(n0b != 0), m1 = __ballot(n1 != 0), m2 = __ballot(n2 != 0);
    const uint32_t pa = (uint32_t)__popcll(ma & lt), pb = (uint32_t)__popcll(mb & lt);
    const uint32_t ta = (uint32_t)__popcll(ma);
    if (lane == 0) {
        L.total[0][wave] = ta + (uint32_t)__popcll(mb);
        L.total[1][wave] = (uint32_t)__popcll(m1);
        L.total[2][wave] = (uint32_t)__popcll(m2);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t k = threadIdx.x;
        uint32_t sum = 0;
        for (int w = 0; w < kShadeWaves; w++) { const uint32_t t = L.total[k][w]; L.base[k][w] = sum; sum += t; }
        uint32_t *ctr = k == 0 ? c0 : (k == 1 ? c1 : c2);
        const uint32_t b = (sum && ctr) ? atomicAdd(ctr, sum) : 0u;
        for (int w = 0; w < kShadeWaves; w++) L.base[k][w] += b;
    }
    __syncthreads();
    // within a wave: all "a" requests first, then the "b" requests
    s0a = L.base[0][wave] + pa;
    s0b = L.base[0][wave] + ta + pb;
    s1 = L.base[1][wave] + (uint32_t)__popcll(m1 & lt);
    s2 = L.base[2][wave] + (uint32_t)__popcll(m2 & lt);
    __syncthreads(); // L may be reused by the next call
}

// per-sample radiance buffer is sample-major: [sample][pixel of the pass] -> k_resolve reads coalesced
__device__ inline uint32_t sample_addr(const PassInfo &P, uint32_t slot)
{
    uint32_t px, smp;
    fdivmod(slot, P.by_spp, px, smp);
    return smp * P.n_pixels + px;
}

__device__ inline void put_ray(const RayQueue &q, uint32_t i, V3 o, V3 d, uint32_t frame, uint32_t meta, uint32_t ctr)
{
    q.ox[i] = o.x; q.oy[i] = o.y; q.oz[i] = o.z; q.dx[i] = d.x; q.dy[i] = d.y; q.dz[i] = d.z;
    q.frame[i] = frame; q.meta[i] = meta;
    if ((meta & 15u) != RK_GI) q.rng_ctr[i] = ctr; // only a refraction chain carries its stream position along
}
__device__ inline uint32_t make_meta(uint32_t kind, uint32_t side, int bounce) { return kind | (side << 4) | ((uint32_t)(bounce & 0xff) << 8); }
__device__ inline void st3(float *a, uint32_t i, V3 v) { a[3 * (size_t)i] = v.x; a[3 * (size_t)i + 1] = v.y; a[3 * (size_t)i + 2] = v.z; }
__device__ inline V3 ld3i(const float *a, uint32_t i) { return v3(a[3 * (size_t)i], a[3 * (size_t)i + 1], a[3 * (size_t)i + 2]); }

// ------------------------------------------------------------------------------------------------
// The camera ray of slot idx = (pixel of the pass, sample): PathTracing + RandomPositionInPixel (Main.cpp:132-153).
// Camera rays are never stored: the first wave step's kernels (k_trace_closest / k_trace_mesh / k_shade with
// kCamera) each call this — ~300 VALU instructions instead of a 36-byte record written once and read twice.
// false: the slot is a pixel of an edge tile outside the image (a dead ray).
__device__ inline bool camera_ray(const DevScene &S, const PassInfo &P, uint32_t idx, V3 &o, V3 &d)
{
    int i = 0, j = 0;
    uint32_t q_local, s;
    fdivmod(idx, P.by_spp, q_local, s);
    o = v3(0, 0, 0); d = v3(0, 0, 0);
    if (!pixel_of(P, P.q0 + q_local, i, j)) return false;
    // PathTracing(), Main.cpp:145: pixel "centre" = corner because 1/2 == 0
    const V3 topLeft = ld3(S.cam.top_left), ddx = ld3(S.cam.dd_x), ddy = ld3(S.cam.dd_y), pos = ld3(S.cam.pos);
    V3 target = (topLeft + (float)i * ddx) - (float)j * ddy;
    if (P.jitter) { // RandomPositionInPixel, Main.cpp:132-139: two raw rand() draws in double
        const float pixelLen = P.pixel_len;
        const V3 ux = ld3(P.jx), uy = ld3(P.jy);
        const uint32_t key = bhrt_sample_key(P.seed, (uint32_t)(j * P.W + i), s);
        float fx = (float)(((double)bhrt_rand31(key, 0) / (BHRT_RAND_MAX)) * 2 - 1);
        target = target + ((ux * fx) * pixelLen) / 2.f;
        float fy = (float)(((double)bhrt_rand31(key, 1) / (BHRT_RAND_MAX)) * 2 - 1);
        target = target + ((uy * fy) * pixelLen) / 2.f;
    }
    o = pos;
    d = target - pos;
    return true;
}
// one queued ray, or (kCamera) the camera ray of slot i: one slot per (pixel, sample), no compaction
template <bool kCamera>
__device__ inline void fetch_ray(const DevScene &S, const PassInfo &P, const RayQueue &q, uint32_t i, V3 &o, V3 &d, uint32_t &meta)
{
    if (kCamera) {
        const bool valid = camera_ray(S, P, i, o, d);
        meta = make_meta(valid ? RK_CAMERA : RK_DEAD, BHRT_HIT_FRONT, 0);
    } else {
        o = v3(q.ox[i], q.oy[i], q.oz[i]); d = v3(q.dx[i], q.dy[i], q.dz[i]);
        meta = q.meta ? q.meta[i] : 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// Files ray i under a list of the shading order (device_types.h::RayOrder): one atomic per wave and list.
// ALL lanes of the wave must call it (cls = RC_NONE for lanes without a ray).
__device__ inline void file_ray(uint32_t cls, uint32_t i, uint32_t shard /* wave-uniform */, const RayOrder &ord, Counters *cnt)
{
    const uint32_t lane = __lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t m0 = __ballot(cls == RC_HEAVY), m1 = __ballot(cls == RC_MEDIUM), m2 = __ballot(cls == RC_LIGHT), m3 = __ballot(cls == RC_MESH);
    uint32_t base = 0;
    if (lane < 4) { // lanes 0..3 reserve room for lists 0..3
        const uint64_t mk = lane == 0 ? m0 : (lane == 1 ? m1 : (lane == 2 ? m2 : m3));
        const uint32_t c = (uint32_t)__popcll(mk);
        if (c) base = atomicAdd(&cnt->cls[lane][shard].v, c);
    }
    // broadcast the bases with ALL lanes executing the shuffles (a lane-0..3 ray may itself be listless)
    const uint32_t wshard = shard;
    const uint32_t b0 = __shfl(base, 0), b1 = __shfl(base, 1), b2 = __shfl(base, 2), b3 = __shfl(base, 3);
    if (cls != RC_NONE) {
        const uint64_t mm = cls == RC_HEAVY ? m0 : (cls == RC_MEDIUM ? m1 : (cls == RC_LIGHT ? m2 : m3));
        const uint32_t pos = (cls == RC_HEAVY ? b0 : (cls == RC_MEDIUM ? b1 : (cls == RC_LIGHT ? b2 : b3))) + (uint32_t)__popcll(mm & lt);
        if (pos < ord.shard_cap) ord.idx[((size_t)cls * BHRT_ORDER_SHARDS + wshard) * ord.shard_cap + pos] = i;
        else atomicOr(&cnt->overflow.v, 8u);
    }
}
// shading class of a finished ray; same predicate as k_shade's `new_frame`
__device__ inline uint32_t shading_class(uint32_t meta, const Hit &hit)
{
    const uint32_t kind = meta & 15u;
    const bool is_hit = hit.node >= 0;
    bool heavy = false;
    if (is_hit) {
        if (kind == RK_CAMERA) heavy = true;
        else if (kind == RK_GI) heavy = fabsf(hit.t) > BHRT_BIAS;
        else if (kind == RK_REFR_IN) heavy = hit.front != 0;
        else heavy = true;
    }
    return heavy ? RC_HEAVY : ((kind == RK_REFR_IN && is_hit) ? RC_MEDIUM : RC_LIGHT);
}

// Closest hit of every queued ray.  meta == nullptr: every ray uses `uniform_side` (public bhrt_trace_closest_*).
// kPark (scenes with meshes, render path): rays that reach a mesh whose root box they hit are parked on list RC_MESH
// with their state in the hit buffer (front = front | (node + 1) << 8) and finished by k_trace_mesh.
// Rays set aside by k_trace_closest<kPark>: parallel to a coordinate axis of the mesh they enter (trace_closest::park_slow).  One of them
// keeps a single lane busy for up to ~100 ms on the 100 k-triangle mesh while the launch's other 10^7 rays are done after a few ms — and the
// wave step cannot end before it.  They are taken out of their wave step and traced by k_trace_slow on a second stream while the pass goes
// on; when the pass's queue has run empty they are moved into it with their hits and shaded, and their paths continue in wave steps of
// their own.  Nothing about a path depends on WHEN its rays are traced: same hit records, same frames, same radiance bits.
struct SlowQueue {
    RayQueue q;
    uint32_t cap;
};
constexpr uint32_t kSlowCap = 1u << 16; // per pass; rays beyond it are simply traced in their own wave step
template <bool kPark, bool kCamera, bool kMeshes = true>
__global__ void __launch_bounds__(kBlock) k_trace_closest(DevScene S, PassInfo P, RayQueue q, uint32_t n, int uniform_side, HitBuf h, RayOrder ord, Counters *cnt,
                                                          SlowQueue slow /* cap 0: nothing is set aside */)
{
    __shared__ bhrt_bvh_node nodelet[(kPark || !kMeshes) ? 1 : BHRT_LDS_NODES]; // top BVH levels of the mesh being traversed (device_trace.h)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1);
    uint32_t meta = 0;
    if (active) fetch_ray<kCamera>(S, P, q, i, o, d, meta);
    const bool has_meta = kCamera || q.meta;
    const int side = has_meta ? (int)((meta >> 4) & 3u) : uniform_side;
    const bool dead = has_meta && (meta & 15u) == RK_DEAD;
    Hit hit;
    uint32_t key = 0;
    bool is_slow = false;
    int parked = trace_closest<kMeshes>(S, o, d, side, hit, active && !dead, (kPark || !kMeshes) ? nullptr : nodelet, 0, kPark, kPark ? &key : nullptr,
                                        (uint16_t *)nullptr, 0, BHRT_LDS_NODES, kCamera, kPark ? &is_slow : nullptr); // uniform call: the block stages nodelets together
    if (kPark && parked >= 0 && is_slow && slow.cap) {
        const uint32_t k = atomicAdd(&cnt->n_slow.v, 1u);
        if (k < slow.cap) { // the ray as a queue entry of its own (a camera ray: owner = its sample slot); traced from scratch in the slow steps
            slow.q.ox[k] = o.x; slow.q.oy[k] = o.y; slow.q.oz[k] = o.z; slow.q.dx[k] = d.x; slow.q.dy[k] = d.y; slow.q.dz[k] = d.z;
            slow.q.frame[k] = kCamera ? i : q.frame[i];
            slow.q.meta[k] = meta;
            slow.q.rng_ctr[k] = (kCamera || (meta & 15u) == RK_GI) ? 0u : q.rng_ctr[i];
            active = false; // not a ray of this wave step any more: neither finished nor filed for shading
            parked = -1;
        }
    }
    if (active) { h.t[i] = hit.t; h.node[i] = hit.node; h.prim[i] = hit.prim; h.front[i] = hit.front | ((parked + 1) << 8); }
    if (kPark && parked >= 0) {
        ord.park_key[i] = key;
        // histogram of the counting sort (the host cleared it); what the atomic returns is the ray's rank inside its bucket: kept, it makes the scatter
        // pass a read of the scanned histogram instead of a second round of atomics.  Camera rays are not sorted.
        if (!kCamera) ord.park_rank[i] = atomicAdd(&ord.park_bucket[key], 1u);
    }
    if (!ord.idx) return; // public trace API: no shading order wanted (uniform)
    uint32_t cls = RC_NONE;
    if (active && !dead) cls = parked >= 0 ? (uint32_t)RC_MESH : shading_class(meta, hit);
    file_ray(cls, i, (i >> 10) & (BHRT_ORDER_SHARDS - 1) /* 16 consecutive waves share a shard: keeps camera-ray neighbourhoods together */, ord, cnt);
}
// Slice `b` (kBlock entries) of the parked list RC_MESH (32 shard segments, each padded to whole slices; table by
// k_mesh_prefix).  Called by a whole workgroup; false = b is beyond the last slice (uniform).
__device__ inline bool parked_entry(const RayOrder &ord, uint32_t b, uint32_t *s_seg, uint32_t &i, uint32_t first_entry = 0 /* of the slice: a workgroup narrower than kBlock takes part of one */)
{
    __syncthreads(); // *s_seg of the previous slice no longer in use
    if (threadIdx.x < 64) {
        const uint32_t l = threadIdx.x;
        const uint32_t a = l <= BHRT_ORDER_SHARDS ? ord.mesh_start[l] : 0xffffffffu;
        const uint32_t n_le = (uint32_t)__popcll(__ballot(l < BHRT_ORDER_SHARDS && a <= b));
        if (l == 0) *s_seg = n_le - 1;
    }
    __syncthreads();
    if (b >= ord.mesh_start[BHRT_ORDER_SHARDS]) return false;
    const uint32_t seg = *s_seg;
    const uint32_t local = (b - ord.mesh_start[seg]) * kBlock + first_entry + threadIdx.x;
    const bool active = local < ord.mesh_count[seg];
    i = active ? ord.idx[((size_t)RC_MESH * BHRT_ORDER_SHARDS + seg) * ord.shard_cap + local] : 0xffffffffu;
    return true;
}
// XCD-local launch order (a speed choice, never correctness: MI355X_MICROARCH.md "Workgroup dispatch, XCD placement").  Workgroups are
// dealt round-robin over the 8 XCDs, so blocks b and b + 8 share an XCD and its 4 MiB L2.  With the work slices of a sorted list taken
// in blockIdx order every L2 sees every eighth slice of the range in flight.  BHRT_XCD_LOCAL = G > 0: within every run of 8 G slices
// the blocks of one XCD take G consecutive slices instead (the last, partial run keeps blockIdx order).  0 = blockIdx order.
// Measured: one contiguous eighth of the whole list per XCD — the largest G — costs more than it saves: the cost
// along a sorted list is anything but uniform (one XCD ends up with the photon focus, or with the octant most rays travel in).
// false: this block has no slice.
#ifndef BHRT_XCD_LOCAL
#define BHRT_XCD_LOCAL 0
#endif
__device__ inline bool xcd_slice(uint32_t b, uint32_t nb, uint32_t &slice)
{
    slice = b;
#if BHRT_XCD_LOCAL
    constexpr uint32_t G = BHRT_XCD_LOCAL, span = 8 * G;
    const uint32_t chunk = b / span, in = b % span;
    if ((chunk + 1) * span <= nb) slice = chunk * span + (in & 7u) * G + (in >> 3);
#endif
    return b < nb;
}
// counting sort of the parked rays by coherence key: histogram (k_trace_closest, as it parks), scan (k_scan_*), scatter.  A fixed grid strides over
// the slices: the number of parked rays is only known on the device, and a grid sized for "all rays parked" spends
// most of a launch retiring empty workgroups.
__global__ void __launch_bounds__(kBlock) k_park_scatter(RayOrder ord)
{
    __shared__ uint32_t s_seg;
    uint32_t i;
    for (uint32_t b = blockIdx.x; parked_entry(ord, b, &s_seg, i); b += gridDim.x)
        if (i != 0xffffffffu) ord.park_sorted[ord.park_bucket[ord.park_key[i]] + ord.park_rank[i]] = i;
}
// The parked rays in key order, one dense workgroup per kBlock of them: resume at the mesh node, finish the scene
// graph, file the ray under its shading class (shard = workgroup mod 32: k_trace_closest left room for that, see
// EnsureWorkspace).
// kPath: the traversal keeps its path in LDS (mesh_closest_vote; every mesh has ids < 2^17 and depth <= 32 — the host checks).
// Six waves per SIMD: the kernel waits on dependent node fetches with 8 of 64 lanes busy; more waves in flight are worth the
// few spilled registers (5 -> 6 waves: -4 %; 7: no further gain), so the nodelet is 256 nodes here (8 KB + 17 KB of path).
constexpr uint32_t kMeshNodelet = 256;
#ifndef BHRT_MESH_BLOCK
#define BHRT_MESH_BLOCK 64 /* threads per workgroup of the key-sorted (non-camera) launches: a wave leaves as soon as its own rays are done (256: +3 % on the closed room) */
#endif
constexpr int kMeshBlock = BHRT_MESH_BLOCK;
template <bool kCamera, int kPath, bool kLS = false> // kPath: 0 parent links, 1 path in LDS with 16-bit entries, 2 with 32-bit entries; kLS: bhrt_opts::leaf_skip
__global__ void __launch_bounds__(kCamera ? kBlock : kMeshBlock, kPath == 1 ? 6 : 4 /* the 32-bit path stack / the nodelet leave room for four */) k_trace_mesh(DevScene S, PassInfo P, RayQueue q, HitBuf h, RayOrder ord, Counters *cnt)
{
    typedef typename std::conditional<kPath == 2, uint32_t, uint16_t>::type PathT;
    constexpr int kTB = kCamera ? kBlock : kMeshBlock;
    __shared__ bhrt_bvh_node nodelet[kPath ? 1 : kMeshNodelet]; // the LDS-path traversal reads its nodes from global memory
    __shared__ PathT path[kPath ? 33 * kTB : 1];
    __shared__ uint32_t s_seg;
    bool active;
    uint32_t i;
    uint32_t slice;
    if (kCamera) {
        // camera rays are parked in slot order (pixel-major, all samples of a pixel together): already coherent, so the
        // list is taken as filed and the key sort (2.6 ms per pass for 66 M slots) is skipped
        if (!xcd_slice(blockIdx.x, ord.mesh_start[BHRT_ORDER_SHARDS], slice)) return; // uniform per workgroup
        if (!parked_entry(ord, slice, &s_seg, i)) return;
        active = i != 0xffffffffu;
        if (!active) i = 0;
    } else {
        const uint32_t total = ord.mesh_count[BHRT_ORDER_SHARDS];
        if (!xcd_slice(blockIdx.x, (total + kTB - 1) / kTB, slice)) return; // uniform per workgroup
        const uint32_t k = slice * kTB + threadIdx.x;
        active = k < total;
        i = active ? ord.park_sorted[k] : 0u;
    }
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1);
    uint32_t meta = 0;
    Hit hit = {BHRT_BIGFLOAT, -1, -1, 1};
    int start = 0;
    if (active) {
        fetch_ray<kCamera>(S, P, q, i, o, d, meta);
        hit.t = h.t[i]; hit.node = h.node[i]; hit.prim = h.prim[i];
        const int fw = h.front[i];
        hit.front = fw & 0xff;
        start = (fw >> 8) - 1;
    }
    trace_closest<true, PathT, kLS>(S, o, d, (int)((meta >> 4) & 3u), hit, active, kPath ? nullptr : nodelet, active ? start : S.n_nodes, false, nullptr,
                               kPath ? path + threadIdx.x : (PathT *)nullptr, kTB, kMeshNodelet);
    if (active) { h.t[i] = hit.t; h.node[i] = hit.node; h.prim[i] = hit.prim; h.front[i] = hit.front; }
    // key-sorted rays are filed for shading by k_file_parked, in queue order instead of traversal order
    if (kCamera) file_ray(active ? shading_class(meta, hit) : (uint32_t)RC_NONE, i, slice & (BHRT_ORDER_SHARDS - 1), ord, cnt);
}
// The key-sorted parked rays of a later wave step, STREAMED through resident waves.  In k_trace_mesh a wave holds 64 rays until the
// longest of them is through: over the wave's life 47 % of its lanes still have a ray (closed room, measured).  Here a wave takes the
// next entries of the sorted list (one atomic on a cursor) whenever kStreamRefill of its lanes are free, so its lanes stay filled until
// the list runs out.  A lane's ray goes through the same operation sequence as in trace_closest(start): the mesh walk it was parked at
// (device_trace.h::walk_begin / walk_round, the rounds of mesh_closest_vote), then the remaining scene nodes in index order; a wave walks
// ONE mesh at a time (M stays in scalar registers), lanes that reach another mesh node wait until the current walks are through.
#ifndef BHRT_STREAM_REFILL
#define BHRT_STREAM_REFILL 32 /* lanes that have to be free before the wave takes new rays (16: +1-3 %; 24-48: the same) */
#endif
constexpr int kStreamRefill = BHRT_STREAM_REFILL;
#ifndef BHRT_STREAM_OCC
#define BHRT_STREAM_OCC 6 /* waves per SIMD the streaming kernels are compiled for */
#endif
#ifdef BHRT_DEBUG_DRAIN
__device__ unsigned long long g_stream_t[4]; // wall clock (100 MHz): first wave in, first wave that found the list exhausted, last wave out
#endif
#ifdef BHRT_DEBUG_STREAM
__device__ unsigned long long g_stream_dbg[8]; // rounds, lanes with a walk, lanes in the round's phase, clocks in rounds, clocks outside, refills, descend / leaf rounds
#endif
template <int kPath, bool kLS = false> // 1: 16-bit path entries, 2: 32-bit (see k_trace_mesh)
__global__ void __launch_bounds__(64, kPath == 1 ? BHRT_STREAM_OCC : (BHRT_STREAM_OCC < 5 ? BHRT_STREAM_OCC : 5) /* 32-bit path entries: LDS holds five waves per SIMD */) k_trace_mesh_stream(DevScene S, RayQueue q, HitBuf h, RayOrder ord, Counters *cnt)
{
    typedef typename std::conditional<kPath == 2, uint32_t, uint16_t>::type PathT;
    __shared__ PathT path[33 * 64];
    const uint32_t total = ord.mesh_count[BHRT_ORDER_SHARDS];
    const uint32_t lane = threadIdx.x;
    PathT *stack = path + lane;
    bool have = false, pending = false, exhausted = false;
    uint32_t i = 0;
    int n = 0, side = 0;
    Hit hit = {BHRT_BIGFLOAT, -1, -1, 1};
    MeshWalk W = {};
    W.st = 3;
    int cur_n = -1; // wave-uniform: the scene node whose mesh the wave walks
    MeshRef M = mesh_ref(S, 0);
#ifdef BHRT_DEBUG_STREAM
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long c_mark = __builtin_readcyclecounter();
#endif
#ifdef BHRT_DEBUG_DRAIN
    if (lane == 0) atomicMin(&g_stream_t[0], wall_clock64());
    bool told = false;
#endif
    for (;;) {
        // (1) lanes through with their mesh: the rest of the scene graph (trace_closest's node loop from n + 1)
        if (have && !pending && W.st == 3) {
            const V3 o = v3(q.ox[i], q.oy[i], q.oz[i]), d = v3(q.dx[i], q.dy[i], q.dz[i]);
            if (W.any) hit.node = n;
            W.any = false;
            for (n++; n < S.n_nodes; n++) {
                const int type = S.nodes[n].obj_type;
                if (type == BHRT_OBJ_NONE) continue;
                if (type == BHRT_OBJ_MESH) { pending = true; break; }
                V3 lp = o, ld = d;
                local_ray(S, n, lp, ld);
                float t;
                int fr;
                if (type == BHRT_OBJ_SPHERE) {
                    if (sphere_hit(lp, ld, side, hit.t, t, fr)) { hit.t = t; hit.node = n; hit.prim = -1; hit.front = fr; }
                } else if (type == BHRT_OBJ_PLANE) {
                    if (plane_hit(lp, ld, side, hit.t, t, fr)) { hit.t = t; hit.node = n; hit.prim = -1; hit.front = fr; }
                }
            }
            if (!pending) { h.t[i] = hit.t; h.node[i] = hit.node; h.prim[i] = hit.prim; h.front[i] = hit.front; have = false; }
        }
        // (2) free lanes take the next rays of the list
        if (!exhausted) {
            const uint64_t idle = __ballot(!have);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle >= (uint32_t)kStreamRefill) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&cnt->mesh_cursor.v, n_idle);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                exhausted = base + n_idle >= total;
#ifdef BHRT_DEBUG_STREAM
                dbg[5]++;
#endif
#ifdef BHRT_DEBUG_DRAIN
                if (exhausted && !told && lane == 0) { atomicMin(&g_stream_t[1], wall_clock64()); }
                told = told || exhausted;
#endif
                const uint32_t k = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!have && k < total) {
                    i = ord.park_sorted[k];
                    hit.t = h.t[i]; hit.node = h.node[i]; hit.prim = h.prim[i];
                    const int fw = h.front[i];
                    hit.front = fw & 0xff;
                    n = (fw >> 8) - 1; // the mesh node the ray was parked at
                    side = (int)(((q.meta ? q.meta[i] : 0u) >> 4) & 3u);
                    have = true;
                    pending = true;
                }
            }
        }
        if (!__ballot(have)) break; // the list has run out (a wave whose lanes are all free has just asked for more)
        // (3) lanes waiting at a mesh node enter its BVH
        const uint64_t waiting = __ballot(pending);
        if (!__ballot(W.st != 3) && waiting) {
            const int nn = __builtin_amdgcn_readlane(n, __ffsll((unsigned long long)waiting) - 1);
            if (nn != cur_n) { cur_n = nn; M = mesh_ref(S, S.nodes[cur_n].mesh); }
        }
        if (pending && n == cur_n) {
            V3 lp = v3(q.ox[i], q.oy[i], q.oz[i]), ld = v3(q.dx[i], q.dy[i], q.dz[i]);
            local_ray(S, n, lp, ld);
            walk_begin(M, W, lp, ld, hit.t);
            pending = false;
        }
        // (4) rounds of the walk, until enough lanes are free again
        const int n_wait = __popcll(__ballot(pending));
#ifdef BHRT_DEBUG_STREAM
        c_mark = __builtin_readcyclecounter();
#endif
        for (;;) {
            const int nD = __popcll(__ballot(W.st == 0)), nL = __popcll(__ballot(W.st == 1)), nC = __popcll(__ballot(W.st == 2));
            const int walking = nD + nL + nC;
            if (walking == 0 || (!exhausted && 64 - walking - n_wait >= kStreamRefill)) break;
#ifdef BHRT_DEBUG_STREAM
#if BHRT_FUSED_CLIMB
            dbg[0]++; dbg[1] += walking; dbg[2] += (nD >= nL && nD > 0) ? nD : nL;
            if (nD >= nL && nD > 0) dbg[6]++; else if (nL > 0) dbg[7]++;
#else
            dbg[0]++; dbg[1] += walking; dbg[2] += (nD >= nL && nD >= nC) ? nD : (nL >= nC ? nL : nC);
            if (nD >= nL && nD >= nC) dbg[6]++; else if (nL >= nC) dbg[7]++;
#endif
#endif
#ifdef BHRT_DEBUG_STREAM
            const unsigned long long r0 = __builtin_readcyclecounter();
#endif
            walk_round<PathT, kLS>(M, W, side, hit.t, hit.prim, hit.front, stack, 64u, nD, nL, nC);
#ifdef BHRT_DEBUG_STREAM
            if (nD >= nL && nD > 0) dbg[4] += __builtin_readcyclecounter() - r0; // descend rounds
#endif
        }
#ifdef BHRT_DEBUG_STREAM
        { const unsigned long long c = __builtin_readcyclecounter(); dbg[3] += c - c_mark; c_mark = c; }
#endif
    }
#ifdef BHRT_DEBUG_STREAM
    if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_stream_dbg[k], dbg[k]);
#endif
#ifdef BHRT_DEBUG_DRAIN
    if (lane == 0) atomicMax(&g_stream_t[2], wall_clock64());
#endif
}
// Files the finished mesh rays of a later wave step under their shading class, walking the parked list as it was filed
// (queue order): k_shade then reads rays, hits and parent frames of neighbouring queue slots together (filed in the
// key-sorted traversal order its loads scatter: C3 k_shade 15.5 vs 13.5 ms).
__global__ void __launch_bounds__(kBlock) k_file_parked(RayQueue q, HitBuf h, RayOrder ord, Counters *cnt)
{
    __shared__ uint32_t s_seg;
    uint32_t i;
    if (!parked_entry(ord, blockIdx.x, &s_seg, i)) return; // uniform per workgroup
    const bool active = i != 0xffffffffu;
    uint32_t cls = RC_NONE;
    if (active) {
        Hit hit;
        hit.t = h.t[i]; hit.node = h.node[i]; hit.prim = h.prim[i]; hit.front = h.front[i];
        cls = shading_class(q.meta[i], hit);
    }
    file_ray(cls, active ? i : 0u, blockIdx.x & (BHRT_ORDER_SHARDS - 1), ord, cnt);
}

// The rays of the slow queue [a, b), traced to the end on a stream of their own while the pass goes on: ONE WAVE PER RAY
// (device_trace.h::mesh_closest_coop: the lanes walk 64 subtrees of the BVH side by side, then the recursion is put together from the
// root) — a walk of the whole 100 k-triangle tree in ~2 ms instead of the ~100 ms a single lane needs for its 10^5 dependent rounds.
__global__ void __launch_bounds__(64) k_trace_slow(DevScene S, SlowQueue slow, uint32_t a, uint32_t b, HitBuf h)
{
    __shared__ CoopLds lds;
    __builtin_amdgcn_s_setprio(3); // a chain of dependent steps beside a full GPU: first in line for its SIMD's issue slots
    const uint32_t i = a + blockIdx.x;
    if (i >= b) return; // uniform
    const V3 o = v3(slow.q.ox[i], slow.q.oy[i], slow.q.oz[i]), d = v3(slow.q.dx[i], slow.q.dy[i], slow.q.dz[i]);
    const uint32_t meta = slow.q.meta[i];
    Hit hit;
#ifdef BHRT_DEBUG_LONG_RAYS
    const unsigned long long c0 = wall_clock64();
#endif
    trace_closest_coop(S, lds, o, d, (int)((meta >> 4) & 3u), hit);
#ifdef BHRT_DEBUG_LONG_RAYS
    const double ms = (double)(wall_clock64() - c0) / 1e5; // 100 MHz
    if (threadIdx.x == 0 && ms > 5.0) printf("slow ray %u: %.1f ms o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) side %u -> node %d prim %d t %.9g\n", i, ms, o.x, o.y, o.z, d.x, d.y, d.z, (meta >> 4) & 3u, hit.node, hit.prim, hit.t);
#endif
    if (threadIdx.x == 0) { h.t[i] = hit.t; h.node[i] = hit.node; h.prim[i] = hit.prim; h.front[i] = hit.front; }
}
// Files rays whose hits are already there (the slow queue, moved into the ray queue) for shading.
// Rays [src0, src0 + m) of the slow queue with their hits (k_trace_slow) into slots [dst0, dst0 + m) of a wave step's queue
__global__ void __launch_bounds__(kBlock) k_inject_slow(RayQueue sq, HitBuf sh, uint32_t src0, uint32_t m, RayQueue q, HitBuf h, uint32_t dst0)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint32_t a = src0 + k, b = dst0 + k;
    q.ox[b] = sq.ox[a]; q.oy[b] = sq.oy[a]; q.oz[b] = sq.oz[a]; q.dx[b] = sq.dx[a]; q.dy[b] = sq.dy[a]; q.dz[b] = sq.dz[a];
    q.frame[b] = sq.frame[a]; q.meta[b] = sq.meta[a]; q.rng_ctr[b] = sq.rng_ctr[a];
    h.t[b] = sh.t[a]; h.node[b] = sh.node[a]; h.prim[b] = sh.prim[a]; h.front[b] = sh.front[a];
}
// files rays [first, n) (their hits are there) under their shading classes
__global__ void __launch_bounds__(kBlock) k_file_all(RayQueue q, HitBuf h, uint32_t first, uint32_t n, RayOrder ord, Counters *cnt)
{
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cls = RC_NONE;
    if (i < n) {
        Hit hit;
        hit.t = h.t[i]; hit.node = h.node[i]; hit.prim = h.prim[i]; hit.front = h.front[i];
        cls = shading_class(q.meta[i], hit);
    }
    file_ray(cls, i, (i >> 10) & (BHRT_ORDER_SHARDS - 1), ord, cnt);
}

// frame == nullptr: visibility goes to vis[i] (public bhrt_trace_shadow_*), else to vis[frame[i]]
// n_dev != nullptr: the queue length is read on the device (launched ahead of the host's copy of the counters with a
// grid sized for the upper bound n)
template <bool kMeshes> // false: scene without meshes (trace_shadow_t<3>)
__global__ void __launch_bounds__(kBlock) k_trace_shadow(DevScene S, ShadowQueue q, uint32_t n, const uint32_t *n_dev, float *vis)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (i >= n) return;
    const V3 o = v3(q.ox[i], q.oy[i], q.oz[i]), d = v3(q.dx[i], q.dy[i], q.dz[i]);
    const float v = kMeshes ? trace_shadow_t<0>(S, o, d, q.tmax[i]) : trace_shadow_t<3>(S, o, d, q.tmax[i]);
    vis[q.frame ? q.frame[i] : i] = v;
}
// Scenes with meshes, render path: spheres and planes here; a ray they leave unoccluded that enters a mesh's root box
// is parked on list RC_MESH and decided by k_shadow_mesh in dense workgroups (same reason as k_trace_mesh).
__global__ void __launch_bounds__(kBlock) k_trace_shadow_park(DevScene S, ShadowQueue q, uint32_t n, const uint32_t *n_dev, float *vis, RayOrder ord, Counters *cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (blockIdx.x * blockDim.x >= n) return; // uniform per workgroup
    float v = 1.f;
    if (i < n) {
        v = trace_shadow_t<1>(S, v3(q.ox[i], q.oy[i], q.oz[i]), v3(q.dx[i], q.dy[i], q.dz[i]), q.tmax[i]);
        if (v != 2.f) vis[q.frame[i]] = v;
    }
    file_ray(v == 2.f ? (uint32_t)RC_MESH : (uint32_t)RC_NONE, i, (i >> 10) & (BHRT_ORDER_SHARDS - 1), ord, cnt);
}
#ifndef BHRT_SHADOW_BLOCK
#define BHRT_SHADOW_BLOCK 64 /* threads per workgroup of k_shadow_mesh (a divisor of kBlock): a wave leaves as soon as its own rays are done (256: any-hit group +6-7 % on C3 and the closed room) */
#endif
constexpr int kShadowBlock = BHRT_SHADOW_BLOCK;
template <int kPath, bool kLS = false> // kPath: traversal path in LDS (mesh_shadow_stack), same modes as k_trace_mesh
__global__ void __launch_bounds__(kShadowBlock) k_shadow_mesh(DevScene S, ShadowQueue q, float *vis, RayOrder ord)
{
    typedef typename std::conditional<kPath == 2, uint32_t, uint16_t>::type PathT;
    __shared__ uint32_t s_seg;
    __shared__ PathT path[kPath ? 33 * kShadowBlock : 1];
    constexpr uint32_t kParts = kBlock / kShadowBlock; // workgroups per slice of kBlock entries
    uint32_t i, slice;
    if (!xcd_slice(blockIdx.x / kParts, ord.mesh_start[BHRT_ORDER_SHARDS], slice)) return;
    if (!parked_entry(ord, slice, &s_seg, i, (blockIdx.x % kParts) * kShadowBlock)) return;
    if (i == 0xffffffffu) return;
    vis[q.frame[i]] = trace_shadow_t<2, PathT, kLS>(S, v3(q.ox[i], q.oy[i], q.oz[i]), v3(q.dx[i], q.dy[i], q.dz[i]), q.tmax[i], kPath ? path + threadIdx.x : (PathT *)nullptr, kShadowBlock);
}

// segment table of the parked mesh rays for k_trace_mesh: 32 shards, one lane each
__global__ void __launch_bounds__(64) k_mesh_prefix(Counters *cnt, RayOrder ord)
{
    const uint32_t s = threadIdx.x;
    uint32_t c = 0;
    if (s < BHRT_ORDER_SHARDS) {
        c = cnt->cls[RC_MESH][s].v;
        cnt->cls[RC_MESH][s].v = 0;
        if (c > ord.shard_cap) c = ord.shard_cap;
    }
    const uint32_t blocks = (c + kBlock - 1) / kBlock;
    uint32_t incl = blocks;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if ((int)s >= off) incl += t; }
    uint32_t tot = c;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(tot, off); if ((int)s >= off) tot += t; }
    if (s < BHRT_ORDER_SHARDS) { ord.mesh_start[s] = incl - blocks; ord.mesh_count[s] = c; }
    if (s == BHRT_ORDER_SHARDS - 1) { ord.mesh_start[BHRT_ORDER_SHARDS] = incl; ord.mesh_count[BHRT_ORDER_SHARDS] = tot; cnt->mesh_cursor.v = 0; }
}

// segment table of the shading order for k_shade (one tiny workgroup per wave step)
// Runs between the trace kernel and k_shade of a wave step; also puts the per-step counters back to zero (class counters
// for the next trace, queue counters for the k_shade that follows), which saves a memset per step.
__global__ void __launch_bounds__(128) k_order_prefix(Counters *cnt, RayOrder ord)
{
    // 96 segments (class-major), one lane each: read + reset its counter, then exclusive scans of the segments' workgroup
    // counts and (heavy class) element counts
    constexpr uint32_t kSegs = 3 * BHRT_ORDER_SHARDS;
    __shared__ uint32_t s_blocks[128], s_frames[128];
    const uint32_t s = threadIdx.x;
    uint32_t c = 0;
    if (s < kSegs) {
        c = cnt->cls[s / BHRT_ORDER_SHARDS][s % BHRT_ORDER_SHARDS].v;
        cnt->cls[s / BHRT_ORDER_SHARDS][s % BHRT_ORDER_SHARDS].v = 0;
        if (c > ord.shard_cap) c = ord.shard_cap; // the writer flagged the overflow; never index past the segment
    }
    if (s == 0) { cnt->n_next.v = 0; cnt->n_shadow.v = 0; }
    const uint32_t blocks = (c + kShadeBlock - 1) / kShadeBlock;
    const uint32_t frames = s < BHRT_ORDER_SHARDS ? c : 0u; // every ray of the heavy class opens exactly one Shade() frame
    s_blocks[s] = blocks;
    s_frames[s] = frames;
    __syncthreads();
    for (uint32_t off = 1; off < 128; off <<= 1) { // Hillis-Steele inclusive scans
        const uint32_t add = s >= off ? s_blocks[s - off] : 0u, addf = s >= off ? s_frames[s - off] : 0u;
        __syncthreads();
        s_blocks[s] += add;
        s_frames[s] += addf;
        __syncthreads();
    }
    const uint32_t incl = s_blocks[s], excl = incl - blocks;
    // Frame numbers without atomics: the heavy rays of segment s get frames [n_frames + (heavy rays before s), ...) in the
    // order of the segment's list; k_shade adds the ray's position in the list.
    const uint32_t base = cnt->n_frames.v;
    if (s < kSegs) { ord.seg_start[s] = excl; ord.seg_count[s] = c; ord.frame_base[s] = base + s_frames[s] - frames; }
    __syncthreads(); // every lane has read n_frames
    if (s == kSegs - 1) {
        ord.seg_start[kSegs] = incl;
        cnt->n_frames.v = base + s_frames[s]; // the host's frame marks; may exceed the capacity: k_shade flags that as overflow
    }
}

// k_shade's capacity flags: a RETURNING atomic whose result is consumed, so that the wave has the acknowledgement before it goes on — the flag
// is then in place when the workgroup takes its ticket (the last ticket holder publishes the flags; a fire-and-forget atomic may still be in flight)
__device__ inline void flag_overflow(Counters *cnt, uint32_t bit)
{
    const uint32_t old = atomicOr(&cnt->overflow.v, bit);
    asm volatile("" ::"v"(old));
}
// The step's queue lengths for the host, written straight into pinned host memory (no copy engine packet, no event:
// each of those costs the stream ~6 us of idle GPU per wave step).  The host spins on `seq` (WaitPublished).
struct HostCounters { uint32_t n_next, n_shadow, n_frames, overflow, n_slow; uint32_t pad[11]; uint32_t seq; };
__device__ inline void publish_counters(Counters *cnt, HostCounters *pub, uint32_t seq) // one lane
{
    volatile HostCounters *p = pub;
    auto ld = [](uint32_t *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }; // other workgroups' atomics, this launch or an earlier one
    p->n_next = ld(&cnt->n_next.v); p->n_shadow = ld(&cnt->n_shadow.v); p->n_frames = ld(&cnt->n_frames.v); p->overflow = ld(&cnt->overflow.v); p->n_slow = ld(&cnt->n_slow.v);
    __threadfence_system();
    __hip_atomic_store(&pub->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}


// ------------------------------------------------------------------------------------------------
// One Shade() entry (MtlBlinn.cpp:89-138): fills frame f, produces up to two closest rays and one shadow ray.
struct ShadeOut {
    bool has_refr, has_gi, has_shadow;
    V3 ro, rd; uint32_t rmeta, rctr; // refraction ray
    V3 go, gd;                       // GI ray
    V3 so, sd; float stmax;          // shadow ray
};

// kTex = false: the scene has no texture map at all (every TexturedColor is its plain colour): the texture sampling code
// (a quarter of the kernel, 200 divisions) is not even compiled in.
template <bool kTex>
__device__ inline void shade_entry(const DevScene &S, const RenderParams &R, const Frames &F, uint32_t f, uint32_t how, V3 rayP, const Attr &a, int node,
                                   int bounce, int gi, uint64_t code, uint32_t skey, ShadeOut &out)
{
    out.has_refr = out.has_gi = out.has_shadow = false;
    const int mi = S.nodes[node].material;
    uint32_t flags = 0, dmode = DM_NONE, light_idx = 0;
    V3 zero = v3(0, 0, 0);
    // no zero-fill of the term arrays: the flags in `info` say which terms exist (k_combine reads only those)
    if (R.photon) st3(F.caustic, f, zero);
    if (mi < 0 || S.materials[mi].kind != BHRT_MTL_BLINN) {
        // node without material: black (the reference dereferences null); empty MultiMtl: white (materials.h:71)
        flags = FF_CONST;
        st3(F.refr, f, (mi >= 0 && S.materials[mi].kind == BHRT_MTL_WHITE) ? v3(1, 1, 1) : zero);
        F.info[f] = how | (flags << 16);
        return;
    }
    const bhrt_material &m = S.materials[mi];
    V3 vN = normalized(a.N);
    V3 vV = normalized(rayP - a.p);
    float cosPhi1 = dot(vN, vV);
    if (cosPhi1 > 1) cosPhi1 = 1;
    if (cosPhi1 <= 0) cosPhi1 = 0;
    const float ior = m.ior;
    // pow(float,int) is double pow in the reference (MtlBlinn.cpp:107-109); device math: repeated product
    const float R0 = S.mat_r0[mi]; // (float)pow((1 - ior) / (1 + ior), 2): pow(float,int) is double pow in the reference (MtlBlinn.cpp:107)
    double omc = (double)(1 - cosPhi1);
    float fresnel = (float)(R0 + (1 - R0) * ((((omc * omc) * omc) * omc) * omc));
    V3 refrC = ld3(m.refraction.color);
    V3 fresSpec = clamp_white(ld3(m.specular.color) + fresnel * refrC);
    bhrt_texcolor newSpecular = m.specular;
    newSpecular.color[0] = fresSpec.x; newSpecular.color[1] = fresSpec.y; newSpecular.color[2] = fresSpec.z;
    float refrGloss = 0;
    if (m.glossiness > 50) refrGloss = m.glossiness;

    // ---- refraction, PathTracing_Refraction (MtlBlinn.cpp:437-473)
    V3 refraction = (1 - fresnel) * refrC;
    if (bounce > 0 && !is_zero(refraction)) {
        DRng g;
        g.key = bhrt_section_key(skey, code, BHRT_SEC_REFRACTION);
        g.ctr = 0;
        float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
        float sinPhi2 = sinPhi1 / ior;
        float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
        V3 vTn = (-cosPhi2) * vN;
        V3 vNxV = cross(vN, vV);
        V3 vTp = normalized(cross(vN, vNxV)) * sinPhi2;
        V3 vT = vTn + vTp;
        V3 vT_sampled = normalized(vT);
        if (refrGloss > 0) {
            float dotSign = 0;
            int guard = 0;
            while (dotSign >= 0 && guard++ < BHRT_MAXLOOP) {
                float theta = 0;
                vT_sampled = sample_along_light_direction(g, vT, refrGloss, theta);
                dotSign = dot(vT_sampled, vN);
            }
        }
        out.has_refr = true;
        out.rd = normalized(vT_sampled);
        out.ro = a.p - vN * BHRT_BIAS; // RefractionRecusive, MtlBlinn.cpp:478-480
        out.rmeta = make_meta(RK_REFR_IN, BHRT_HIT_FRONT_AND_BACK, bounce);
        out.rctr = g.ctr;
        st3(F.refr_color, f, refraction);
        flags |= FF_HAS_REFR;
    }
    // ---- global illumination, PathTracing_GlobalIllumination (MtlBlinn.cpp:383-433)
    const bool textured = kTex && (m.diffuse.map >= 0 || newSpecular.map >= 0);
    // diffuse.Sample(uvw, duvw) / specular.Sample(uvw, duvw): the reference evaluates them again at every use (GI multiplier,
    // direct term, caustic term) with the same arguments — 32 filter taps each; once is enough
    V3 kd_s = ld3(m.diffuse.color), ks_s = ld3(newSpecular.color);
    if (textured) {
        kd_s = tc_sample_d(S, m.diffuse, a.uvw, a.du, a.dv);
        ks_s = tc_sample_d(S, newSpecular, a.uvw, a.du, a.dv);
    }
    if (gi >= 0) {
        DRng g;
        g.key = bhrt_section_key(skey, code, BHRT_SEC_GI);
        g.ctr = 0;
        bool useSpecular;
        out.gd = gi_direction(g, useSpecular, vN, vV, tc_max(m.diffuse.color), tc_max(newSpecular.color), m.glossiness);
        out.go = a.p + vN * BHRT_BIAS;
        out.has_gi = true;
        flags |= FF_HAS_GI;
        const bhrt_texcolor &tcs = useSpecular ? newSpecular : m.diffuse;
        st3(F.gi_mult, f, useSpecular ? ks_s : kd_s);
    }
    // ---- direct light, PathTracing_DiffuseNSpecular (MtlBlinn.cpp:304-351)
    if (S.n_lights > 0) {
        DRng g;
        g.key = bhrt_section_key(skey, code, BHRT_SEC_DIRECT);
        g.ctr = 0;
        float rnd = g.rnd01();
        int li = 0;
        while (rnd > S.light_pick[li] && li < S.n_lights - 1) li++;
        const bhrt_light &light = S.lights[li];
        light_idx = (uint32_t)li;
        V3 vL = sample_in_light(g, m.diffuse.color, newSpecular.color, light, a.p, m.glossiness);
        float cosTheta = dot(vL, vN);
        if (cosTheta > 0) {
            V3 vH = normalized(vL + vV);
            // Illuminate: lights.h:32,50; PointLight.cpp:7-18
            if (light.type == BHRT_LIGHT_AMBIENT) dmode = DM_AMBIENT;
            else if (light.type == BHRT_LIGHT_DIRECT) {
                dmode = DM_DIRECT;
                out.has_shadow = true;
                out.so = a.p; out.sd = -ld3(light.vec); out.stmax = BHRT_BIGFLOAT;
            } else {
                V3 centerDir = ld3(light.vec) - a.p;
                float r = length(centerDir);
                float rr = r * r;
                if (rr == 0) dmode = DM_POINT_ZERO;
                else {
                    dmode = DM_POINT;
                    F.rr[f] = rr;
                    out.has_shadow = true;
                    out.so = a.p;
                    out.sd = light.size > 0 ? sample_along_normal(g, centerDir, light.size) : centerDir;
                    out.stmax = 1;
                }
            }
            const V3 kd = kd_s, ks = ks_s;
            st3(F.brdf, f, kd * cosTheta + ks * dm::powf_(dot(vH, vN), m.glossiness));
        }
    }
    if (R.photon) { // inputs of the caustic term (MtlBlinn.cpp:329-342), evaluated by k_photon_gather_frames
        st3(F.ph_p, f, a.p); st3(F.ph_n, f, a.N); st3(F.ph_v, f, vV);
        st3(F.ph_kd, f, kd_s);
        st3(F.ph_ks, f, ks_s);
    }
    F.info[f] = how | (dmode << 3) | (light_idx << 8) | (flags << 16) | ((uint32_t)(mi & 0xfff) << 20); // the frame's only write of info
}

#ifndef BHRT_SHADE_WAVES
#define BHRT_SHADE_WAVES 4 /* waves per SIMD the register allocation must allow */
#endif
// kFused (camera step of a scene without meshes): the kernel traces its camera rays itself, in slot order — no k_trace_closest in front of it (which
// computes the same camera ray, ~300 instructions, and writes 24-byte hit records this kernel reads back), no shading order (all samples of a pixel
// sit in one wave: hits and misses are as uniform per workgroup in slot order as in the sorted one), frame numbers from one atomic per workgroup.
template <bool kCamera, bool kTex, bool kFused = false>
__device__ __forceinline__ void shade_block(const DevScene &S, const RenderParams &R, const PassInfo &P, const RayQueue &qin, const HitBuf &hb, uint32_t n, const RayQueue &qout,
                                            const ShadowQueue &qs, const Frames &F, float *samples, Counters *cnt, const RayOrder &ord)
{
    static_assert(!kFused || kCamera, "only camera rays are traced in the shading kernel");
    __shared__ BlockAllocLds lds;
    uint32_t seg = 0, local = 0, i;
    bool active;
    if (kFused) {
        i = blockIdx.x * kShadeBlock + threadIdx.x;
        active = i < n;
        if (!active) i = 0;
    } else {
        // this workgroup's slice of the shading order: segments (class, shard) in class-major order, each padded to whole
        // workgroups; seg_start[] (k_order_prefix) is ascending, the segment is the last one starting at or before blockIdx
        constexpr uint32_t kSegs = 3 * BHRT_ORDER_SHARDS;
        __shared__ uint32_t s_seg;
        if (threadIdx.x < 64) {
            const uint32_t l = threadIdx.x;
            const uint32_t a = ord.seg_start[l], b = (l + 64 < kSegs) ? ord.seg_start[l + 64] : 0xffffffffu;
            const uint32_t n_le = (uint32_t)__popcll(__ballot(a <= blockIdx.x)) + (uint32_t)__popcll(__ballot(b <= blockIdx.x));
            if (l == 0) s_seg = n_le - 1; // seg_start[0] == 0 <= blockIdx always
        }
        __syncthreads();
        seg = s_seg;
        if (blockIdx.x >= ord.seg_start[kSegs]) return; // uniform per workgroup: beyond the last segment
        const uint32_t seg_cnt = ord.seg_count[seg];
        const uint32_t bb = blockIdx.x - ord.seg_start[seg];
        local = bb * kShadeBlock + threadIdx.x;
        active = local < seg_cnt;
        i = active ? ord.idx[(size_t)seg * ord.shard_cap + local] : 0u;
    }
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1);
    uint32_t owner = 0, meta = 0, ctr = 0;
    Hit hit = {BHRT_BIGFLOAT, -1, -1, 1};
    if (active) {
        fetch_ray<kCamera>(S, P, qin, i, o, d, meta);
        active = (meta & 15u) != RK_DEAD;
    }
    if (kFused) // recursive() (Main.cpp:389) for the camera ray, as k_trace_closest<false, true, false> runs it
        trace_closest<false>(S, o, d, BHRT_HIT_FRONT, hit, active, nullptr, 0, false, nullptr, (uint16_t *)nullptr, 0, BHRT_LDS_NODES, true);
    if (active) {
        if (kCamera) { owner = i; ctr = 0; } // the sample slot
        else { owner = qin.frame[i]; ctr = (meta & 15u) != RK_GI ? qin.rng_ctr[i] : 0u; }
        if (!kFused) { hit.t = hb.t[i]; hit.node = hb.node[i]; hit.prim = hb.prim[i]; hit.front = hb.front[i]; }
    }
    const uint32_t kind = meta & 15u;
    int bounce = (int)((meta >> 8) & 0xffu);
    const bool is_hit = active && hit.node >= 0;
    // which rays open a new Shade() frame
    bool new_frame = false;
    if (is_hit) {
        if (kind == RK_CAMERA) new_frame = true;
        else if (kind == RK_GI) new_frame = fabsf(hit.t) > BHRT_BIAS; // MtlBlinn.cpp:401
        else if (kind == RK_REFR_IN) new_frame = hit.front != 0;      // MtlBlinn.cpp:507-510
        else new_frame = true;                                        // RK_REFR_OUT, MtlBlinn.cpp:527-533
    }
    // frame number = the segment's first frame (k_order_prefix) + the ray's place in the heavy list: no atomic, no barrier.
    // new_frame holds for every ray of a heavy segment and for no other (k_trace_closest files rays with this predicate).
    uint32_t f, u2;
    if (kFused) { uint32_t u0, u1; block_alloc3(lds, &cnt->n_frames.v, new_frame ? 1u : 0u, 0u, nullptr, 0u, nullptr, 0u, f, u0, u1, u2); } // frame numbers: one atomic per workgroup
    else f = ord.frame_base[seg] + local;
    if (new_frame && f >= R.cap_frames) { flag_overflow(cnt, 1u); new_frame = false; }

    ShadeOut so;
    so.has_refr = so.has_gi = so.has_shadow = false;
    uint32_t ray_owner = owner; // frame that owns the rays emitted by this lane

    if (new_frame) {
        // ---- child (or root) frame bookkeeping
        uint64_t code = 1;
        uint32_t skey;
        int gi;
        uint32_t how;
        V3 mult = v3(1, 1, 1);
        if (kind == RK_CAMERA) {
            int pi, pj;
            uint32_t opx, osmp;
            fdivmod(owner, P.by_spp, opx, osmp);
            pixel_of(P, P.q0 + opx, pi, pj);
            skey = bhrt_sample_key(P.seed, (uint32_t)(pj * P.W + pi), osmp);
            gi = R.gi_bounces;
            bounce = R.internal_bounces;
            how = FH_ROOT;
        } else {
            const uint64_t pcode = F.code[owner];
            skey = F.skey[owner];
            const uint32_t pi2 = F.info2[owner];
            gi = (int)(pi2 & 0xffu) - 64 - 1;
            if (kind == RK_GI) {
                code = pcode * 2 + 1;
                how = FH_GI;
                bounce = (int)((pi2 >> 8) & 0xffu); // Shade(GIRay, ..., o_bounceCount, gi-1): the parent's own bounce count
            } else if (kind == RK_REFR_IN) {
                code = pcode * 2;
                how = FH_REFR_FRONT;
            } else {
                code = pcode * 2;
                how = FH_REFR_OUT;
                // Beer absorption, RefractionOut (MtlBlinn.cpp:529-533): refraction * absorptionFactor
                const bhrt_material &pm = S.materials[(F.info[owner] >> 20) & 0xfffu];
                V3 af = v3(dm::powf_(BHRT_EULER, -pm.absorption[0] * hit.t), dm::powf_(BHRT_EULER, -pm.absorption[1] * hit.t),
                           dm::powf_(BHRT_EULER, -pm.absorption[2] * hit.t));
                mult = ld3i(F.refr_color, owner) * af;
            }
        }
        F.parent[f] = owner;
        F.info2[f] = (uint32_t)(gi + 64) | ((uint32_t)(bounce & 0xff) << 8);
        F.skey[f] = skey;
        F.code[f] = code;
        if (how == FH_REFR_OUT) st3(F.mult, f, mult); // a GI frame's multiplier stays where it is: the parent's gi_mult (k_combine)
        const int mi = S.nodes[hit.node].material;
        const bool need_uv = kTex && mi >= 0 && (S.materials[mi].diffuse.map >= 0 || S.materials[mi].specular.map >= 0);
        Attr a;
        hit_attrs(S, o, d, hit.t, hit.node, hit.prim, need_uv, a);
        shade_entry<kTex>(S, R, F, f, how, o, a, hit.node, bounce, gi, code, skey, so);
        ray_owner = f;
    } else if (active) {
        if (kind == RK_CAMERA) {
            // background.Sample((i/W, j/H, 0)), Main.cpp:166-167
            int pi, pj;
            pixel_of(P, P.q0 + fdiv(owner, P.by_spp), pi, pj);
            st3(samples, sample_addr(P, owner), kTex ? tc_sample(S, S.background, v3((float)pi / S.cam.width, (float)pj / S.cam.height, 0.0f)) : ld3(S.background.color));
        } else if (kind == RK_GI) {
            V3 mult = ld3i(F.gi_mult, owner);
            V3 outc = v3(0, 0, 0);
            if (is_hit) {
                outc = outc + v3(0, 0, 0) * mult; // |z| <= Bias: indirect stays black (MtlBlinn.cpp:398-406)
            } else if (d.x == d.y && d.x == 0) {
                outc = outc + v3(1.0f, 0.0f, 1.0f); // MtlBlinn.cpp:411-415
            } else {
                V3 env = (kTex ? sample_environment(S, S.environment, d) : ld3(S.environment.color)) * mult;
                if (!(isnan_f(env.x) || isnan_f(env.y) || isnan_f(env.z))) outc = outc + env;
            }
            if (isnan_f(outc.x)) outc = v3(1.0f, 0.0f, 1.0f);
            else outc = clamp_white(outc);
            st3(F.gi, owner, outc);
        } else if (kind == RK_REFR_IN) {
            if (!is_hit) {
                st3(F.refr, owner, v3(1.0f, 0.0f, 1.0f)); // RefractionRecusive returns NANPurple (MtlBlinn.cpp:517)
            } else {
                // back face: HandleRayWhenRefractionRayOut (MtlBlinn.cpp:543-589)
                Attr a;
                hit_attrs(S, o, d, hit.t, hit.node, hit.prim, false, a);
                const bhrt_material &m = S.materials[(F.info[owner] >> 20) & 0xfffu];
                const float refrGloss = m.glossiness > 50 ? m.glossiness : 0.f;
                V3 vN = a.N;
                V3 vV = -d;
                float cosPhi1 = dot(vV, -vN);
                float sinPhi1 = sqrtf(1 - cosPhi1 * cosPhi1);
                float sinPhi2 = m.ior * sinPhi1;
                if (sinPhi2 <= 1) {
                    float cosPhi2 = sqrtf(1 - sinPhi2 * sinPhi2);
                    V3 vTn = vN * cosPhi2;
                    V3 vNxV = cross(vN, vV);
                    V3 vTp = normalized(cross(vN, vNxV)) * sinPhi2;
                    V3 vT = vTn + vTp;
                    V3 vT_sampled = normalized(vT);
                    if (refrGloss > 0) {
                        DRng g;
                        g.key = bhrt_section_key(F.skey[owner], F.code[owner], BHRT_SEC_REFRACTION);
                        g.ctr = ctr;
                        float dotSign = 0;
                        int guard = 0;
                        while (dotSign <= 0 && guard++ < BHRT_MAXLOOP) {
                            float theta = 0;
                            vT_sampled = sample_along_light_direction(g, vT, refrGloss, theta);
                            dotSign = dot(vT_sampled, vN);
                        }
                        ctr = g.ctr;
                    }
                    so.has_refr = true;
                    so.rd = normalized(vT_sampled);
                    so.ro = a.p + vN * BHRT_BIAS;
                    so.rmeta = make_meta(RK_REFR_OUT, BHRT_HIT_FRONT, bounce);
                    so.rctr = ctr;
                } else if (bounce <= 0) {
                    st3(F.refr, owner, v3(0, 0, 0)); // MtlBlinn.cpp:496-500
                } else {
                    V3 vR = ((-2 * cosPhi1) * vN - vV); // total internal reflection, MtlBlinn.cpp:580-587,502-503
                    so.has_refr = true;
                    so.rd = vR;
                    so.ro = a.p - vN * BHRT_BIAS;
                    so.rmeta = make_meta(RK_REFR_IN, BHRT_HIT_FRONT_AND_BACK, bounce - 1);
                    so.rctr = ctr;
                }
            }
        } else { // RK_REFR_OUT miss: refraction * environment (MtlBlinn.cpp:535-539)
            st3(F.refr, owner, clamp_white(ld3i(F.refr_color, owner) * (kTex ? sample_environment(S, S.environment, d) : ld3(S.environment.color))));
        }
    }

    // ---- convergent pushes
    uint32_t r0, r1, s0;
    block_alloc3(lds, &cnt->n_next.v, so.has_refr ? 1u : 0u, so.has_gi ? 1u : 0u, &cnt->n_shadow.v, so.has_shadow ? 1u : 0u, nullptr, 0u, r0, r1, s0, u2);
    if (so.has_refr) {
        if (r0 < R.cap_rays) put_ray(qout, r0, so.ro, so.rd, ray_owner, so.rmeta, so.rctr);
        else flag_overflow(cnt, 2u);
    }
    if (so.has_gi) {
        if (r1 < R.cap_rays) put_ray(qout, r1, so.go, so.gd, ray_owner, make_meta(RK_GI, BHRT_HIT_FRONT, 0), 0);
        else flag_overflow(cnt, 2u);
    }
    if (so.has_shadow) {
        if (s0 < R.cap_shadow) {
            qs.ox[s0] = so.so.x; qs.oy[s0] = so.so.y; qs.oz[s0] = so.so.z; qs.dx[s0] = so.sd.x; qs.dy[s0] = so.sd.y; qs.dz[s0] = so.sd.z;
            qs.tmax[s0] = so.stmax; qs.frame[s0] = ray_owner;
        } else flag_overflow(cnt, 4u);
    }
}
// The workgroup that finishes LAST hands the step's queue lengths to the host (what a one-lane kernel behind k_shade did: k_publish, ~6 us of
// launch and ~5 us of gap per wave step — 2 % of a C2 frame).  Every workgroup's counter updates are atomics at agent scope and come before its
// ticket (release fence); the last ticket holder reads them with atomic loads behind an acquire fence.
template <bool kCamera, bool kTex, bool kFused = false>
__global__ void __launch_bounds__(kShadeBlock, BHRT_SHADE_WAVES) k_shade(DevScene S, RenderParams R, PassInfo P, RayQueue qin, HitBuf hb, uint32_t n, RayQueue qout,
                                                   ShadowQueue qs, Frames F, float *samples, Counters *cnt, RayOrder ord, HostCounters *pub, uint32_t seq)
{
    shade_block<kCamera, kTex, kFused>(S, R, P, qin, hb, n, qout, qs, F, samples, cnt, ord);
    if (!pub) return;
    __syncthreads(); // every wave is through: its queue counters were added to by returning atomics whose results it has used (block_alloc), its
    // capacity flags likewise (flag_overflow) — all acknowledged.  No agent-scope fence: only counters travel, all by atomics at agent scope; a release
    // fence here writes the XCD's whole L2 back, per workgroup (measured: C2 frame 4.9 -> 9.8 ms).
    if (threadIdx.x == 0 && atomicAdd(&cnt->shade_done.v, 1u) == gridDim.x - 1u) {
        __hip_atomic_store(&cnt->shade_done.v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next step's launch
        publish_counters(cnt, pub, seq);
    }
}

// ------------------------------------------------------------------------------------------------
// Fold frames [f0, f1) (all created in one wave step) into their parents / the sample buffer.
__global__ void __launch_bounds__(kBlock) k_combine(DevScene S, PassInfo P, Frames F, uint32_t f0, uint32_t f1, float *samples, int photon)
{
    const uint32_t f = f0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= f1) return;
    const uint32_t info = F.info[f];
    const uint32_t how = info & 7u, dmode = (info >> 3) & 7u, li = (info >> 8) & 0xffu, flags = (info >> 16) & 0xfu;
    V3 out;
    if (flags & FF_CONST) out = ld3i(F.refr, f);
    else {
        // Shade(), MtlBlinn.cpp:117-137
        out = v3(0, 0, 0);
        if (flags & FF_HAS_REFR) out = out + ld3i(F.refr, f); // else PathTracing_Refraction returned black
        bool done = out.x >= 1 && out.y >= 1 && out.z >= 1;
        if (!done) {
            if (flags & FF_HAS_GI) out = out + ld3i(F.gi, f);  // else gi < 0: black (MtlBlinn.cpp:386)
            done = out.x >= 1 && out.y >= 1 && out.z >= 1;
        }
        if (!done) {
            // PathTracing_DiffuseNSpecular, MtlBlinn.cpp:304-351
            V3 dc = v3(0, 0, 0);
            if (dmode != DM_NONE) {
                const bhrt_light &l = S.lights[li];
                V3 I = ld3(l.intensity), irrad;
                const float vis = F.vis[f];
                if (dmode == DM_AMBIENT) irrad = I;
                else if (dmode == DM_DIRECT) irrad = vis * I;                       // lights.h:50
                else if (dmode == DM_POINT) irrad = (vis * I) / F.rr[f];            // PointLight.cpp:14-17
                else irrad = v3(1, 1, 1) * BHRT_BIGFLOAT;                           // PointLight.cpp:12
                dc = dc + irrad * ld3i(F.brdf, f);
            }
            if (photon) dc = dc + ld3i(F.caustic, f);
            dc = clamp_white(dc);
            if (isnan_f(dc.x)) dc = v3(0, 0, 0);
            out = out + dc;
            done = out.x >= 1 && out.y >= 1 && out.z >= 1;
            if (!done && isnan_f(out.x)) out = v3(1.0f, 0.0f, 1.0f);
        }
    }
    const uint32_t parent = F.parent[f];
    if (how == FH_ROOT) st3(samples, sample_addr(P, parent), out);
    else if (how == FH_GI) { // MtlBlinn.cpp:406,427-432
        V3 oc = v3(0, 0, 0) + out * ld3i(F.gi_mult, parent);
        if (isnan_f(oc.x)) oc = v3(1.0f, 0.0f, 1.0f);
        else oc = clamp_white(oc);
        st3(F.gi, parent, oc);
    } else if (how == FH_REFR_FRONT) st3(F.refr, parent, clamp_white(out)); // MtlBlinn.cpp:509-511
    else st3(F.refr, parent, clamp_white(ld3i(F.mult, f) * out));           // MtlBlinn.cpp:533,539
}

// in-order sample average (Main.cpp:150-170), gamma (Main.cpp:220-226), Color24 (cyColor.h:271-272)
__global__ void __launch_bounds__(kBlock) k_resolve(PassInfo P, const float *samples, float *radiance, uint8_t *rgb8)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P.n_pixels) return;
    int i, j;
    if (!pixel_of(P, P.q0 + q, i, j)) return;
    V3 sum = v3(0, 0, 0);
    for (int s = 0; s < P.spp; s++) sum = sum + ld3i(samples, (uint32_t)s * P.n_pixels + q);
    V3 out = sum / (float)P.spp;
    const size_t pix = (size_t)j * P.W + i;
    if (radiance) st3(radiance, (uint32_t)pix, out);
    if (rgb8) {
        if (P.gamma) {
            const float inv = 1 / 2.2f;
            out = v3(dm::powf_(out.x, inv), dm::powf_(out.y, inv), dm::powf_(out.z, inv));
        }
        int r = int(out.x * 255 + 0.5f), g = int(out.y * 255 + 0.5f), b = int(out.z * 255 + 0.5f);
        rgb8[pix * 3] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        rgb8[pix * 3 + 1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
        rgb8[pix * 3 + 2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
}

// ------------------------------------------------------------------------------------------------
// Images beside the colour image: RenderImage's z-buffer (scene.h:532, the store commented out at
// Main.cpp:231) and the albedo / normal inputs DenoiseImage leaves unset (Main.cpp:70-71), all from the first hit of the
// un-jittered camera ray of every pixel (the pixel corner).  One lane per pixel, row-major.
__global__ void __launch_bounds__(kBlock) k_first_hit(DevScene S, int W, int H, float *z, float *normal, float *albedo)
{
    __shared__ bhrt_bvh_node nodelet[BHRT_LDS_NODES];
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < (uint32_t)(W * H);
    const int j = (int)(q / (uint32_t)W), i = (int)(q - (uint32_t)j * (uint32_t)W);
    const V3 topLeft = ld3(S.cam.top_left), ddx = ld3(S.cam.dd_x), ddy = ld3(S.cam.dd_y), pos = ld3(S.cam.pos);
    const V3 o = pos, d = ((topLeft + (float)i * ddx) - (float)j * ddy) - pos; // Main.cpp:145,153
    Hit hit;
    trace_closest(S, o, d, BHRT_HIT_FRONT, hit, active, nodelet); // uniform call: the block stages nodelets together
    if (!active) return;
    V3 N = v3(0, 0, 0), kd = v3(0, 0, 0);
    if (hit.node >= 0 && (normal || albedo)) {
        const int mi = S.nodes[hit.node].material;
        const bool blinn = mi >= 0 && S.materials[mi].kind == BHRT_MTL_BLINN;
        Attr a;
        hit_attrs(S, o, d, hit.t, hit.node, hit.prim, albedo && blinn && S.materials[mi].diffuse.map >= 0, a);
        N = a.N;
        if (blinn) kd = tc_sample_d(S, S.materials[mi].diffuse, a.uvw, a.du, a.dv); // diffuse.Sample(uvw, duvw), MtlBlinn.cpp:393
        else if (mi >= 0 && S.materials[mi].kind == BHRT_MTL_WHITE) kd = v3(1, 1, 1);
    }
    if (z) z[q] = hit.node >= 0 ? hit.t : BHRT_BIGFLOAT;
    if (normal) st3(normal, q, N);
    if (albedo) st3(albedo, q, kd);
}

// RenderImage::ComputeZBufferImage (scene.h:578-600): min / max over the hits, then (zmax - z) / (zmax - zmin) * 255 truncated.
// Min and max are exact in any order; one workgroup strides over the image.
__global__ void __launch_bounds__(1024) k_z_range(const float *z, uint32_t n, float *range)
{
    __shared__ float s_min[1024], s_max[1024];
    float zmin = BHRT_BIGFLOAT, zmax = 0;
    for (uint32_t k = threadIdx.x; k < n; k += 1024) {
        const float v = z[k];
        if (v == BHRT_BIGFLOAT) continue;
        if (zmin > v) zmin = v;
        if (zmax < v) zmax = v;
    }
    s_min[threadIdx.x] = zmin; s_max[threadIdx.x] = zmax;
    __syncthreads();
    for (uint32_t w = 512; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            if (s_min[threadIdx.x] > s_min[threadIdx.x + w]) s_min[threadIdx.x] = s_min[threadIdx.x + w];
            if (s_max[threadIdx.x] < s_max[threadIdx.x + w]) s_max[threadIdx.x] = s_max[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { range[0] = s_min[0]; range[1] = s_max[0]; }
}
__global__ void __launch_bounds__(kBlock) k_z_image(const float *z, uint32_t n, const float *range, uint8_t *img)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float zmin = range[0], zmax = range[1], v = z[k];
    if (v == BHRT_BIGFLOAT) { img[k] = 0; return; }
    const float f = (zmax - v) / (zmax - zmin);
    int c = f != f ? (int)0x80000000 : (f >= 2147483648.f ? (int)0x80000000 : (f * 255 <= -2147483648.f ? (int)0x80000000 : int(f * 255))); // x86 cvttss2si out-of-range / NaN value
    if (c < 0) c = 0;
    if (c > 255) c = 255;
    img[k] = (uint8_t)c;
}
// colorArray of BeginRender (Main.cpp:202,219-229): the gamma-corrected pixel colour as floats = DenoiseImage's "color" input
__global__ void __launch_bounds__(kBlock) k_color_image(const float *radiance, uint32_t n3, int gamma, float *color)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n3) return;
    const float c = radiance[k];
    color[k] = gamma ? dm::powf_(c, 1 / 2.2f) : c;
}

// test hook: the deterministic math of bhrt_detmath.h evaluated on the device (tests compare its bits with the host's)
__global__ void k_math_eval(int fn, const float *a, const float *b, uint32_t n, float *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 0.f;
    float r = 0;
    switch (fn) {
    case 0: r = dm::sinf_(x); break;
    case 1: r = dm::cosf_(x); break;
    case 2: r = dm::tanf_(x); break;
    case 3: r = dm::acosf_(x); break;
    case 4: r = dm::asinf_(x); break;
    case 5: r = dm::atan2f_(x, y); break;
    case 6: r = dm::powf_(x, y); break;
    case 7: r = dm::rand_to_unit((int)dm::fbits(x)); break;
    case 8: r = x / y; break;
    case 9: r = sqrtf(x); break;
    }
    out[i] = r;
}

__global__ void k_copy_samples(PassInfo P, const float *samples, int x0, int y0, int x1, int y1, float *out)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P.n_pixels) return;
    int i, j;
    if (!pixel_of(P, P.q0 + q, i, j)) return;
    if (i < x0 || i >= x1 || j < y0 || j >= y1) return;
    const size_t pix = (size_t)(j - y0) * (x1 - x0) + (i - x0);
    for (int s = 0; s < P.spp; s++)
        for (int c = 0; c < 3; c++) out[(pix * P.spp + s) * 3 + c] = samples[((size_t)s * P.n_pixels + q) * 3 + c];
}

// ------------------------------------------------------------------------------------------------
// photon map kernels
template <bool kGlobal>
__global__ void __launch_bounds__(kBlock) k_photon_emit(DevScene S, uint32_t seed, uint64_t e0, uint32_t E, const int32_t *plights, int n_plights,
                                                         float sum_intensity, DPhoton *tmp, uint32_t cap, uint32_t *counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    counts[i] = emit_photon_path<kGlobal>(S, seed, e0 + i, plights, n_plights, sum_intensity, tmp + (size_t)i * cap, cap);
}
// stable compaction in emission order: path i's photons go to out[1 + offsets[i] + k] (slot 0 stays unused)
__global__ void __launch_bounds__(kBlock) k_photon_compact(const DPhoton *tmp, uint32_t cap, const uint32_t *counts, const uint32_t *offsets, uint32_t base, uint32_t E,
                                                            uint32_t max_photons, DPhoton *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const uint32_t c = counts[i], off = (uint32_t)min((uint64_t)base + offsets[i], (uint64_t)max_photons); // base: photons of the earlier batches
    for (uint32_t k = 0; k < c && k < cap; k++)
        if (off + k < max_photons) out[1 + off + k] = tmp[(size_t)i * cap + k];
}
__global__ void __launch_bounds__(kBlock) k_photon_scale(DPhoton *photons, uint32_t n, float scale) // ScalePhotonPowers, cyPhotonMap.h:119
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) photons[1 + i].power *= scale;
}
// caustic term of frames [f0, f1): brdf * irradiance (MtlBlinn.cpp:329-342) -> F.caustic
// decode every photon once (hot / cold split, see PhotonMapDev)
__global__ void __launch_bounds__(kBlock) k_photon_expand(const DPhoton *photons, uint32_t n, float4 *hot, float4 *cold)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const DPhoton p = photons[i];
    const V3 d = photon_direction(p), pw = photon_power(p);
    hot[i] = make_float4(p.pos[0], p.pos[1], p.pos[2], __uint_as_float((uint32_t)(p.planeAndDirZ & 0x3)));
    cold[2 * (size_t)i] = make_float4(d.x, d.y, d.z, p.power);
    cold[2 * (size_t)i + 1] = make_float4(pw.x, pw.y, pw.z, 0.f);
}

// What a gather kernel does with one finished query.
struct GatherToFrames { // the caustic term of Shade(), MtlBlinn.cpp:329-342
    Frames F;
    const bhrt_material *materials;
    __device__ bool skip(uint32_t f) const { return ((F.info[f] >> 16) & FF_CONST) != 0; }
    __device__ V3 pos(uint32_t f) const { return ld3i(F.ph_p, f); }
    __device__ V3 nrm(uint32_t f) const { return ld3i(F.ph_n, f); }
    __device__ void done(uint32_t f, bool found, V3 irr, V3 vL) const
    {
        if (!found) return; // no photon: irrad = 0 and vL = (0,0,0) -> cosTheta = -0 is not > 0: no contribution (F.caustic stays 0)
        const V3 vN = normalized(ld3i(F.ph_n, f));
        const float cosTheta = -dot(vL, vN);
        if (cosTheta > 0) {
            const V3 vH = normalized(vL + ld3i(F.ph_v, f));
            const float gloss = materials[(F.info[f] >> 20) & 0xfffu].glossiness;
            const V3 brdf = ld3i(F.ph_kd, f) + ld3i(F.ph_ks, f) * dm::powf_(dot(vH, vN), gloss) / cosTheta;
            st3(F.caustic, f, brdf * irr);
        }
    }
};
struct GatherToArrays { // bhrt_photon_gather_host
    const float *p, *n;
    float *irrad, *dir;
    __device__ bool skip(uint32_t) const { return false; }
    __device__ V3 pos(uint32_t i) const { return ld3i(p, i); }
    __device__ V3 nrm(uint32_t i) const { return ld3i(n, i); }
    __device__ void done(uint32_t i, bool, V3 irr, V3 d) const { st3(irrad, i, irr); st3(dir, i, d); }
};

// Gather order.  A wave whose 64 queries lie in one small cell walks the same tree path and touches the same photons
// (coalesced loads, no divergence), so the queries are counting-sorted by grid cell first: 128^3 cells over the photons'
// bounds (+ radius), cells numbered along a Morton curve so that neighbouring waves share cache lines too.  The order
// inside a cell is whatever the atomics give: every query's result is independent of the order.
#define BHRT_GATHER_CELL_BITS 9
#define BHRT_GATHER_CELLS (1u << (3 * BHRT_GATHER_CELL_BITS))
struct GatherGrid { float lo[3], inv_cell[3]; };
__device__ inline uint32_t spread3(uint32_t v) // up to 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ inline uint32_t gather_cell(const GatherGrid &G, V3 p)
{
    const float m = (float)((1 << BHRT_GATHER_CELL_BITS) - 1);
    const float fx = fminf(fmaxf((p.x - G.lo[0]) * G.inv_cell[0], 0.f), m), fy = fminf(fmaxf((p.y - G.lo[1]) * G.inv_cell[1], 0.f), m),
                fz = fminf(fmaxf((p.z - G.lo[2]) * G.inv_cell[2], 0.f), m); // NaN -> 0 (fmaxf returns the number)
    return spread3((uint32_t)fx) | (spread3((uint32_t)fy) << 1) | (spread3((uint32_t)fz) << 2);
}
// Queries that cannot meet a photon — farther than the radius from the photons' bounds (most pixels of a caustic scene), or
// frames without the term — are answered here and left out of the order: pass 1 then runs over the rest only.
#define BHRT_GATHER_NO_CELL 0xffffffffu
// One add per wave and cell instead of one per lane: the 64 queries of a wave are mostly the samples of ONE pixel (frames are numbered in the
// order of the heavy list, which a camera wave fills with its 64 slots), i.e. one or two cells — and a lane's atomic costs a 32-byte sector
// of L2 traffic whatever it adds (5.8e8 queries per C5 frame: the two sort kernels were bound by exactly that).  Up to kCellLeaders cells per wave
// are handled by a leader each; lanes of further cells add for themselves.  Returns the lane's slot in its cell (the value its own atomicAdd
// would have returned, up to the order among the lanes of one wave); all lanes of the wave must call it (active = has a cell).
constexpr int kCellLeaders = 4;
__device__ inline uint32_t cell_add_wave(uint32_t *cells, uint32_t c, bool active)
{
    const uint32_t lane = __lane_id();
    uint64_t todo = __ballot(active);
    uint32_t slot = 0;
    for (int k = 0; k < kCellLeaders && todo; k++) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, lead);
        const uint64_t same = __ballot(active && c == cl) & todo;
        uint32_t base = 0;
        if ((int)lane == lead) base = atomicAdd(&cells[cl], (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, lead);
        if ((same >> lane) & 1ull) slot = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) slot = atomicAdd(&cells[c], 1u);
    return slot;
}
// The slot cell_add_wave returns in the COUNT pass is the query's rank inside its cell: kept (rank_of), it makes the scatter pass a plain read of
// the scanned counter instead of a second round of atomics on the 512 MB table (round 3).
template <class Sink>
__global__ void __launch_bounds__(kBlock) k_gather_cell_count(Sink sink, uint32_t q0, uint32_t cnt, GatherGrid G, PhotonMapDev M, float radius, uint32_t *cell_of,
                                                              uint32_t *rank_of, uint32_t *cell_count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = BHRT_GATHER_NO_CELL;
    if (i < cnt) {
        const uint32_t q = q0 + i;
        if (!sink.skip(q)) {
            const V3 p = sink.pos(q);
            if (photon_outside_bounds(M, p, radius)) sink.done(q, false, v3(0, 0, 0), v3(0, 0, 0)); // photon_estimate_fast's own first test: no photon, zero estimate
            else c = gather_cell(G, p);
        }
        cell_of[i] = c;
    }
    const uint32_t r = cell_add_wave(cell_count, c, c != BHRT_GATHER_NO_CELL);
    if (i < cnt) rank_of[i] = r;
}
// exclusive scan of cell_count[BHRT_GATHER_CELLS] in place, three launches: per-block sums, scan of the sums, add back
constexpr uint32_t kScanBlock = 1024, kScanPerThread = 8, kScanTile = kScanBlock * kScanPerThread;
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds /* kScanBlock/64 + 1 */, uint32_t &total)
{
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if ((int)lane >= off) incl += t; }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t w = 0; w < kScanBlock / 64; w++) { const uint32_t t = lds[w]; lds[w] = run; run += t; } lds[kScanBlock / 64] = run; }
    __syncthreads();
    const uint32_t r = lds[wave] + incl - v;
    total = lds[kScanBlock / 64];
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(kScanBlock) k_scan_tiles(uint32_t *data, uint32_t n, uint32_t *tile_sums)
{
    __shared__ uint32_t lds[kScanBlock / 64 + 1];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread], sum = 0;
    for (uint32_t k = 0; k < kScanPerThread; k++) { v[k] = base + k < n ? data[base + k] : 0u; sum += v[k]; }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, lds, total);
    for (uint32_t k = 0; k < kScanPerThread; k++) { if (base + k < n) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kScanBlock) k_scan_sums(uint32_t *tile_sums, uint32_t n_tiles) // one workgroup, chunk after chunk
{
    __shared__ uint32_t lds[kScanBlock / 64 + 1];
    uint32_t carry = 0;
    for (uint32_t c = 0; c < n_tiles; c += kScanBlock) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sums[i] : 0u;
        uint32_t total;
        const uint32_t r = block_exclusive_scan(v, lds, total);
        if (i < n_tiles) tile_sums[i] = carry + r;
        carry += total;
    }
}
__global__ void __launch_bounds__(kScanBlock) k_scan_add(uint32_t *data, uint32_t n, const uint32_t *tile_sums)
{
    const uint32_t add = tile_sums[blockIdx.x];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
    for (uint32_t k = 0; k < kScanPerThread; k++) if (base + k < n) data[base + k] += add;
}
} // namespace bhrt
#include "device_photon_build.h"
namespace bhrt {
__global__ void __launch_bounds__(kBlock) k_gather_cell_scatter(uint32_t q0, uint32_t cnt, const uint32_t *cell_of, const uint32_t *rank_of, const uint32_t *cell_start, uint32_t *order)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = i < cnt ? cell_of[i] : BHRT_GATHER_NO_CELL;
    if (c != BHRT_GATHER_NO_CELL) order[cell_start[c] + rank_of[i]] = q0 + i;
}
// The same order from a radix sort of (cell, query) pairs (gather_sort.hip): the key of a query that takes no part is one bit above every cell, so
// those sort to the end; the number that do take part = the position of the first such key (k_gather_first_out: one lane, a binary search).
#define BHRT_GATHER_KEY_OUT BHRT_GATHER_CELLS
template <class Sink>
__global__ void __launch_bounds__(kBlock) k_gather_cell_key(Sink sink, uint32_t q0, uint32_t cnt, GatherGrid G, PhotonMapDev M, float radius, uint32_t *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t q = q0 + i;
    uint32_t c = BHRT_GATHER_KEY_OUT;
    if (!sink.skip(q)) {
        const V3 p = sink.pos(q);
        if (photon_outside_bounds(M, p, radius)) sink.done(q, false, v3(0, 0, 0), v3(0, 0, 0)); // photon_estimate_fast's own first test: no photon, zero estimate
        else c = gather_cell(G, p);
    }
    keys[i] = c;
    vals[i] = q;
}
__global__ void k_gather_first_out(const uint32_t *sorted_keys, uint32_t cnt, uint32_t *out)
{
    uint32_t lo = 0, hi = cnt; // first index whose key is >= BHRT_GATHER_KEY_OUT
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (sorted_keys[mid] < BHRT_GATHER_KEY_OUT) lo = mid + 1; else hi = mid; }
    *out = lo;
}
int GatherSortPairs(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n, void *temp, size_t *temp_bytes, int end_bit,
                    hipStream_t stream); // gather_sort.hip


// Pass 1: every query walks the map without a candidate list (photon_estimate_fast).  Queries that meet their 1000th
// photon are appended to `heavy` (pass 3), queries whose walk is longer than the lane budget to `longq` (pass 2).
// order: optional permutation of the queries (cell-sorted, see above).
#define BHRT_GATHER_LANE_BUDGET 4096
#ifndef BHRT_SEL_WAVES_PER_CU
#define BHRT_SEL_WAVES_PER_CU 16 /* selection pass: one-wave workgroups per CU = 4 per SIMD (128 VGPRs), 8 KB of LDS each */
#endif
__device__ inline void wave_append(bool flag, uint32_t value, uint32_t *list, uint32_t *count)
{
    const uint64_t m = __ballot(flag);
    if (!m) return;
    uint32_t base = 0;
    if (__lane_id() == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (flag) list[base + (uint32_t)__popcll(m & ((1ull << __lane_id()) - 1ull))] = value;
}
// kd-tree nodes examined by the gather (bhrt_stats.photon_nodes_visited): one 64-bit atomic per wave
__device__ inline void count_visited(uint32_t visited, uint32_t *counts)
{
    for (int off = 32; off > 0; off >>= 1) visited += __shfl_xor(visited, off);
    if (__lane_id() == 0 && visited) atomicAdd((unsigned long long *)(counts + 2), (unsigned long long)visited);
}
template <class Sink, bool kFound = false> // kFound: the statistics instantiation (knob "gather_stats")
__global__ void __launch_bounds__(kBlock) k_photon_gather_fast(Sink sink, uint32_t q0, uint32_t cnt, const uint32_t *order, PhotonMapDev M, float radius,
                                                               int lane_budget, uint32_t *heavy, uint32_t *longq, uint32_t *counts /* [0] heavy, [1] long, [2..3] visited, [8..9] found, [10..11] answered */)
{
    uint32_t slice;
    xcd_slice(blockIdx.x, gridDim.x, slice); // cell-sorted order: one contiguous eighth of it per XCD
    const uint32_t i = slice * blockDim.x + threadIdx.x;
    int r = 0, found = 0;
    uint32_t q = 0, visited = 0;
    if (i < cnt) {
        q = order ? order[i] : q0 + i;
        if (!sink.skip(q)) {
            V3 irr, d;
            r = photon_estimate_fast<kFound>(M, sink.pos(q), sink.nrm(q), radius, lane_budget, irr, d, visited, kFound ? &found : nullptr);
            if (r < 2) sink.done(q, r == 1, irr, d);
        }
    }
    wave_append(r == 2, q, heavy, &counts[0]);
    wave_append(r == 3, q, longq, &counts[1]);
    count_visited(visited, counts);
    if (kFound) { // photons the queries answered HERE were made of (the floor of the nodes a walk has to examine), and how many queries those were
        uint32_t f = r < 2 ? (uint32_t)found : 0u, nq = (i < cnt && r < 2) ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) { f += __shfl_xor(f, off); nq += __shfl_xor(nq, off); }
        if (__lane_id() == 0 && nq) { atomicAdd((unsigned long long *)(counts + 8), (unsigned long long)f); atomicAdd((unsigned long long *)(counts + 10), (unsigned long long)nq); }
    }
}
// Exact replay: the queries of heavy[h0, h0+cnt) with the full candidate heap, one scratch column per lane.
template <class Sink>
__global__ void __launch_bounds__(kBlock) k_photon_gather_heap(Sink sink, const uint32_t *heavy, uint32_t h0, uint32_t cnt, PhotonMapDev M, float radius,
                                                               unsigned long long *scr, size_t stride, uint32_t *counts)
{
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t visited = 0;
    if (lane < cnt) {
        const uint32_t q = heavy[h0 + lane];
        V3 irr, d;
        const bool found = photon_estimate_heap(M, sink.pos(q), sink.nrm(q), radius, scr + (size_t)lane * BHRT_HEAP_COLUMN, 1, irr, d, visited);
        sink.done(q, found, irr, d);
    }
    count_visited(visited, counts);
}

// Pass 2: one wave per query (photon_estimate_select) for what the lane pass set aside — the heavy queries (>= 1000 photons in the
// radius: wave-cooperative selection) and the long walks (walk-order sums of fewer than 1000 photons).  The waves pull queries from a
// shared cursor (their costs differ by orders of magnitude).  Undecided queries go to `undecided` for the exact replay.
// exact_only: heavy queries are not decided here (bhrt_opts.photon_exact).  knn (test hook): BHRT_PHOTON_K + 2 words per query id.
template <class Sink>
__global__ void __launch_bounds__(64, BHRT_SEL_WAVES_PER_CU / 4) k_photon_gather_select(Sink sink, const uint32_t *heavy, uint32_t n_heavy, const uint32_t *longq, uint32_t n_long,
                                                             PhotonMapDev M, float radius, uint32_t *undecided,
                                                             uint32_t *counts /* [0] undecided, [1] cursor, [2..3] visited */, uint32_t *knn, uint32_t *scratch, int exact_only)
{
    __shared__ SelectLds lds;
    __shared__ uint32_t s_next;
    uint32_t visited = 0, dbg[2] = {0, 0};
    SelectScratch C;
    C.d2 = scratch + (size_t)blockIdx.x * BHRT_SEL_SCRATCH_WORDS; C.idx = C.d2 + BHRT_SEL_CAP; C.sides = C.idx + BHRT_SEL_CAP;
    C.sp_node = C.sides + BHRT_SEL_CAP; C.sp_sides = C.sp_node + BHRT_SEL_SPILL; C.sp_plane = (float *)(C.sp_sides + BHRT_SEL_SPILL);
    while (true) {
        if (threadIdx.x == 0) s_next = atomicAdd(&counts[1], 1u);
        __syncthreads();
        const uint32_t i = s_next;
        __syncthreads();
        if (i >= n_heavy + n_long) break;
        const uint32_t q = i < n_long ? longq[i] : heavy[i - n_long]; // the long walks first: the launch's tail is then made of short queries
        V3 irr, d;
        const int r = photon_estimate_select(M, lds, C, sink.pos(q), sink.nrm(q), radius, irr, d, visited, knn ? knn + (size_t)q * (BHRT_PHOTON_K + 2) : nullptr, dbg,
                                             exact_only != 0);
        if (threadIdx.x == 0) {
            if (r == 4) undecided[atomicAdd(&counts[0], 1u)] = q;
            else sink.done(q, r == 1, irr, d);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&counts[4], dbg[0]); atomicAdd(&counts[5], dbg[1]); }
    count_visited(visited, counts);
}

// ------------------------------------------------------------------------------------------------
// multi-GPU framebuffer exchange: tile t (row-major, tile x tile pixels) belongs to rank t mod world.
// A rank's block of the exchange buffer = its tiles_per_rank tiles, float radiance section first (12 B / pixel), RGB8
// section behind it (3 B / pixel); tile k of rank r is tile k * world + r; pixels outside the image are zero.
__device__ inline bool tile_pixel(uint32_t p, int W, int H, int tile, int tiles_x, int world, int rank, int &i, int &j, bool &exists)
{
    const uint32_t tp = (uint32_t)(tile * tile);
    const uint32_t k = p / tp, within = p % tp;
    const uint32_t t = k * (uint32_t)world + (uint32_t)rank;
    const int tiles_y = (H + tile - 1) / tile;
    exists = t < (uint32_t)(tiles_x * tiles_y);
    i = (int)(t % (uint32_t)tiles_x) * tile + (int)(within % (uint32_t)tile);
    j = (int)(t / (uint32_t)tiles_x) * tile + (int)(within / (uint32_t)tile);
    return exists && i < W && j < H;
}
__global__ void __launch_bounds__(kBlock) k_tiles_pack(const uint8_t *rgb8, const float *radiance, int W, int H, int tile, int tiles_x, int rank, int world,
                                                        uint32_t px_per_rank, float *out_rad, uint8_t *out_rgb)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= px_per_rank) return;
    int i, j;
    bool exists;
    const bool in = tile_pixel(p, W, H, tile, tiles_x, world, rank, i, j, exists);
    const size_t pix = (size_t)j * W + i;
    for (int c = 0; c < 3; c++) {
        out_rad[(size_t)p * 3 + c] = in ? radiance[pix * 3 + c] : 0.f;
        out_rgb[(size_t)p * 3 + c] = in ? rgb8[pix * 3 + c] : (uint8_t)0;
    }
}
__global__ void __launch_bounds__(kBlock) k_tiles_unpack(const uint8_t *gathered, size_t block_bytes, int W, int H, int tile, int tiles_x, int world, uint32_t px_per_rank,
                                                          uint8_t *rgb8, float *radiance)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; // (rank, pixel of the rank's block)
    if (g >= px_per_rank * (uint32_t)world) return;
    const int rank = (int)(g / px_per_rank);
    const uint32_t p = g % px_per_rank;
    int i, j;
    bool exists;
    if (!tile_pixel(p, W, H, tile, tiles_x, world, rank, i, j, exists)) return;
    const uint8_t *blk = gathered + (size_t)rank * block_bytes;
    const float *rad = (const float *)blk;
    const uint8_t *rgb = blk + (size_t)px_per_rank * 12;
    const size_t pix = (size_t)j * W + i;
    for (int c = 0; c < 3; c++) {
        radiance[pix * 3 + c] = rad[(size_t)p * 3 + c];
        rgb8[pix * 3 + c] = rgb[(size_t)p * 3 + c];
    }
}

// ================================================================================================
// host side
// ================================================================================================
struct DeviceState {
    int device = -1;
    uint32_t n_cus = 256; // compute units of the device (hipDeviceProp_t::multiProcessorCount): sizes the grids of the resident-wave kernels
    uint8_t *d_blob = nullptr;
    int32_t *d_chain = nullptr;
    float *d_aux = nullptr; // mat_r0 + light_pick (DevScene)
    DevScene S;
    // wavefront workspace
    uint32_t cap_samples = 0, cap_rays = 0, cap_frames = 0;
    float *d_rayf[2] = {nullptr, nullptr};     // 6 * cap_rays floats each
    uint32_t *d_rayu[2] = {nullptr, nullptr};  // 3 * cap_rays
    float *d_hitf = nullptr;                   // cap_rays
    int32_t *d_hiti = nullptr;                 // 3 * cap_rays
    float *d_shf = nullptr;                    // 7 * cap_rays
    uint32_t *d_shu = nullptr;                 // cap_rays
    uint32_t *d_fu = nullptr;                  // 4 * cap_frames
    uint64_t *d_fcode = nullptr;               // cap_frames
    float *d_ff = nullptr;                     // (3*7 + 2) * cap_frames
    float *d_samples = nullptr;                // 3 * cap_samples
    uint32_t *d_order = nullptr;               // 4 * BHRT_ORDER_SHARDS * order_shard_cap (shading order, device_types.h::RayOrder)
    uint32_t order_shard_cap = 0;
    uint32_t *d_seg = nullptr;                 // seg_start[97] + seg_count[96] + mesh_start[33] + mesh_count[33] + frame_base[96]
    uint32_t *d_park = nullptr;                // park_key[cap_rays] + park_sorted[cap_rays] + park_rank[cap_rays] + buckets + tile sums (RayOrder)
    float *d_slowf = nullptr;                  // slow queue (SlowQueue): 6 * kSlowCap floats, then kSlowCap hit distances
    uint32_t *d_slowu = nullptr;               // 3 * kSlowCap, then 3 * kSlowCap hit words (node, prim, front)
    hipStream_t stream2 = nullptr;             // k_trace_slow runs here, beside the pass
    std::vector<hipEvent_t> slow_events;       // one per k_trace_slow launch of a pass (its hits are in place)
    // any-hit work of a wave step beside the next step's closest-hit work (RenderRange, knobs.shadow_overlap): its own stream, the second shadow queue,
    // its own parked list (the RC_MESH part of a RayOrder), segment table and counters
    hipStream_t stream3 = nullptr;
    float *d_shf2 = nullptr;                   // 7 * cap_rays
    uint32_t *d_shu2 = nullptr;                // cap_rays
    uint32_t *d_order_sh = nullptr;            // BHRT_ORDER_SHARDS * order_shard_cap
    uint32_t *d_seg_sh = nullptr;              // mesh_start[33] + mesh_count[33]
    Counters *d_cnt_sh = nullptr;
    hipEvent_t ev_shade = nullptr, ev_shadow[2] = {nullptr, nullptr};
    Counters *d_cnt = nullptr;
    HostCounters *h_pub = nullptr; // pinned, device-visible: written by publish_counters
    HostCounters *d_pub = nullptr; // the device's address of h_pub
    uint32_t pub_seq = 0;
    int timers = 0;                // bhrt_opts.timers of the running call
    int photon_exact = 0;          // bhrt_opts.photon_exact of the running call
    uint32_t *d_knn = nullptr;     // test hook of bhrt_photon_gather_host_ex: the selection pass's photon lists
    uint32_t *d_sel = nullptr;     // candidate scratch of the selection pass's waves
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct PendingTimer { int e0, e1; double *acc; };
    std::vector<hipEvent_t> ev_pool;
    std::vector<PendingTimer> ev_pending;
    std::vector<int> ev_free;                  // pool indices not in use
    // caustic photon map (balanced, heap order, slot 0 unused) + gather scratch
    DPhoton *d_photons = nullptr;
    uint32_t n_photons = 0;
    std::vector<HostPhoton> h_photons; // balanced copy for bhrt_photon_export
    float *d_ph_frames = nullptr;      // 15 * cap_frames floats (p, N, V, kd, ks per frame), only with photon_map
    uint32_t ph_frames_cap = 0;
    float4 *d_ph_hot = nullptr, *d_ph_cold = nullptr, *d_ph_dbox = nullptr; // decoded copy the gather walks (PhotonMapDev)
    PhotonMapDev pm;
    unsigned long long *d_scr = nullptr; // candidate heaps of the heavy queries: (K+1) x scr_lanes, element-major
    uint32_t scr_lanes = 0;
    uint32_t *d_heavy = nullptr; // queries that met 1000 photons in pass 1
    uint32_t heavy_cap = 0;
    uint32_t *d_n_heavy = nullptr, *h_n_heavy = nullptr; // [0] heavy, [1] long
    uint32_t *d_long = nullptr;  // queries whose walk outlasted the lane budget in pass 1
    uint32_t *d_cell_of = nullptr, *d_gorder = nullptr, *d_rank_of = nullptr, *d_keys_out = nullptr; // gather order (cell sort): heavy_cap entries each
    void *d_sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    // Development switches, read from the environment ONCE, when the scene is uploaded (none changes a result), and the two test knobs, which
    // no environment variable reaches: only bhrt_scene_knob sets them.
    struct Knobs {
        int stream_waves = -1;          // BHRT_STREAM_WAVES: resident waves of k_trace_mesh_stream; 0 = the launch-per-64-rays kernel; -1 = default
        bool fused_camera = true;       // BHRT_FUSED_CAMERA=0: two-kernel camera step in mesh-free scenes
        bool no_slow_queue = false;     // BHRT_NO_SLOW_QUEUE: axis-parallel rays stay in their wave steps
        bool debug_slow = false, debug_gather = false, debug_drain = false; // BHRT_DEBUG_*: statistics on stderr
        bool balance_host = false;      // BHRT_PHOTON_BALANCE_HOST: photon_host.cpp instead of k_pb_level (the tests' second opinion)
        int frame_cap = 0;              // knob "frame_cap": a frame pool that overflows (the retry path under test); 0 = off
        int gather_lane_budget = 0;     // knob "gather_lane_budget": photons a lane may visit before its query goes to the one-wave pass; 0 = default
        bool gather_counting_sort = false; // BHRT_GATHER_COUNTING_SORT=1: the cell order by the counting sort instead of the radix sort of pairs
        bool shadow_overlap = true;     // BHRT_SHADOW_OVERLAP=0: the any-hit kernels of a wave step on the pass's own stream, in front of the next step
        int gather_stats = 0;           // knob "gather_stats": the lane pass counts the photons its answers are made of (bhrt_stats.photon_found), 7 % slower
        void FromEnv()
        {
            if (const char *e = getenv("BHRT_STREAM_WAVES")) stream_waves = atoi(e);
            if (const char *e = getenv("BHRT_FUSED_CAMERA")) fused_camera = atoi(e) != 0;
            no_slow_queue = getenv("BHRT_NO_SLOW_QUEUE") != nullptr;
            debug_slow = getenv("BHRT_DEBUG_SLOW") != nullptr; debug_gather = getenv("BHRT_DEBUG_GATHER") != nullptr; debug_drain = getenv("BHRT_DEBUG_DRAIN") != nullptr;
            if (const char *e = getenv("BHRT_PHOTON_BALANCE_HOST")) balance_host = atoi(e) != 0;
            if (const char *e = getenv("BHRT_GATHER_COUNTING_SORT")) gather_counting_sort = atoi(e) != 0;
            if (const char *e = getenv("BHRT_SHADOW_OVERLAP")) shadow_overlap = atoi(e) != 0;
        }
    } knobs;
    uint32_t *d_cells = nullptr, *d_tile_sums = nullptr;
    // a capacity overflow halves the pass (RenderRange); later frames of the same scene and options start from the reduced size
    uint64_t pass_hint_key = 0;
    uint32_t pass_hint = 0;
    uint64_t frames_seen_key = 0; // Shade() frames per sample slot the passes of a render (scene, options: the key) have needed so far (max)
    double frames_seen = 0;
    // bhrt_render: the device copy of the frame, kept between calls
    uint8_t *d_frame_rgb = nullptr;
    float *d_frame_rad = nullptr;
    size_t frame_px = 0;
    // scratch for the public trace API
    float *d_api_f = nullptr;
    int32_t *d_api_i = nullptr;
    size_t api_cap = 0;
};

void DestroyDeviceState(DeviceState *d)
{
    if (!d) return;
    if (d->device >= 0) (void)hipSetDevice(d->device);
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(d->d_blob); fr(d->d_chain);
    for (int k = 0; k < 2; k++) { fr(d->d_rayf[k]); fr(d->d_rayu[k]); }
    fr(d->d_hitf); fr(d->d_hiti); fr(d->d_shf); fr(d->d_shu); fr(d->d_fu); fr(d->d_fcode); fr(d->d_ff); fr(d->d_samples); fr(d->d_order); fr(d->d_park); fr(d->d_seg); fr(d->d_cnt); fr(d->d_aux);
    fr(d->d_frame_rgb); fr(d->d_frame_rad); fr(d->d_sel); fr(d->d_slowf); fr(d->d_slowu);
    fr(d->d_shf2); fr(d->d_shu2); fr(d->d_order_sh); fr(d->d_seg_sh); fr(d->d_cnt_sh);
    if (d->ev_shade) (void)hipEventDestroy(d->ev_shade);
    for (int k = 0; k < 2; k++) if (d->ev_shadow[k]) (void)hipEventDestroy(d->ev_shadow[k]);
    if (d->stream3) (void)hipStreamDestroy(d->stream3);
    fr(d->d_api_f); fr(d->d_api_i); fr(d->d_photons); fr(d->d_ph_frames); fr(d->d_scr); fr(d->d_ph_hot); fr(d->d_ph_cold); fr(d->d_ph_dbox); fr(d->d_heavy); fr(d->d_long); fr(d->d_n_heavy); fr(d->d_cell_of); fr(d->d_gorder); fr(d->d_rank_of); fr(d->d_keys_out); if (d->d_sort_temp) (void)hipFree(d->d_sort_temp); fr(d->d_cells); fr(d->d_tile_sums);
    if (d->h_n_heavy) (void)hipHostFree(d->h_n_heavy);
    if (d->h_pub) (void)hipHostFree(d->h_pub);
    for (int k = 0; k < 2; k++) if (d->ev[k]) (void)hipEventDestroy(d->ev[k]);
    for (hipEvent_t e : d->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->slow_events) (void)hipEventDestroy(e);
    if (d->stream2) (void)hipStreamDestroy(d->stream2);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

static RayQueue MakeRayQueue(float *f, uint32_t *u, size_t cap)
{
    RayQueue q;
    q.ox = f; q.oy = f + cap; q.oz = f + 2 * cap; q.dx = f + 3 * cap; q.dy = f + 4 * cap; q.dz = f + 5 * cap;
    q.frame = u; q.meta = u ? u + cap : nullptr; q.rng_ctr = u ? u + 2 * cap : nullptr;
    return q;
}

static int EnsureUploaded(bhrt_scene *scene)
{
    if (!scene) { SetError("null scene"); return BHRT_ERR_ARG; }
    if (scene->dev) { HIP_CHECK(hipSetDevice(scene->dev->device)); return BHRT_OK; }
    return bhrt_scene_upload(scene, 0);
}

static int EnsureWorkspace(DeviceState *D, uint32_t cap_samples, double frames_per_sample)
{
    const size_t cf_wanted = (size_t)std::ceil((double)cap_samples * frames_per_sample);
    if (cf_wanted > 0xffffffffull) { SetError("frame pool beyond 2^32 frames"); return BHRT_ERR_ARG; }
    if (D->cap_samples >= cap_samples && D->cap_frames >= cf_wanted) return BHRT_OK;
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    for (int k = 0; k < 2; k++) { fr(D->d_rayf[k]); fr(D->d_rayu[k]); D->d_rayf[k] = nullptr; D->d_rayu[k] = nullptr; }
    fr(D->d_shf2); fr(D->d_shu2); fr(D->d_order_sh); D->d_shf2 = nullptr; D->d_shu2 = nullptr; D->d_order_sh = nullptr;
    fr(D->d_hitf); fr(D->d_hiti); fr(D->d_shf); fr(D->d_shu); fr(D->d_fu); fr(D->d_fcode); fr(D->d_ff); fr(D->d_samples); fr(D->d_order); fr(D->d_park);
    D->d_order = nullptr; D->d_park = nullptr;
    D->d_hitf = nullptr; D->d_hiti = nullptr; D->d_shf = nullptr; D->d_shu = nullptr; D->d_fu = nullptr; D->d_fcode = nullptr; D->d_ff = nullptr; D->d_samples = nullptr;
    D->cap_samples = 0;
    const size_t cr = (size_t)cap_samples * 2, cf = cf_wanted;
    for (int k = 0; k < 2; k++) {
        HIP_CHECK(hipMalloc(&D->d_rayf[k], cr * 6 * sizeof(float)));
        HIP_CHECK(hipMalloc(&D->d_rayu[k], cr * 3 * sizeof(uint32_t)));
    }
    HIP_CHECK(hipMalloc(&D->d_hitf, cr * sizeof(float)));
    HIP_CHECK(hipMalloc(&D->d_hiti, cr * 3 * sizeof(int32_t)));
    HIP_CHECK(hipMalloc(&D->d_shf, cr * 7 * sizeof(float)));
    HIP_CHECK(hipMalloc(&D->d_shu, cr * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&D->d_fu, cf * 4 * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&D->d_fcode, cf * sizeof(uint64_t)));
    HIP_CHECK(hipMalloc(&D->d_ff, cf * 23 * sizeof(float)));
    HIP_CHECK(hipMalloc(&D->d_samples, (size_t)cap_samples * 3 * sizeof(float)));
    // rays [1024 g, 1024 g + 1024) file under shard g mod 32 (k_trace_closest); k_trace_mesh files its workgroups round-robin:
    // at most (all parked rays) / 32 + one workgroup more per shard -> twice the even share always fits
    D->order_shard_cap = (uint32_t)(2 * ((((cr + 1023) / 1024 + BHRT_ORDER_SHARDS - 1) / BHRT_ORDER_SHARDS) * 1024 + 1024));
    HIP_CHECK(hipMalloc(&D->d_order, (size_t)4 * BHRT_ORDER_SHARDS * D->order_shard_cap * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&D->d_park, (3 * cr + (1u << BHRT_PARK_KEY_BITS) + kScanBlock) * sizeof(uint32_t)));
    if (D->stream3) {
        HIP_CHECK(hipMalloc(&D->d_shf2, cr * 7 * sizeof(float)));
        HIP_CHECK(hipMalloc(&D->d_shu2, cr * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_order_sh, (size_t)BHRT_ORDER_SHARDS * D->order_shard_cap * sizeof(uint32_t)));
    }
    D->cap_samples = cap_samples; D->cap_rays = (uint32_t)cr; D->cap_frames = (uint32_t)cf;
    if (D->d_ph_frames) { (void)hipFree(D->d_ph_frames); D->d_ph_frames = nullptr; D->ph_frames_cap = 0; }
    return BHRT_OK;
}

// The largest power of two of camera samples in flight (<= 2^28) whose wavefront buffers fit into 85 % of what the device has free, the present
// workspace counted as free (EnsureWorkspace releases it before it allocates).  Per sample: two ray slots (72 B each: 36 B in each of the two
// queues), their hit (16 B) and shadow (32 B) slots, shading order and park lists (40 B per ray slot), in scenes with meshes the second shadow queue and
// the any-hit kernels' own parked list (32 + 8 B: RenderRange, "any-hit work beside the pass"), 12 B of radiance, and `frames_per_sample`
// Shade() frames (116 B each, + 60 B with the photon map): six by default = 1.03-1.11 KB per sample, 2^27 samples = 138-149 GB.
static uint32_t DefaultPassSamples(DeviceState *D, bool photon_map, double frames_per_sample)
{
    const double frame_b = 116 + (photon_map ? 60 : 0);
    const double per_sample = 2 * (72 + 16 + 32 + 40 + (D->stream3 ? 32 + 8 : 0)) + 12 + frames_per_sample * frame_b;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 1u << 26;
    const double have = (double)D->cap_samples * (2 * (72 + 16 + 32 + 40 + (D->stream3 ? 32 + 8 : 0)) + 12) + (double)D->cap_frames * frame_b;
    for (uint32_t p = 1u << 28; p > (1u << 22); p >>= 1)
        if ((D->cap_samples >= p && (double)D->cap_frames >= (double)p * frames_per_sample) || (double)p * per_sample <= 0.85 * ((double)free_b + have)) return p;
    return 1u << 22;
}

static Frames MakeFrames(DeviceState *D)
{
    Frames F;
    const size_t c = D->cap_frames;
    F.parent = D->d_fu; F.info = D->d_fu + c; F.info2 = D->d_fu + 2 * c; F.skey = D->d_fu + 3 * c;
    F.code = D->d_fcode;
    float *p = D->d_ff;
    F.mult = p; p += 3 * c; F.refr = p; p += 3 * c; F.gi = p; p += 3 * c; F.gi_mult = p; p += 3 * c;
    F.brdf = p; p += 3 * c; F.refr_color = p; p += 3 * c; F.caustic = p; p += 3 * c;
    F.rr = p; p += c; F.vis = p;
    float *q = D->d_ph_frames;
    F.ph_p = q; F.ph_n = q ? q + 3 * c : nullptr; F.ph_v = q ? q + 6 * c : nullptr; F.ph_kd = q ? q + 9 * c : nullptr; F.ph_ks = q ? q + 12 * c : nullptr;
    return F;
}

// owned pixels [q0, q0+n) that lie inside the image (edge tiles stick out); same mapping as pixel_of()
static uint64_t CountValidPixels(const PassInfo &P, uint32_t n)
{
    const uint64_t tp = (uint64_t)P.tile * P.tile;
    uint64_t valid = 0, q = P.q0, end = (uint64_t)P.q0 + n;
    while (q < end) {
        const uint64_t k = q / tp, t_begin = k * tp, t_end = t_begin + tp;
        const uint64_t tid = (uint64_t)P.rank + k * (uint64_t)P.world;
        const uint64_t ty = tid / (uint64_t)P.tiles_x, tx = tid % (uint64_t)P.tiles_x;
        const uint64_t a = q, b = std::min<uint64_t>(end, t_end);
        if (ty < (uint64_t)P.tiles_y) {
            const int64_t w = std::min<int64_t>(P.tile, (int64_t)P.W - (int64_t)tx * P.tile), h = std::min<int64_t>(P.tile, (int64_t)P.H - (int64_t)ty * P.tile);
            if (a == t_begin && b == t_end) valid += (uint64_t)(std::max<int64_t>(w, 0) * std::max<int64_t>(h, 0));
            else
                for (uint64_t x = a; x < b; x++) {
                    const uint64_t within = x - t_begin;
                    if ((int64_t)(within % P.tile) < w && (int64_t)(within / P.tile) < h) valid++;
                }
        }
        q = b;
    }
    return valid;
}

// Kernel timing with HIP events on the library's stream, without a host round trip per kernel: start/stop events
// come from a pool and are only read back (FlushTimers) after a stream synchronisation the pipeline needs anyway.
// bhrt_opts.timers: 0 = the shading kernel only (the dominant one; bench.py's roofline), 1 = every kernel group, -1 = none.
// An event between two kernels costs ~6 us of idle GPU, which is why the default does not time everything.
struct Timer {
    DeviceState *D;
    double *acc;
    int e0;
    bool on;
    static hipEvent_t Get(DeviceState *d, int &idx)
    {
        if (d->ev_free.empty()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); d->ev_pool.push_back(e); d->ev_free.push_back((int)d->ev_pool.size() - 1); }
        idx = d->ev_free.back();
        d->ev_free.pop_back();
        return d->ev_pool[idx];
    }
    hipStream_t s;
    Timer(DeviceState *d, double *a, int level = 1, hipStream_t on_stream = nullptr) : D(d), acc(a), e0(-1), on(d->timers >= level), s(on_stream ? on_stream : d->stream) { if (on) (void)hipEventRecord(Get(D, e0), s); }
    void Stop()
    {
        if (!on) return;
        int e1;
        (void)hipEventRecord(Get(D, e1), s);
        D->ev_pending.push_back({e0, e1, acc});
    }
};
// Host side of publish_counters (k_shade's last workgroup): spins until the step's counters have arrived.  Leaves the loop when the stream reports an error
// or has drained without the flag appearing.
static int WaitPublished(DeviceState *D, uint32_t seq)
{
    auto arrived = [&]() { return __atomic_load_n(&D->h_pub->seq, __ATOMIC_ACQUIRE) == seq; };
    for (uint64_t spin = 1; !arrived(); spin++) {
        __builtin_ia32_pause();
        if ((spin & 0xffff) == 0) {
            const hipError_t e = hipStreamQuery(D->stream);
            if (e == hipErrorNotReady) continue;
            if (e != hipSuccess) { SetError(std::string("stream error while waiting for the step counters: ") + hipGetErrorString(e)); return BHRT_ERR_HIP; }
            if (!arrived()) { SetError("stream drained without publishing the step counters"); return BHRT_ERR_HIP; }
        }
    }
    return BHRT_OK;
}

// Reads back the timers whose kernels have finished; the others (kernels launched ahead of the host, e.g. the shadow
// trace of the current step) stay pending until a later call.  final = true: after a full stream synchronisation.
static void FlushTimers(DeviceState *D, bool final = false)
{
    size_t keep = 0;
    for (size_t k = 0; k < D->ev_pending.size(); k++) {
        const auto p = D->ev_pending[k];
        if (!final && hipEventQuery(D->ev_pool[p.e1]) != hipSuccess) { D->ev_pending[keep++] = p; continue; }
        float ms = 0;
        if (p.acc && hipEventElapsedTime(&ms, D->ev_pool[p.e0], D->ev_pool[p.e1]) == hipSuccess) *p.acc += ms * 1e-3;
        D->ev_free.push_back(p.e0);
        D->ev_free.push_back(p.e1);
    }
    D->ev_pending.resize(keep);
}

static int EnsurePhotonScratch(DeviceState *D, uint32_t lanes)
{
    if (D->scr_lanes >= lanes) return BHRT_OK;
    if (D->d_scr) (void)hipFree(D->d_scr);
    D->d_scr = nullptr; D->scr_lanes = 0;
    HIP_CHECK(hipMalloc(&D->d_scr, (size_t)lanes * BHRT_HEAP_COLUMN * sizeof(unsigned long long)));
    D->scr_lanes = lanes;
    return BHRT_OK;
}

// Caustic gather of queries [q0, q0+cnt) of `sink`: pass 1 without candidate lists over all of them, pass 2 with the
// candidate heap for the few that met 1000 photons (in chunks of the scratch columns).
template <class Sink>
static int RunGather(DeviceState *D, const Sink &sink, uint32_t q0, uint32_t cnt, float radius, bhrt_stats *st)
{
    if (cnt == 0) return BHRT_OK;
    if (D->heavy_cap < cnt) {
        auto fr = [](uint32_t *&p) { if (p) (void)hipFree(p); p = nullptr; };
        fr(D->d_heavy); fr(D->d_long); fr(D->d_cell_of); fr(D->d_gorder); fr(D->d_rank_of); fr(D->d_keys_out);
        if (D->d_sort_temp) (void)hipFree(D->d_sort_temp);
        D->d_sort_temp = nullptr; D->sort_temp_bytes = 0;
        D->heavy_cap = 0;
        HIP_CHECK(hipMalloc(&D->d_heavy, (size_t)cnt * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_long, (size_t)cnt * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_cell_of, (size_t)cnt * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_gorder, (size_t)cnt * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_rank_of, (size_t)cnt * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_keys_out, (size_t)cnt * sizeof(uint32_t)));
        if (GatherSortPairs(nullptr, nullptr, nullptr, nullptr, cnt, nullptr, &D->sort_temp_bytes, 28, D->stream) != 0) { SetError("gather sort: temp size"); return BHRT_ERR_HIP; }
        HIP_CHECK(hipMalloc(&D->d_sort_temp, std::max<size_t>(D->sort_temp_bytes, 16)));
        D->heavy_cap = cnt;
    }
    if (!D->d_n_heavy) {
        HIP_CHECK(hipMalloc(&D->d_n_heavy, 12 * sizeof(uint32_t))); // [0] heavy, [1] long, [2..3] nodes visited (64-bit), [4] selection rounds, [5] compactions, [8..11] lane pass: found, answered
        HIP_CHECK(hipHostMalloc(&D->h_n_heavy, 12 * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_cells, ((size_t)BHRT_GATHER_CELLS + 1) * sizeof(uint32_t)));
        HIP_CHECK(hipMalloc(&D->d_tile_sums, (size_t)(BHRT_GATHER_CELLS / kScanTile + kScanBlock) * sizeof(uint32_t)));
    }
    dim3 grid((cnt + kBlock - 1) / kBlock);
    const dim3 block(kBlock);
    const uint32_t *order = nullptr;
    uint32_t n_walk = cnt; // queries of pass 1
    if (cnt >= (1u << 16)) { // small batches are latency-bound anyway
        GatherGrid G;
        for (int k = 0; k < 3; k++) {
            const float lo = D->pm.lo[k] - radius, hi = D->pm.hi[k] + radius;
            G.lo[k] = lo;
            G.inv_cell[k] = hi > lo ? (float)(1 << BHRT_GATHER_CELL_BITS) / (hi - lo) : 0.f;
        }
        if (D->knobs.gather_counting_sort) { // the counting sort of rounds 1-3 (BHRT_GATHER_COUNTING_SORT=1: A/B and second opinion)
            const uint32_t n_tiles = BHRT_GATHER_CELLS / kScanTile;
            HIP_CHECK(hipMemsetAsync(D->d_cells, 0, ((size_t)BHRT_GATHER_CELLS + 1) * sizeof(uint32_t), D->stream));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_cell_count<Sink>), grid, block, 0, D->stream, sink, q0, cnt, G, D->pm, radius, D->d_cell_of, D->d_rank_of, D->d_cells);
            // exclusive scan over the cells and one more entry, which ends up holding the number of queries that take part
            hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles + 1), dim3(kScanBlock), 0, D->stream, D->d_cells, BHRT_GATHER_CELLS + 1, D->d_tile_sums);
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, D->stream, D->d_tile_sums, n_tiles + 1);
            hipLaunchKernelGGL(k_scan_add, dim3(n_tiles + 1), dim3(kScanBlock), 0, D->stream, D->d_cells, BHRT_GATHER_CELLS + 1, D->d_tile_sums);
            hipLaunchKernelGGL(k_gather_cell_scatter, grid, block, 0, D->stream, q0, cnt, D->d_cell_of, D->d_rank_of, D->d_cells, D->d_gorder);
            HIP_CHECK(hipMemcpyAsync(D->h_n_heavy, D->d_cells + BHRT_GATHER_CELLS, sizeof(uint32_t), hipMemcpyDeviceToHost, D->stream));
        } else { // (cell, query) pairs through a stable radix sort: no atomics, the queries of a cell stay in index order
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_cell_key<Sink>), grid, block, 0, D->stream, sink, q0, cnt, G, D->pm, radius, D->d_cell_of, D->d_rank_of);
            size_t tb = D->sort_temp_bytes;
            if (GatherSortPairs(D->d_cell_of, D->d_keys_out, D->d_rank_of, D->d_gorder, cnt, D->d_sort_temp, &tb, 28, D->stream) != 0) { SetError("gather sort failed"); return BHRT_ERR_HIP; }
            hipLaunchKernelGGL(k_gather_first_out, dim3(1), dim3(1), 0, D->stream, D->d_keys_out, cnt, D->d_n_heavy + 7);
            HIP_CHECK(hipMemcpyAsync(D->h_n_heavy, D->d_n_heavy + 7, sizeof(uint32_t), hipMemcpyDeviceToHost, D->stream));
        }
        order = D->d_gorder;
        HIP_CHECK(hipStreamSynchronize(D->stream));
        n_walk = D->h_n_heavy[0];
        grid = dim3((n_walk + kBlock - 1) / kBlock);
    }
    HIP_CHECK(hipMemsetAsync(D->d_n_heavy, 0, 12 * sizeof(uint32_t), D->stream));
    int lane_budget = BHRT_GATHER_LANE_BUDGET;
    if (D->knobs.gather_lane_budget > 0) lane_budget = D->knobs.gather_lane_budget; // test knob: a tiny budget sends every query through pass 2
    // (Streaming the queries through resident waves, lanes refilled from a cursor as in k_trace_mesh_stream, is SLOWER here — 0.69 -> 0.81-1.01 s per
    // frame for refill thresholds of 60-16 lanes: the 64 queries of a wave come from one cell and walk the tree in step, so their loads hit
    // the same lines; refilled lanes are out of step with their neighbours and every load becomes a gather.  33 of 64 lanes busy is the cheaper evil.)
    if (n_walk > 0) {
        if (D->knobs.gather_stats)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_gather_fast<Sink, true>), grid, block, 0, D->stream, sink, q0, n_walk, order, D->pm, radius, lane_budget, D->d_heavy,
                               D->d_long, D->d_n_heavy);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_gather_fast<Sink>), grid, block, 0, D->stream, sink, q0, n_walk, order, D->pm, radius, lane_budget, D->d_heavy,
                               D->d_long, D->d_n_heavy);
    }
    HIP_CHECK(hipMemcpyAsync(D->h_n_heavy, D->d_n_heavy, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost, D->stream));
    HIP_CHECK(hipStreamSynchronize(D->stream));
    const uint32_t n_heavy = D->h_n_heavy[0], n_long = D->h_n_heavy[1];
    if (st) {
        st->photon_found += (uint64_t)D->h_n_heavy[8] | ((uint64_t)D->h_n_heavy[9] << 32); // knob "gather_stats" only
        st->photon_lane_queries += (uint64_t)n_walk - n_heavy - n_long;
        st->photon_lane_nodes += (uint64_t)D->h_n_heavy[2] | ((uint64_t)D->h_n_heavy[3] << 32);
        st->photon_queries += cnt;
        st->photon_wave_queries += n_long;
        st->photon_heavy_queries += n_heavy;
        st->photon_nodes_visited += (uint64_t)D->h_n_heavy[2] | ((uint64_t)D->h_n_heavy[3] << 32);
    }
    if (n_heavy + n_long == 0) return BHRT_OK;
    Timer t(D, st ? &st->seconds_photon_heavy : nullptr, 0);
    // pass 2, one wave per query: the long walks always; the heavy queries unless the exact replay is asked for.  Undecided ones go to a list of
    // their own (d_cell_of is free again after the cell sort)
    const uint32_t n_sel_heavy = D->photon_exact ? 0u : n_heavy;
    uint32_t *undecided = D->d_cell_of;
    uint32_t n_exact = 0;
    if (n_sel_heavy + n_long) {
        HIP_CHECK(hipMemsetAsync(D->d_n_heavy, 0, 8 * sizeof(uint32_t), D->stream));
        const uint32_t sel_waves = D->n_cus * BHRT_SEL_WAVES_PER_CU; // persistent one-wave workgroups, each with its scratch (candidates 24 KB + stack spill 48 KB)
        if (!D->d_sel) HIP_CHECK(hipMalloc(&D->d_sel, (size_t)sel_waves * BHRT_SEL_SCRATCH_WORDS * sizeof(uint32_t)));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_gather_select<Sink>), dim3(std::min<uint32_t>(n_sel_heavy + n_long, sel_waves)), dim3(64), 0, D->stream, sink, D->d_heavy,
                           n_sel_heavy, D->d_long, n_long, D->pm, radius, undecided, D->d_n_heavy, D->d_knn, D->d_sel, D->photon_exact);
        HIP_CHECK(hipMemcpyAsync(D->h_n_heavy, D->d_n_heavy, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, D->stream));
        HIP_CHECK(hipStreamSynchronize(D->stream));
        if (st) st->photon_nodes_visited += (uint64_t)D->h_n_heavy[2] | ((uint64_t)D->h_n_heavy[3] << 32);
        if (D->knobs.debug_gather)
            fprintf(stderr, "select pass: %u heavy + %u long queries, %llu nodes, %u rounds, %u compactions, %u undecided\n", n_sel_heavy, n_long,
                    (unsigned long long)((uint64_t)D->h_n_heavy[2] | ((uint64_t)D->h_n_heavy[3] << 32)), D->h_n_heavy[4], D->h_n_heavy[5], D->h_n_heavy[0]);
        n_exact = D->h_n_heavy[0];
    }
    // the reference's candidate-heap history replayed, one lane per query: what pass 2 left undecided, and (bhrt_opts.photon_exact) every heavy query
    for (int part = 0; part < 2; part++) {
        const uint32_t *list = part == 0 ? undecided : D->d_heavy;
        const uint32_t m_all = part == 0 ? n_exact : (D->photon_exact ? n_heavy : 0u);
        if (!m_all) continue;
        const uint32_t heap_lanes = 1u << 20; // as many heaps in flight as possible: the pass is a chain of dependent accesses per query (65 k lanes: 1.8x slower)
        int rc = EnsurePhotonScratch(D, std::min<uint32_t>(heap_lanes, (m_all + 4095u) & ~4095u));
        if (rc) return rc;
        HIP_CHECK(hipMemsetAsync(D->d_n_heavy + 2, 0, 2 * sizeof(uint32_t), D->stream));
        const uint32_t chunk = std::min<uint32_t>(D->scr_lanes, heap_lanes);
        for (uint32_t h0 = 0; h0 < m_all; h0 += chunk) {
            const uint32_t m = std::min<uint32_t>(chunk, m_all - h0);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_gather_heap<Sink>), dim3((m + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, sink, list, h0, m,
                               D->pm, radius, D->d_scr, (size_t)D->scr_lanes, D->d_n_heavy);
        }
        if (st) {
            st->photon_exact_queries += m_all;
            HIP_CHECK(hipMemcpyAsync(D->h_n_heavy + 2, D->d_n_heavy + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, D->stream));
            HIP_CHECK(hipStreamSynchronize(D->stream));
            st->photon_nodes_visited += (uint64_t)D->h_n_heavy[2] | ((uint64_t)D->h_n_heavy[3] << 32);
        }
    }
    t.Stop();
    return BHRT_OK;
}


// Renders owned pixels [q_begin, q_end) of this rank; samples_out (device) receives the per-sample buffer
// of the region when requested (parity tests).
static int RenderRange(bhrt_scene *scene, const bhrt_opts &o, uint8_t *d_rgb8, float *d_radiance, bhrt_stats *st, float *d_region_samples, int x0,
                       int y0, int x1, int y1)
{
    DeviceState *D = scene->dev;
    const bhrt_flat_header *H = scene->flat.hdr();
    const int W = H->camera.width, Hh = H->camera.height;
    const int tile = o.tile_size > 0 ? o.tile_size : 32;
    const int world = o.world_size > 0 ? o.world_size : 1;
    if (o.rank < 0 || o.rank >= world) { SetError("rank outside world_size"); return BHRT_ERR_ARG; }
    if (o.spp <= 0 || o.spp > 65535) { SetError("spp must be in 1..65535"); return BHRT_ERR_ARG; }
    if (o.internal_bounces < 0 || o.internal_bounces > 255 || o.gi_bounces < -1 || o.gi_bounces > 60) { SetError("bounce counts out of range"); return BHRT_ERR_ARG; }
    if (H->n_materials > 4095 || H->n_lights > 255) { SetError("too many materials/lights for the frame record"); return BHRT_ERR_UNSUPPORTED; }
    PassInfo P;
    P.W = W; P.H = Hh; P.tile = tile; P.tiles_x = (W + tile - 1) / tile; P.tiles_y = (Hh + tile - 1) / tile;
    P.rank = o.rank; P.world = world; P.spp = o.spp; P.seed = o.seed; P.jitter = o.jitter; P.gamma = o.gamma;
    P.by_spp = MakeFastDiv((uint32_t)o.spp); P.by_tile_px = MakeFastDiv((uint32_t)(tile * tile)); P.by_tiles_x = MakeFastDiv((uint32_t)P.tiles_x);
    P.by_tile = MakeFastDiv((uint32_t)tile);
    {
        const V3 ddx = v3(H->camera.dd_x[0], H->camera.dd_x[1], H->camera.dd_x[2]), ddy = v3(H->camera.dd_y[0], H->camera.dd_y[1], H->camera.dd_y[2]);
        const V3 ux = normalized(ddx), uy = normalized(ddy); // IEEE sqrt and divisions, no contraction: the bits the kernels used to compute per sample
        P.jx[0] = ux.x; P.jx[1] = ux.y; P.jx[2] = ux.z; P.jy[0] = uy.x; P.jy[1] = uy.y; P.jy[2] = uy.z;
        P.pixel_len = length(ddx);
    }
    const uint32_t n_tiles = (uint32_t)(P.tiles_x * P.tiles_y);
    const uint32_t owned_tiles = n_tiles > (uint32_t)o.rank ? (n_tiles - (uint32_t)o.rank + (uint32_t)world - 1) / (uint32_t)world : 0;
    const uint64_t owned_pixels = (uint64_t)owned_tiles * tile * tile;

    // Sized for 288 GB of HBM: few, large passes (every pass ends in a tail of nearly empty wavefront steps, and the big launches of a pass's
    // first steps run the denser the more rays they hold: ONE pass of 1.3e8 samples instead of two of 6.6e7 takes 7-9 % off the C3 and closed-room
    // frames).  ~1 KB of wavefront state per camera sample (1.4 KB with the photon-map frames) -> the default of 2^27 samples in flight takes
    // ~138 GB (186 GB); with less memory free the default halves until it fits.
    // Shade() frames per sample slot: six are provided for (an overflow halves the pass and redoes it).  A frame that does not fit into one pass
    // with six — C4's 2.7e8 samples per GPU — takes what the earlier passes of the same render (scene, options) have needed, + 30 %: C4 needs 1.1
    // frames per sample, and with 1.7 provided its 2^28 slots fit into 180 GB: one pass per frame instead of two.
    const uint64_t frames_key = ((uint64_t)(uint32_t)o.spp << 48) ^ ((uint64_t)(uint32_t)(o.gi_bounces + 1) << 40) ^ ((uint64_t)(uint32_t)o.internal_bounces << 32) ^
                                ((uint64_t)(uint32_t)world << 24) ^ ((uint64_t)(uint32_t)tile << 8) ^ (uint64_t)(o.photon_map ? 1 : 0);
    double frames_per_sample = 6.0;
    bool frames_learned = false;
    if (o.samples_per_pass <= 0 && D->frames_seen_key == frames_key && D->frames_seen > 0 && D->frames_seen < 4.0 &&
        owned_pixels * (uint64_t)o.spp > DefaultPassSamples(D, o.photon_map != 0, 6.0)) {
        frames_per_sample = std::min(6.0, D->frames_seen * 1.3 + 0.25);
        frames_learned = true;
    }
    uint32_t pass_samples = o.samples_per_pass > 0 ? (uint32_t)o.samples_per_pass : DefaultPassSamples(D, o.photon_map != 0, frames_per_sample);
    pass_samples = (uint32_t)std::min<uint64_t>(pass_samples, std::max<uint64_t>(owned_pixels * (uint64_t)o.spp, 1)); // never more than this render needs
    if (pass_samples < (uint32_t)o.spp) pass_samples = (uint32_t)o.spp;
    D->timers = o.timers;
    D->photon_exact = o.photon_exact;
    const bool ls = o.leaf_skip != 0; // the walks' instantiations with device_trace.h::leaf_skip compiled in
    int path_mode = 1; // the traversal's path in LDS: 1 = 16-bit pair indices, 2 = 32-bit (a mesh with 2^17 nodes or more), 0 = no (deeper than 32 levels)
    {
        const bhrt_mesh *hm = (const bhrt_mesh *)(scene->flat.blob.data() + H->off_meshes);
        for (uint32_t k = 0; k < H->n_meshes; k++) {
            if (hm[k].n_bvh_nodes > (1u << 17) && path_mode == 1) path_mode = 2;
            if (hm[k].bvh_depth > 32) path_mode = 0;
        }
    }
    // resident waves of the streaming mesh kernels (k_trace_mesh_stream): 6 per SIMD; BHRT_STREAM_WAVES=0 selects the launch-per-64-rays kernels
    const uint32_t stream_waves = D->knobs.stream_waves >= 0 ? (uint32_t)D->knobs.stream_waves : D->n_cus * 4u * (uint32_t)BHRT_STREAM_OCC;
    RenderParams R;
    R.internal_bounces = o.internal_bounces; R.gi_bounces = o.gi_bounces; R.photon = o.photon_map;
    auto wall0 = std::chrono::steady_clock::now();

    uint64_t q = 0;
    uint32_t pass_limit = 0; // samples actually put in flight per pass (<= buffer capacity)
    const uint64_t hint_key = ((uint64_t)(uint32_t)o.spp << 48) ^ ((uint64_t)(uint32_t)(o.gi_bounces + 1) << 40) ^ ((uint64_t)(uint32_t)o.internal_bounces << 32) ^
                              ((uint64_t)(uint32_t)world << 24) ^ ((uint64_t)(uint32_t)tile << 8) ^ (uint64_t)(o.photon_map ? 1 : 0) ^ ((uint64_t)pass_samples << 1);
    if (D->pass_hint_key == hint_key && D->pass_hint) pass_limit = D->pass_hint;
    while (q < owned_pixels) {
        int rc = EnsureWorkspace(D, pass_samples, frames_per_sample);
        if (rc) return rc;
        R.cap_rays = D->cap_rays; R.cap_shadow = D->cap_rays; R.cap_frames = D->cap_frames;
        if (D->knobs.frame_cap > 0) R.cap_frames = std::min<uint32_t>(R.cap_frames, (uint32_t)D->knobs.frame_cap); // test knob: a pass that overflows
        if (o.photon_map) {
            if (D->ph_frames_cap < D->cap_frames) {
                if (D->d_ph_frames) (void)hipFree(D->d_ph_frames);
                D->d_ph_frames = nullptr;
                HIP_CHECK(hipMalloc(&D->d_ph_frames, (size_t)D->cap_frames * 15 * sizeof(float)));
                D->ph_frames_cap = D->cap_frames;
            }
        }
        if (pass_limit == 0) pass_limit = pass_samples;
        if (pass_limit > D->cap_samples) pass_limit = D->cap_samples;
        const uint32_t px_per_pass = pass_limit / (uint32_t)o.spp;
        const uint32_t npx = (uint32_t)std::min<uint64_t>(px_per_pass, owned_pixels - q);
        P.q0 = (uint32_t)q; P.n_pixels = npx;
        RayQueue Q[2] = {MakeRayQueue(D->d_rayf[0], D->d_rayu[0], D->cap_rays), MakeRayQueue(D->d_rayf[1], D->d_rayu[1], D->cap_rays)};
        HitBuf HB; HB.t = D->d_hitf; HB.node = D->d_hiti; HB.prim = D->d_hiti + D->cap_rays; HB.front = D->d_hiti + 2 * (size_t)D->cap_rays;
        ShadowQueue SQ; { float *p = D->d_shf; const size_t c = D->cap_rays; SQ.ox = p; SQ.oy = p + c; SQ.oz = p + 2 * c; SQ.dx = p + 3 * c; SQ.dy = p + 4 * c; SQ.dz = p + 5 * c; SQ.tmax = p + 6 * c; SQ.frame = D->d_shu; }
        // Any-hit work beside the next step's closest-hit work (stream3): the shadow rays of step s are only needed by k_combine at the end of the pass, so their
        // kernels go to a stream of their own as soon as k_shade has written them, with a shadow queue per step parity, and fill what the pass's stream leaves
        // idle — above all the tail of every k_trace_mesh_stream launch (the longest walks of its last batch: 0.1-0.5 ms per wave step).
        const bool sh_overlap = D->knobs.shadow_overlap && D->stream3 != nullptr && D->d_shf2 != nullptr;
        ShadowQueue SQ2 = SQ; if (sh_overlap) { float *p = D->d_shf2; const size_t c = D->cap_rays; SQ2.ox = p; SQ2.oy = p + c; SQ2.oz = p + 2 * c; SQ2.dx = p + 3 * c; SQ2.dy = p + 4 * c; SQ2.dz = p + 5 * c; SQ2.tmax = p + 6 * c; SQ2.frame = D->d_shu2; }
        bool sh_pending[2] = {false, false};
        Frames F = MakeFrames(D);
        RayOrder RO = {D->d_order, D->order_shard_cap, D->d_seg, D->d_seg + 3 * BHRT_ORDER_SHARDS + 1, D->d_seg + 6 * BHRT_ORDER_SHARDS + 1, D->d_seg + 7 * BHRT_ORDER_SHARDS + 2,
                       D->d_park, D->d_park + D->cap_rays, D->d_park + 3 * (size_t)D->cap_rays, D->d_seg + 8 * BHRT_ORDER_SHARDS + 3, D->d_park + 2 * (size_t)D->cap_rays};
        RayOrder RO_sh = RO; // the parked list of the any-hit rays when they run beside the pass: only its RC_MESH part exists
        if (sh_overlap) { RO_sh.idx = D->d_order_sh - (size_t)RC_MESH * BHRT_ORDER_SHARDS * D->order_shard_cap; RO_sh.mesh_start = D->d_seg_sh; RO_sh.mesh_count = D->d_seg_sh + BHRT_ORDER_SHARDS + 1; }
        HIP_CHECK(hipMemsetAsync(D->d_cnt, 0, sizeof(Counters), D->stream));
        // d_samples needs no clearing: every slot of a valid pixel is written exactly once (k_shade: background of a camera
        // miss; k_combine: root frame), and k_resolve never reads the slots of edge-tile pixels outside the image
        const uint32_t total = npx * (uint32_t)o.spp;
        uint32_t n_cur = total; // first wave step: one slot per (pixel, sample); the kernels compute the camera rays themselves
        bool first_step = true;
        int cur = 0;
        std::vector<uint32_t> frame_marks = {0};
        bool overflow = false;
        uint64_t pass_closest = 0, pass_camera = 0, pass_shadow = 0, pass_deferred = 0;
        uint32_t pass_steps = 0; // ray counters of this pass: added to *st only when the pass completes (an overflowing pass is redone)
        // rays set aside (SlowQueue): collected while the pass runs, traced and shaded in wave steps of their own once the queue is empty
        SlowQueue slowq;
        slowq.q = MakeRayQueue(D->d_slowf, D->d_slowu, kSlowCap);
        slowq.cap = D->d_slowf ? kSlowCap : 0u;
        if (D->knobs.no_slow_queue) slowq.cap = 0;
        const SlowQueue no_slow = {slowq.q, 0u};
        HitBuf slow_hits; slow_hits.t = D->d_slowf ? D->d_slowf + (size_t)6 * kSlowCap : nullptr; slow_hits.node = (int32_t *)(D->d_slowu ? D->d_slowu + (size_t)3 * kSlowCap : nullptr);
        slow_hits.prim = slow_hits.node ? slow_hits.node + kSlowCap : nullptr; slow_hits.front = slow_hits.node ? slow_hits.node + 2 * (size_t)kSlowCap : nullptr;
        uint32_t slow_pending = 0, slow_traced = 0, slow_injected = 0; // set aside so far this pass / of those handed to k_trace_slow / of those moved into a wave step
        struct SlowBatch { uint32_t end; uint32_t step; hipEvent_t done; };
        std::vector<SlowBatch> slow_batches; // k_trace_slow launches of this pass, in order
        size_t slow_batch_next = 0;          // first batch not yet moved in
        // A batch set aside by step s rides along with step s + 2 when that step is a big one (its hits, ~5 ms of walking on the second stream, are long
        // there; the pass's stream waits for the batch's event on the device).  What is set aside late waits until the queue has run empty, as before.
#ifndef BHRT_INJECT_MIN_LOG2
#define BHRT_INJECT_MIN_LOG2 20 /* 18 / 20 / 22: C3 44.0 / 43.9 / 44.2 ms, closed room 487.6 / 486.4 / 491.1 ms */
#endif
        constexpr uint32_t kInjectMinRays = 1u << BHRT_INJECT_MIN_LOG2;
        bool injected = false; // this wave step shades the rays that were set aside: their hits are there, and they were counted in their own step
        if (D->stream2) HIP_CHECK(hipStreamSynchronize(D->stream2)); // nothing of an abandoned pass still reads the queue
        if (D->stream3) HIP_CHECK(hipStreamSynchronize(D->stream3));
        // the any-hit kernels of the step whose counters the host has just read, on stream3 behind `after` (an event of the pass's stream)
        int sh_wait_par = 0; uint32_t sh_wait_n = 0;
        auto launch_any_hit = [&](hipEvent_t after) -> int {
            if (!sh_wait_n) return BHRT_OK;
            const int par = sh_wait_par; const uint32_t n_sh = sh_wait_n;
            sh_wait_n = 0;
            const ShadowQueue &sq_s = par ? SQ2 : SQ;
            HIP_CHECK(hipStreamWaitEvent(D->stream3, after, 0));
            Timer t(D, &st->seconds_trace_shadow, 1, D->stream3);
            const dim3 hg((n_sh + kBlock - 1) / kBlock), hb(kBlock);
            if (H->n_meshes > 0) {
                hipLaunchKernelGGL(k_trace_shadow_park, hg, hb, 0, D->stream3, D->S, sq_s, n_sh, (const uint32_t *)nullptr, F.vis, RO_sh, D->d_cnt_sh);
                hipLaunchKernelGGL(k_mesh_prefix, dim3(1), dim3(64), 0, D->stream3, D->d_cnt_sh, RO_sh);
                hipLaunchKernelGGL(path_mode == 1 ? (ls ? k_shadow_mesh<1, true> : k_shadow_mesh<1>) : path_mode == 2 ? (ls ? k_shadow_mesh<2, true> : k_shadow_mesh<2>) : k_shadow_mesh<0>, dim3((hg.x + BHRT_ORDER_SHARDS) * (kBlock / kShadowBlock)), dim3(kShadowBlock), 0, D->stream3, D->S, sq_s, F.vis, RO_sh);
            } else hipLaunchKernelGGL(k_trace_shadow<false>, hg, hb, 0, D->stream3, D->S, sq_s, n_sh, (const uint32_t *)nullptr, F.vis);
            t.Stop();
            HIP_CHECK(hipEventRecord(D->ev_shadow[par], D->stream3));
            sh_pending[par] = true;
            return BHRT_OK;
        };
        while (n_cur > 0 || slow_pending > slow_injected) {
            injected = false;
            uint32_t n_extra = 0; // slow rays that ride along with this step: slots [n_cur, n_cur + n_extra), hits in place, filed before k_shade
            if (n_cur == 0) {
                const auto w0 = std::chrono::steady_clock::now();
                HIP_CHECK(hipStreamSynchronize(D->stream2)); // the slow rays' hits
                if (D->knobs.debug_slow)
                    fprintf(stderr, "slow rays: %u moved in after wave step %u, waited %.1f ms for their hits, pass time so far %.1f ms\n", slow_pending, pass_steps,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() * 1e3, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() * 1e3);
                n_cur = slow_pending - slow_injected;
                hipLaunchKernelGGL(k_inject_slow, dim3((n_cur + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, slowq.q, slow_hits, slow_injected, n_cur, Q[cur], HB, 0u);
                pass_deferred += n_cur;
                slow_injected = slow_pending;
                slow_batch_next = slow_batches.size();
                injected = true;
            } else if (!first_step && n_cur >= kInjectMinRays) {
                uint32_t upto = slow_injected;
                size_t b = slow_batch_next;
                while (b < slow_batches.size() && slow_batches[b].step + 2 <= pass_steps) { upto = slow_batches[b].end; b++; }
                if (upto > slow_injected && (uint64_t)n_cur + (upto - slow_injected) <= D->cap_rays) {
                    HIP_CHECK(hipStreamWaitEvent(D->stream, slow_batches[b - 1].done, 0)); // stream2 runs its launches in order: the last one's event covers them all
                    n_extra = upto - slow_injected;
                    hipLaunchKernelGGL(k_inject_slow, dim3((n_extra + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, slowq.q, slow_hits, slow_injected, n_extra, Q[cur], HB, n_cur);
                    if (D->knobs.debug_slow) fprintf(stderr, "slow rays: %u ride along with wave step %u (%u rays)\n", n_extra, pass_steps, n_cur);
                    pass_deferred += n_extra;
                    slow_injected = upto;
                    slow_batch_next = b;
                }
            }
            const SlowQueue &sq = slowq;
            // the camera step of a scene without meshes: k_shade traces its rays itself (shade_block's kFused); BHRT_FUSED_CAMERA=0: the two-kernel form
            const bool fused = first_step && H->n_meshes == 0 && D->knobs.fused_camera;
            if (fused) {
            } else if (injected) {
                Timer t(D, &st->seconds_trace_closest);
                hipLaunchKernelGGL(k_file_all, dim3((n_cur + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, Q[cur], HB, 0u, n_cur, RO, D->d_cnt);
                t.Stop();
            } else {
                Timer t(D, &st->seconds_trace_closest);
                const dim3 tg((n_cur + kBlock - 1) / kBlock), tb(kBlock);
                if (H->n_meshes > 0) { // park the mesh rays, then finish them in dense workgroups
                    const uint32_t n_buckets = 1u << BHRT_PARK_KEY_BITS, n_tiles = n_buckets / kScanTile;
                    if (first_step) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_closest<true, true>), tg, tb, 0, D->stream, D->S, P, Q[cur], n_cur, 0, HB, RO, D->d_cnt, sq);
                    else {
                        HIP_CHECK(hipMemsetAsync(RO.park_bucket, 0, n_buckets * sizeof(uint32_t), D->stream)); // the trace kernel counts the keys as it parks
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_closest<true, false>), tg, tb, 0, D->stream, D->S, P, Q[cur], n_cur, 0, HB, RO, D->d_cnt, sq);
                    }
                    hipLaunchKernelGGL(k_mesh_prefix, dim3(1), dim3(64), 0, D->stream, D->d_cnt, RO);
                    if (!first_step) { // counting sort of the parked rays by coherence key (the camera step keeps slot order)
                        const dim3 pg(std::min<uint32_t>(tg.x + BHRT_ORDER_SHARDS, 4096u));
                        hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kScanBlock), 0, D->stream, RO.park_bucket, n_buckets, RO.park_bucket + n_buckets);
                        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, D->stream, RO.park_bucket + n_buckets, n_tiles);
                        hipLaunchKernelGGL(k_scan_add, dim3(n_tiles), dim3(kScanBlock), 0, D->stream, RO.park_bucket, n_buckets, RO.park_bucket + n_buckets);
                        hipLaunchKernelGGL(k_park_scatter, pg, tb, 0, D->stream, RO);
                    }
                    auto mesh_kernel = first_step ? (path_mode == 1 ? (ls ? k_trace_mesh<true, 1, true> : k_trace_mesh<true, 1>) : path_mode == 2 ? (ls ? k_trace_mesh<true, 2, true> : k_trace_mesh<true, 2>) : k_trace_mesh<true, 0>)
                                                  : (path_mode == 1 ? k_trace_mesh<false, 1> : path_mode == 2 ? k_trace_mesh<false, 2> : k_trace_mesh<false, 0>);
                    if (sh_wait_n) { // the last step's any-hit kernels start with this step's mesh walk: they get the SIMDs its finished waves leave
                        HIP_CHECK(hipEventRecord(D->ev_shade, D->stream));
                        rc = launch_any_hit(D->ev_shade); if (rc) return rc;
                    }
                    if (!first_step && path_mode != 0 && stream_waves > 0)
                    {
#ifdef BHRT_DEBUG_DRAIN
                        { unsigned long long t0[4] = {~0ull, ~0ull, 0, 0}; HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stream_t), t0, sizeof(t0))); }
#endif
                        hipLaunchKernelGGL(path_mode == 1 ? (ls ? k_trace_mesh_stream<1, true> : k_trace_mesh_stream<1>) : (ls ? k_trace_mesh_stream<2, true> : k_trace_mesh_stream<2>), dim3(std::min<uint32_t>((n_cur + 63) / 64, stream_waves)), dim3(64), 0, D->stream, D->S,
                                           Q[cur], HB, RO, D->d_cnt);
#ifdef BHRT_DEBUG_DRAIN
                        unsigned long long t1[4];
                        HIP_CHECK(hipMemcpyFromSymbol(t1, HIP_SYMBOL(g_stream_t), sizeof(t1)));
                        static double dbg_total = 0, dbg_drain = 0;
                        if (t1[1] != ~0ull) { dbg_total += (double)(t1[2] - t1[0]) / 1e5; dbg_drain += (double)(t1[2] - t1[1]) / 1e5; }
                        if (D->knobs.debug_drain) fprintf(stderr, "mesh launch: %u rays, %.3f ms, of which %.3f ms after the list ran out (sums %.1f / %.1f ms)\n", n_cur, (double)(t1[2] - t1[0]) / 1e5, t1[1] != ~0ull ? (double)(t1[2] - t1[1]) / 1e5 : 0.0, dbg_total, dbg_drain);
#endif
                    }
                    else
                    hipLaunchKernelGGL(mesh_kernel, first_step ? dim3(tg.x + BHRT_ORDER_SHARDS) /* shard segments padded to whole slices */ : dim3((n_cur + kMeshBlock - 1) / kMeshBlock),
                                       first_step ? tb : dim3(kMeshBlock), 0, D->stream, D->S, P, Q[cur], HB, RO, D->d_cnt);
                    if (!first_step) hipLaunchKernelGGL(k_file_parked, dim3(tg.x + BHRT_ORDER_SHARDS), tb, 0, D->stream, Q[cur], HB, RO, D->d_cnt);
                } else if (first_step) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_closest<false, true, false>), tg, tb, 0, D->stream, D->S, P, Q[cur], n_cur, 0, HB, RO, D->d_cnt, no_slow);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_closest<false, false, false>), tg, tb, 0, D->stream, D->S, P, Q[cur], n_cur, 0, HB, RO, D->d_cnt, no_slow);
                t.Stop();
            }
            if (n_extra) hipLaunchKernelGGL(k_file_all, dim3((n_extra + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, Q[cur], HB, n_cur, n_cur + n_extra, RO, D->d_cnt);
            if (!fused) {
                st->launches_trace_closest++;
                hipLaunchKernelGGL(k_order_prefix, dim3(1), dim3(128), 0, D->stream, D->d_cnt, RO);
            }
            const uint32_t seq = ++D->pub_seq;
            {
                Timer t(D, &st->seconds_shade, 0);
                const dim3 sg((n_cur + n_extra + kShadeBlock - 1) / kShadeBlock + (fused ? 0 : 3 * BHRT_ORDER_SHARDS)), sb(kShadeBlock);
                const bool tex = H->n_texmaps > 0;
                auto shade = fused ? (tex ? k_shade<true, true, true> : k_shade<true, false, true>)
                                   : first_step ? (tex ? k_shade<true, true> : k_shade<true, false>) : (tex ? k_shade<false, true> : k_shade<false, false>);
                const int par = (int)(pass_steps & 1u);
                if (sh_wait_n) { HIP_CHECK(hipEventRecord(D->ev_shade, D->stream)); rc = launch_any_hit(D->ev_shade); if (rc) return rc; }
                if (sh_overlap && sh_pending[par]) { HIP_CHECK(hipStreamWaitEvent(D->stream, D->ev_shadow[par], 0)); sh_pending[par] = false; } // the any-hit kernels of two steps ago still read this queue
                hipLaunchKernelGGL(shade, sg, sb, 0, D->stream, D->S, R, P, Q[cur], HB, n_cur + n_extra, Q[cur ^ 1], (sh_overlap && par) ? SQ2 : SQ, F, D->d_samples, D->d_cnt, RO, D->d_pub, seq); // + n_next, n_shadow, n_frames, overflow to the host
                t.Stop();
                if (sh_overlap) HIP_CHECK(hipEventRecord(D->ev_shade, D->stream));
            }
            if (!sh_overlap) { // the shadow trace of this step goes out before the host has the counters: its grid covers the upper bound
              // (<= 1 shadow ray per shaded ray, <= the queue's capacity) and the kernels read the length on the device
                Timer t(D, &st->seconds_trace_shadow);
                const uint32_t bound = std::min<uint32_t>(n_cur + n_extra, R.cap_shadow);
                const dim3 hg((bound + kBlock - 1) / kBlock), hb(kBlock);
                if (H->n_meshes > 0) {
                    hipLaunchKernelGGL(k_trace_shadow_park, hg, hb, 0, D->stream, D->S, SQ, bound, &D->d_cnt->n_shadow.v, F.vis, RO, D->d_cnt);
                    hipLaunchKernelGGL(k_mesh_prefix, dim3(1), dim3(64), 0, D->stream, D->d_cnt, RO);
                    // (streamed like k_trace_mesh_stream the any-hit walks gain nothing: they are short, C4 +4 ms, closed room -2 ms)
                    hipLaunchKernelGGL(path_mode == 1 ? (ls ? k_shadow_mesh<1, true> : k_shadow_mesh<1>) : path_mode == 2 ? (ls ? k_shadow_mesh<2, true> : k_shadow_mesh<2>) : k_shadow_mesh<0>, dim3((hg.x + BHRT_ORDER_SHARDS) * (kBlock / kShadowBlock)), dim3(kShadowBlock), 0, D->stream, D->S, SQ, F.vis, RO);
                } else hipLaunchKernelGGL(k_trace_shadow<false>, hg, hb, 0, D->stream, D->S, SQ, bound, &D->d_cnt->n_shadow.v, F.vis);
                t.Stop();
            }
            rc = WaitPublished(D, seq);
            if (rc) return rc;
            const HostCounters hc = *D->h_pub;
            FlushTimers(D);
            if (hc.overflow) { overflow = true; break; }
            if (first_step) { // camera step: dead rays of edge tiles are not rays
                const uint64_t valid_px = CountValidPixels(P, npx);
                pass_closest = valid_px * (uint64_t)o.spp;
                pass_camera = pass_closest;
                first_step = false;
            } else if (!injected) pass_closest += n_cur;
            slow_pending = std::min<uint32_t>(hc.n_slow, kSlowCap);
            if (slow_pending > slow_traced) { // the rays this step set aside: traced beside the pass
                const uint32_t cnt_new = slow_pending - slow_traced;
                hipLaunchKernelGGL(k_trace_slow, dim3(cnt_new), dim3(64), 0, D->stream2, D->S, slowq, slow_traced, slow_pending, slow_hits);
                if (D->knobs.debug_slow) fprintf(stderr, "slow rays: %u set aside in wave step %u at %.1f ms\n", cnt_new, pass_steps, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() * 1e3);
                slow_traced = slow_pending;
                const size_t bi = slow_batches.size();
                if (bi >= D->slow_events.size()) { hipEvent_t e = nullptr; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); D->slow_events.push_back(e); }
                HIP_CHECK(hipEventRecord(D->slow_events[bi], D->stream2));
                slow_batches.push_back({slow_pending, pass_steps, D->slow_events[bi]});
            }
            const uint32_t n_sh = hc.n_shadow;
            if (n_sh) { pass_shadow += n_sh; st->launches_trace_shadow++; }
            if (sh_overlap && n_sh) { // beside the next step, with the queue's exact length
                sh_wait_par = (int)(pass_steps & 1u); sh_wait_n = n_sh;
                // launched when the NEXT step's mesh walk is (below): behind an event of the pass's stream, so k_shade has ENDED and its rays are in memory for every XCD
                if (hc.n_next == 0) { rc = launch_any_hit(D->ev_shade); if (rc) return rc; } // no next step
            }
            frame_marks.push_back(hc.n_frames);
            n_cur = hc.n_next;
            cur ^= 1;
            pass_steps++;
        }
        if (sh_overlap) for (int k = 0; k < 2; k++) if (sh_pending[k]) HIP_CHECK(hipStreamWaitEvent(D->stream, D->ev_shadow[k], 0)); // every visibility is in its frame before the frames are read
        if (overflow) {
            // a capacity was exceeded: redo this pass with half the pixels in the same buffers
            // (results do not depend on the pass size: every sample has its own RNG key)
            if (pass_limit <= (uint32_t)o.spp) { SetError("wavefront buffers overflow even with one pixel per pass"); return BHRT_ERR_OVERFLOW; }
            pass_limit = std::max<uint32_t>((uint32_t)o.spp, pass_limit / 2);
            D->pass_hint_key = hint_key; D->pass_hint = pass_limit;
            if (frames_learned) D->frames_seen = 6.0; // the smaller frame pool was not enough after all: the next render of this kind provides six again
            continue;
        }
#ifdef BHRT_DEBUG_STREAM
        {
            unsigned long long d[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            HIP_CHECK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_stream_dbg), sizeof(d)));
            HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stream_dbg), z, sizeof(z)));
            unsigned long long w[16], wz[16] = {};
            HIP_CHECK(hipMemcpyFromSymbol(w, HIP_SYMBOL(g_walk_dbg), sizeof(w)));
            HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_walk_dbg), wz, sizeof(wz)));
            fprintf(stderr, "walk: descend bodies %llu (%.1f lanes), exact boxes %llu (%.1f lanes), climbs %llu (%.1f lanes), climbs after a hit %llu (%.1f lanes), triangle slots %llu (%.1f lanes), barycentric parts %llu (%.1f lanes), exact grazing quotients %llu; exact boxes %llu; box-missed leaf siblings %llu, of them left out %llu\n",
                    w[0], w[0] ? (double)w[1] / w[0] : 0., w[2], w[2] ? (double)w[3] / w[2] : 0., w[4], w[4] ? (double)w[5] / w[4] : 0., w[6], w[6] ? (double)w[7] / w[6] : 0., w[8],
                    w[8] ? (double)w[9] / w[8] : 0., w[10], w[10] ? (double)w[11] / w[10] : 0., w[12], w[13], w[14], w[15]);
            fprintf(stderr, "stream: %llu rounds (%llu descend, %llu leaf), walking %.3f, in phase %.3f, clocks in rounds %.3g of which descend rounds %.3g, %llu refills\n", d[0], d[6],
                    d[7], d[0] ? (double)d[1] / (64.0 * d[0]) : 0.0, d[0] ? (double)d[2] / (64.0 * d[0]) : 0.0, (double)d[3], (double)d[4], d[5]);
        }
#endif
        st->camera_samples += pass_camera; st->shadow_rays += pass_shadow; st->wave_iterations += pass_steps; st->deferred_rays += pass_deferred;
        { // what this pass needed of its frame pool, per sample slot
            const double seen = (double)frame_marks.back() / (double)std::max<uint64_t>((uint64_t)npx * (uint64_t)o.spp, 1);
            if (D->frames_seen_key != frames_key) { D->frames_seen_key = frames_key; D->frames_seen = 0; }
            D->frames_seen = std::max(D->frames_seen, seen);
        }
        if (o.photon_map && frame_marks.back() > 0) {
            // caustic term (MtlBlinn.cpp:329-342) of every frame of the pass in ONE gather: a gather launch lasts as long as its
            // longest query (15-80 ms for a query that fills the 1000-candidate heap), so a gather per wave step — 23 steps
            // per pass, most with a few hundred frames — spent 2 s per frame waiting for single lanes
            Timer t(D, &st->seconds_photon_gather, 0);
            const GatherToFrames sink = {F, D->S.materials};
            int rc = RunGather(D, sink, 0, frame_marks.back(), o.photon_radius > 0.f ? o.photon_radius : 0.5f /* MAX_Area, MtlBlinn.cpp:29 */, st);
            if (rc) return rc;
            t.Stop();
        }
        st->shade_calls += D->h_pub->n_frames;
        st->closest_rays += pass_closest;
        {
            Timer t(D, &st->seconds_other);
            for (size_t k = frame_marks.size(); k-- > 1;) {
                const uint32_t f0 = frame_marks[k - 1], f1 = frame_marks[k];
                if (f1 > f0) hipLaunchKernelGGL(k_combine, dim3((f1 - f0 + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, D->S, P, F, f0, f1, D->d_samples, o.photon_map);
            }
            hipLaunchKernelGGL(k_resolve, dim3((npx + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, P, D->d_samples, d_radiance, d_rgb8);
            if (d_region_samples)
                hipLaunchKernelGGL(k_copy_samples, dim3((npx + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, P, D->d_samples, x0, y0, x1, y1, d_region_samples);
            t.Stop();
        }
        HIP_CHECK(hipGetLastError());
        st->passes++;
        q += npx;
    }
    HIP_CHECK(hipStreamSynchronize(D->stream));
    FlushTimers(D, true);
    st->seconds_total += std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    return BHRT_OK;
}

} // namespace bhrt

using namespace bhrt;

extern "C" {

int bhrt_device_count(int *n)
try {
    if (!n) { SetError("null argument"); return BHRT_ERR_ARG; }
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; SetError(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); return BHRT_ERR_NO_DEVICE; }
    *n = c;
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

// Knobs: "shadow_overlap" — 0: the any-hit kernels of a wave step run on the pass's own stream, in front of the next step (bench.py times the kernel
// groups alone that way; same results either way).  Test knobs (tests/): "frame_cap" — a frame pool of that many Shade() frames, so that a pass overflows and is redone in halves; "gather_lane_budget" —
// photons a lane of the gather's first pass may visit before its query is handed to a whole wave.  0 switches a knob off.  Neither changes a result,
// and no environment variable sets them: a stray variable in a user's environment cannot send a render through the retry path.
int bhrt_scene_knob(bhrt_scene *scene, const char *name, int value)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!name || value < 0) { SetError("knob: bad arguments"); return BHRT_ERR_ARG; }
    if (!strcmp(name, "frame_cap")) scene->dev->knobs.frame_cap = value;
    else if (!strcmp(name, "gather_lane_budget")) scene->dev->knobs.gather_lane_budget = value;
    else if (!strcmp(name, "gather_stats")) scene->dev->knobs.gather_stats = value;
    else if (!strcmp(name, "shadow_overlap")) scene->dev->knobs.shadow_overlap = value != 0;
    else { SetError(std::string("knob: unknown name ") + name); return BHRT_ERR_ARG; }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_scene_upload(bhrt_scene *scene, int device)
try {
    if (!scene) { SetError("null scene"); return BHRT_ERR_ARG; }
    if (scene->dev && scene->dev->device == device) { HIP_CHECK(hipSetDevice(device)); return BHRT_OK; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { SetError("no HIP device available (this path has no CPU fallback)"); return BHRT_ERR_NO_DEVICE; }
    if (device < 0 || device >= count) { SetError("device index out of range"); return BHRT_ERR_ARG; }
    if (scene->max_bvh_depth > 64) { SetError("BVH deeper than 64 levels: the stackless trail holds 64"); return BHRT_ERR_UNSUPPORTED; }
    if (scene->dev) { DestroyDeviceState(scene->dev); scene->dev = nullptr; }
    HIP_CHECK(hipSetDevice(device));
    DeviceState *D = new DeviceState;
    D->knobs.FromEnv();
    D->device = device;
    scene->dev = D;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) D->n_cus = (uint32_t)cus;
    }
    const std::vector<uint8_t> &blob = scene->flat.blob;
    const bhrt_flat_header *H = scene->flat.hdr();
    HIP_CHECK(hipStreamCreate(&D->stream));
    HIP_CHECK(hipEventCreate(&D->ev[0]));
    HIP_CHECK(hipEventCreate(&D->ev[1]));
    HIP_CHECK(hipMalloc(&D->d_blob, blob.size()));
    HIP_CHECK(hipMemcpy(D->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    // ancestor chains (depth-1 ancestor first, the node itself last)
    std::vector<int32_t> chain((size_t)std::max<uint32_t>(H->n_nodes, 1) * BHRT_MAX_NODE_DEPTH, 0);
    const bhrt_node *nodes = (const bhrt_node *)(blob.data() + H->off_nodes);
    for (uint32_t n = 0; n < H->n_nodes; n++) {
        int k = nodes[n].depth - 1, c = (int)n;
        while (c >= 0 && k >= 0) { chain[(size_t)n * BHRT_MAX_NODE_DEPTH + k] = c; c = nodes[c].parent; k--; }
    }
    HIP_CHECK(hipMalloc(&D->d_chain, chain.size() * sizeof(int32_t)));
    HIP_CHECK(hipMemcpy(D->d_chain, chain.data(), chain.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&D->d_cnt, sizeof(Counters)));
    if (H->n_meshes > 0) { // SlowQueue: the rays parallel to a coordinate axis of the mesh they enter
        HIP_CHECK(hipMalloc(&D->d_slowf, (size_t)kSlowCap * 7 * sizeof(float)));
        HIP_CHECK(hipMalloc(&D->d_slowu, (size_t)kSlowCap * 6 * sizeof(uint32_t)));
        {
            int lo_prio = 0, hi_prio = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
            HIP_CHECK(hipStreamCreateWithPriority(&D->stream2, hipStreamNonBlocking, hi_prio));
            {
                HIP_CHECK(hipStreamCreateWithPriority(&D->stream3, hipStreamNonBlocking, hi_prio));
                HIP_CHECK(hipEventCreateWithFlags(&D->ev_shade, hipEventDisableTiming));
                for (int k = 0; k < 2; k++) HIP_CHECK(hipEventCreateWithFlags(&D->ev_shadow[k], hipEventDisableTiming));
                HIP_CHECK(hipMalloc(&D->d_seg_sh, (2 * BHRT_ORDER_SHARDS + 2) * sizeof(uint32_t)));
                HIP_CHECK(hipMalloc(&D->d_cnt_sh, sizeof(Counters)));
                HIP_CHECK(hipMemset(D->d_cnt_sh, 0, sizeof(Counters)));
            }
        }
    }
    HIP_CHECK(hipMalloc(&D->d_seg, (11 * BHRT_ORDER_SHARDS + 3) * sizeof(uint32_t)));
    HIP_CHECK(hipHostMalloc(&D->h_pub, sizeof(HostCounters), hipHostMallocMapped));
    memset(D->h_pub, 0, sizeof(HostCounters));
    HIP_CHECK(hipHostGetDevicePointer((void **)&D->d_pub, D->h_pub, 0));
    DevScene &S = D->S;
    S.blob = D->d_blob;
    S.nodes = (const bhrt_node *)(D->d_blob + H->off_nodes);
    S.meshes = (const bhrt_mesh *)(D->d_blob + H->off_meshes);
    S.materials = (const bhrt_material *)(D->d_blob + H->off_materials);
    S.lights = (const bhrt_light *)(D->d_blob + H->off_lights);
    S.texmaps = (const bhrt_texmap *)(D->d_blob + H->off_texmaps);
    S.textures = (const bhrt_texture *)(D->d_blob + H->off_textures);
    S.chain = D->d_chain;
    S.n_nodes = (int32_t)H->n_nodes; S.n_lights = (int32_t)H->n_lights;
    S.all_light_intensity = H->all_light_intensity;
    {
        const bhrt_material *mats = (const bhrt_material *)(scene->flat.blob.data() + H->off_materials);
        const bhrt_light *lts = (const bhrt_light *)(scene->flat.blob.data() + H->off_lights);
        const size_t n_cam = (size_t)std::max<uint32_t>(H->n_nodes, 1) * BHRT_MAX_NODE_DEPTH * 3;
        std::vector<float> aux(H->n_materials + H->n_lights + 1 + n_cam, 0.f);
        { // DevScene::cam_chain: the camera position through every node's chain, p' = itm * (p - pos) per level
            float *cc = aux.data() + H->n_materials + H->n_lights + 1;
            for (uint32_t n = 0; n < H->n_nodes; n++) {
                V3 p = v3(H->camera.pos[0], H->camera.pos[1], H->camera.pos[2]);
                for (int k = 0; k < nodes[n].depth; k++) {
                    const bhrt_xform &t = nodes[chain[(size_t)n * BHRT_MAX_NODE_DEPTH + k]].xf;
                    p = mat_mul(t.itm, p - v3(t.pos[0], t.pos[1], t.pos[2]));
                    float *o3 = cc + ((size_t)n * BHRT_MAX_NODE_DEPTH + k) * 3;
                    o3[0] = p.x; o3[1] = p.y; o3[2] = p.z;
                }
            }
        }
        for (uint32_t m = 0; m < H->n_materials; m++) {
            const float ior = mats[m].ior;
            const double r0d = (double)((1 - ior) / (1 + ior));
            aux[m] = (float)(r0d * r0d);
        }
        for (uint32_t l = 0; l < H->n_lights; l++) aux[H->n_materials + l] = ((lts[l].intensity[0] + lts[l].intensity[1] + lts[l].intensity[2]) / 3.0f) / H->all_light_intensity;
        HIP_CHECK(hipMalloc(&D->d_aux, aux.size() * sizeof(float)));
        HIP_CHECK(hipMemcpy(D->d_aux, aux.data(), aux.size() * sizeof(float), hipMemcpyHostToDevice));
        S.mat_r0 = D->d_aux;
        S.light_pick = D->d_aux + H->n_materials;
        S.cam_chain = D->d_aux + H->n_materials + H->n_lights + 1;
    }
    S.cam = H->camera; S.background = H->background; S.environment = H->environment;
    S.tapx[0] = S.tapy[0] = 0;
    for (int i = 1; i < 32; i++) { // scene.h:322-329 with the deterministic sin/cos (host and device agree bit for bit)
        auto halton = [](int index, int base) { float r = 0, f = 1.0f / (float)base; for (int k = index; k > 0; k /= base) { r += f * (k % base); f /= (float)base; } return r; };
        float x = halton(i, 2), y = halton(i, 3);
        float r = sqrtf(x) * 0.5f;
        S.tapx[i] = r * dm::sinf_(y * (float)M_PI * 2);
        S.tapy[i] = r * dm::cosf_(y * (float)M_PI * 2);
        if (!(fabsf(S.tapx[i]) <= 0.5f && fabsf(S.tapy[i]) <= 0.5f)) { SetError("footprint tap outside the half-unit disc"); return BHRT_ERR_UNSUPPORTED; } // device_shade.h::checker_taps_in_one_cell relies on it
    }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

static int EnsureApiScratch(DeviceState *D, size_t n)
{
    if (D->api_cap >= n) return BHRT_OK;
    if (D->d_api_f) (void)hipFree(D->d_api_f);
    if (D->d_api_i) (void)hipFree(D->d_api_i);
    D->d_api_f = nullptr; D->d_api_i = nullptr; D->api_cap = 0;
    HIP_CHECK(hipMalloc(&D->d_api_f, n * 9 * sizeof(float)));
    HIP_CHECK(hipMalloc(&D->d_api_i, n * 3 * sizeof(int32_t)));
    D->api_cap = n;
    return BHRT_OK;
}

int bhrt_trace_closest_dev(bhrt_scene *scene, const float *d_rays_soa, int hit_side, size_t n, bhrt_hits d_out, void *stream)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!d_rays_soa || !d_out.t || !d_out.node || !d_out.prim || !d_out.front) { SetError("null buffer"); return BHRT_ERR_ARG; }
    if (hit_side < 1 || hit_side > 3) { SetError("hit_side must be 1, 2 or 3"); return BHRT_ERR_ARG; }
    if (n == 0) return BHRT_OK;
    if (n > 0x7fffffffu) { SetError("too many rays for one call"); return BHRT_ERR_ARG; }
    RayQueue q = MakeRayQueue(const_cast<float *>(d_rays_soa), nullptr, n);
    HitBuf h; h.t = d_out.t; h.node = d_out.node; h.prim = d_out.prim; h.front = d_out.front;
    hipStream_t s = stream ? (hipStream_t)stream : scene->dev->stream;
    RayOrder no_order = {nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    auto closest_kernel = scene->flat.hdr()->n_meshes > 0 ? k_trace_closest<false, false, true> : k_trace_closest<false, false, false>;
    hipLaunchKernelGGL(closest_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       scene->dev->S, PassInfo(), q, (uint32_t)n, hit_side, h, no_order, (Counters *)nullptr, SlowQueue{q, 0u});
    HIP_CHECK(hipGetLastError());
    if (!stream) HIP_CHECK(hipStreamSynchronize(s));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_trace_closest_host(bhrt_scene *scene, const float *rays_soa, int hit_side, size_t n, bhrt_hits out)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (n == 0) return BHRT_OK;
    if (!rays_soa || !out.t || !out.node || !out.prim || !out.front) { SetError("null buffer"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    rc = EnsureApiScratch(D, n);
    if (rc) return rc;
    HIP_CHECK(hipMemcpy(D->d_api_f, rays_soa, n * 6 * sizeof(float), hipMemcpyHostToDevice));
    bhrt_hits d; d.t = D->d_api_f + 6 * n; d.node = D->d_api_i; d.prim = D->d_api_i + n; d.front = D->d_api_i + 2 * n;
    rc = bhrt_trace_closest_dev(scene, D->d_api_f, hit_side, n, d, nullptr);
    if (rc) return rc;
    HIP_CHECK(hipMemcpy(out.t, d.t, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out.node, d.node, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out.prim, d.prim, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out.front, d.front, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_trace_shadow_dev(bhrt_scene *scene, const float *d_rays_soa, const float *d_tmax, size_t n, float *d_vis, void *stream)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!d_rays_soa || !d_tmax || !d_vis) { SetError("null buffer"); return BHRT_ERR_ARG; }
    if (n == 0) return BHRT_OK;
    if (n > 0x7fffffffu) { SetError("too many rays for one call"); return BHRT_ERR_ARG; }
    ShadowQueue q;
    float *f = const_cast<float *>(d_rays_soa);
    q.ox = f; q.oy = f + n; q.oz = f + 2 * n; q.dx = f + 3 * n; q.dy = f + 4 * n; q.dz = f + 5 * n;
    q.tmax = const_cast<float *>(d_tmax); q.frame = nullptr;
    hipStream_t s = stream ? (hipStream_t)stream : scene->dev->stream;
    auto shadow_kernel = scene->flat.hdr()->n_meshes > 0 ? k_trace_shadow<true> : k_trace_shadow<false>;
    hipLaunchKernelGGL(shadow_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, scene->dev->S, q,
                       (uint32_t)n, (const uint32_t *)nullptr, d_vis);
    HIP_CHECK(hipGetLastError());
    if (!stream) HIP_CHECK(hipStreamSynchronize(s));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_trace_shadow_host(bhrt_scene *scene, const float *rays_soa, const float *tmax, size_t n, float *vis)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (n == 0) return BHRT_OK;
    if (!rays_soa || !tmax || !vis) { SetError("null buffer"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    rc = EnsureApiScratch(D, n);
    if (rc) return rc;
    HIP_CHECK(hipMemcpy(D->d_api_f, rays_soa, n * 6 * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(D->d_api_f + 6 * n, tmax, n * sizeof(float), hipMemcpyHostToDevice));
    rc = bhrt_trace_shadow_dev(scene, D->d_api_f, D->d_api_f + 6 * n, n, D->d_api_f + 7 * n, nullptr);
    if (rc) return rc;
    HIP_CHECK(hipMemcpy(vis, D->d_api_f + 7 * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_render_dev(bhrt_scene *scene, const bhrt_opts *opts, uint8_t *d_rgb8, float *d_radiance, bhrt_stats *stats, void *stream)
try {
    (void)stream; // the render pipeline synchronises its own stream per wave step
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!opts) { SetError("null opts"); return BHRT_ERR_ARG; }
    if (opts->photon_map && !scene->dev->d_photons) { SetError("photon_map = 1 needs bhrt_photon_build first"); return BHRT_ERR_ARG; }
    bhrt_stats local;
    memset(&local, 0, sizeof local);
    rc = RenderRange(scene, *opts, d_rgb8, d_radiance, &local, nullptr, 0, 0, 0, 0);
    if (stats) *stats = local;
    return rc;
} catch (...) { return bhrt::AbiException(); }

int bhrt_render(bhrt_scene *scene, const bhrt_opts *opts, uint8_t *rgb8, float *radiance, bhrt_stats *stats)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!opts) { SetError("null opts"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    const bhrt_flat_header *H = scene->flat.hdr();
    const size_t npix = (size_t)H->camera.width * H->camera.height;
    // The frame lives in HBM between calls (no allocation per frame).  The copies run at the full PCIe rate when the caller's
    // buffers are pinned (bhrt_host_alloc); pageable buffers work at about half of it.
    if (D->frame_px < npix) {
        if (D->d_frame_rgb) (void)hipFree(D->d_frame_rgb);
        if (D->d_frame_rad) (void)hipFree(D->d_frame_rad);
        D->d_frame_rgb = nullptr; D->d_frame_rad = nullptr; D->frame_px = 0;
        HIP_CHECK(hipMalloc(&D->d_frame_rgb, npix * 3));
        HIP_CHECK(hipMalloc(&D->d_frame_rad, npix * 3 * sizeof(float)));
        D->frame_px = npix;
    }
    if (opts->world_size > 1) { // pixels of tiles owned by other ranks keep the caller's values: they have to be in the device copy first
        if (rgb8) HIP_CHECK(hipMemcpyAsync(D->d_frame_rgb, rgb8, npix * 3, hipMemcpyHostToDevice, D->stream));
        if (radiance) HIP_CHECK(hipMemcpyAsync(D->d_frame_rad, radiance, npix * 12, hipMemcpyHostToDevice, D->stream));
    }
    rc = bhrt_render_dev(scene, opts, rgb8 ? D->d_frame_rgb : nullptr, radiance ? D->d_frame_rad : nullptr, stats, nullptr);
    if (rc) return rc;
    if (rgb8) HIP_CHECK(hipMemcpyAsync(rgb8, D->d_frame_rgb, npix * 3, hipMemcpyDeviceToHost, D->stream));
    if (radiance) HIP_CHECK(hipMemcpyAsync(radiance, D->d_frame_rad, npix * 12, hipMemcpyDeviceToHost, D->stream));
    HIP_CHECK(hipStreamSynchronize(D->stream));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

// pinned host memory for frame buffers handed to bhrt_render (and anything else that crosses PCIe)
int bhrt_host_alloc(void **ptr, size_t bytes)
try {
    if (!ptr || bytes == 0) { SetError("bhrt_host_alloc: bad argument"); return BHRT_ERR_ARG; }
    *ptr = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { SetError("no HIP device available"); return BHRT_ERR_NO_DEVICE; }
    HIP_CHECK(hipHostMalloc(ptr, bytes));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }
void bhrt_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

int bhrt_render_samples(bhrt_scene *scene, const bhrt_opts *opts, int x0, int y0, int x1, int y1, float *samples, bhrt_stats *stats)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!opts || !samples) { SetError("null argument"); return BHRT_ERR_ARG; }
    const bhrt_flat_header *H = scene->flat.hdr();
    if (x0 < 0 || y0 < 0 || x1 > H->camera.width || y1 > H->camera.height || x0 >= x1 || y0 >= y1) { SetError("bad region"); return BHRT_ERR_ARG; }
    if (opts->photon_map && !scene->dev->d_photons) { SetError("photon_map = 1 needs bhrt_photon_build first"); return BHRT_ERR_ARG; }
    const size_t nfl = (size_t)(x1 - x0) * (y1 - y0) * opts->spp * 3;
    float *d_s = nullptr;
    HIP_CHECK(hipMalloc(&d_s, nfl * sizeof(float)));
    HIP_CHECK(hipMemset(d_s, 0, nfl * sizeof(float)));
    bhrt_stats local;
    memset(&local, 0, sizeof local);
    rc = RenderRange(scene, *opts, nullptr, nullptr, &local, d_s, x0, y0, x1, y1);
    if (rc == BHRT_OK) HIP_CHECK(hipMemcpy(samples, d_s, nfl * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_s);
    if (stats) *stats = local;
    return rc;
} catch (...) { return bhrt::AbiException(); }

// ---- images beside the colour image ---------------------------------------------------------------
int bhrt_first_hit_dev(bhrt_scene *scene, float *d_z, float *d_normal, float *d_albedo, void *stream)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    DeviceState *D = scene->dev;
    const bhrt_flat_header *H = scene->flat.hdr();
    const int W = H->camera.width, Hh = H->camera.height;
    if (!d_z && !d_normal && !d_albedo) return BHRT_OK;
    hipStream_t st = stream ? (hipStream_t)stream : D->stream;
    hipLaunchKernelGGL(k_first_hit, dim3(((uint32_t)(W * Hh) + kBlock - 1) / kBlock), dim3(kBlock), 0, st, D->S, W, Hh, d_z, d_normal, d_albedo);
    HIP_CHECK(hipGetLastError());
    if (!stream) HIP_CHECK(hipStreamSynchronize(st));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_first_hit(bhrt_scene *scene, float *z, float *normal, float *albedo)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    DeviceState *D = scene->dev;
    const bhrt_flat_header *H = scene->flat.hdr();
    const size_t n = (size_t)H->camera.width * H->camera.height;
    float *d = nullptr;
    HIP_CHECK(hipMalloc(&d, n * 7 * sizeof(float)));
    rc = bhrt_first_hit_dev(scene, z ? d : nullptr, normal ? d + n : nullptr, albedo ? d + 4 * n : nullptr, nullptr);
    hipError_t e = hipSuccess;
    if (!rc && z) e = hipMemcpy(z, d, n * sizeof(float), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && normal) e = hipMemcpy(normal, d + n, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && albedo) e = hipMemcpy(albedo, d + 4 * n, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    (void)D;
    if (rc) return rc;
    HIP_CHECK(e);
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_zbuffer_image_dev(bhrt_scene *scene, const float *d_z, size_t n, uint8_t *d_img, void *stream)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!d_z || !d_img || n == 0 || n > 0xffffffffull) { SetError("bad z-buffer arguments"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    rc = EnsureApiScratch(D, 16);
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : D->stream;
    hipLaunchKernelGGL(k_z_range, dim3(1), dim3(1024), 0, st, d_z, (uint32_t)n, D->d_api_f);
    hipLaunchKernelGGL(k_z_image, dim3(((uint32_t)n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, d_z, (uint32_t)n, D->d_api_f, d_img);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st)); // the range scratch is shared with the other API calls
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_color_image_dev(bhrt_scene *scene, const float *d_radiance, size_t n_pixels, int gamma, float *d_color, void *stream)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!d_radiance || !d_color || n_pixels * 3 > 0xffffffffull) { SetError("bad colour image arguments"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    hipStream_t st = stream ? (hipStream_t)stream : D->stream;
    if (n_pixels) hipLaunchKernelGGL(k_color_image, dim3(((uint32_t)(n_pixels * 3) + kBlock - 1) / kBlock), dim3(kBlock), 0, st, d_radiance, (uint32_t)(n_pixels * 3), gamma, d_color);
    HIP_CHECK(hipGetLastError());
    if (!stream) HIP_CHECK(hipStreamSynchronize(st));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

static bool TileArgsOk(int W, int H, int tile, int rank, int world)
{
    if (W <= 0 || H <= 0 || tile <= 0 || world <= 0 || rank < 0 || rank >= world) { SetError("bad tile partition"); return false; }
    return true;
}
size_t bhrt_tiles_block_bytes(int width, int height, int tile, int world)
{
    if (width <= 0 || height <= 0 || tile <= 0 || world <= 0) return 0;
    const size_t n_tiles = (size_t)((width + tile - 1) / tile) * (size_t)((height + tile - 1) / tile);
    const size_t per_rank = (n_tiles + (size_t)world - 1) / (size_t)world;
    return (per_rank * (size_t)tile * tile * 15 + 15) & ~(size_t)15; // blocks stay 16-byte aligned in the gathered buffer
}
static uint32_t TilesPixelsPerRank(int width, int height, int tile, int world)
{
    const size_t n_tiles = (size_t)((width + tile - 1) / tile) * (size_t)((height + tile - 1) / tile);
    return (uint32_t)(((n_tiles + (size_t)world - 1) / (size_t)world) * (size_t)tile * tile);
}
int bhrt_tiles_pack_dev(const uint8_t *d_rgb8, const float *d_radiance, int width, int height, int tile, int rank, int world, void *d_block, void *stream)
try {
    if (!TileArgsOk(width, height, tile, rank, world)) return BHRT_ERR_ARG;
    if (!d_rgb8 || !d_radiance || !d_block) { SetError("null buffer"); return BHRT_ERR_ARG; }
    const uint32_t px = TilesPixelsPerRank(width, height, tile, world);
    hipLaunchKernelGGL(k_tiles_pack, dim3((px + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, d_rgb8, d_radiance, width, height, tile,
                       (width + tile - 1) / tile, rank, world, px, (float *)d_block, (uint8_t *)d_block + (size_t)px * 12);
    HIP_CHECK(hipGetLastError());
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }
int bhrt_tiles_unpack_dev(const void *d_blocks, int width, int height, int tile, int world, uint8_t *d_rgb8, float *d_radiance, void *stream)
try {
    if (!TileArgsOk(width, height, tile, 0, world)) return BHRT_ERR_ARG;
    if (!d_rgb8 || !d_radiance || !d_blocks) { SetError("null buffer"); return BHRT_ERR_ARG; }
    const size_t bb = bhrt_tiles_block_bytes(width, height, tile, world);
    const uint32_t px = TilesPixelsPerRank(width, height, tile, world);
    const uint64_t total = (uint64_t)px * (uint64_t)world;
    hipLaunchKernelGGL(k_tiles_unpack, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, (const uint8_t *)d_blocks, bb, width,
                       height, tile, (width + tile - 1) / tile, world, px, d_rgb8, d_radiance);
    HIP_CHECK(hipGetLastError());
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_math_eval_dev(int fn, const float *a, const float *b, size_t n, float *out)
try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { SetError("no HIP device available"); return BHRT_ERR_NO_DEVICE; }
    if (!a || !out || fn < 0 || fn > 9) { SetError("bad argument"); return BHRT_ERR_ARG; }
    if (n == 0) return BHRT_OK;
    float *d = nullptr;
    HIP_CHECK(hipMalloc(&d, n * 3 * sizeof(float)));
    HIP_CHECK(hipMemcpy(d, a, n * sizeof(float), hipMemcpyHostToDevice));
    if (b) HIP_CHECK(hipMemcpy(d + n, b, n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, fn, d, b ? d + n : nullptr, (uint32_t)n, d + 2 * n);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

// PrepareForIrradianceEstimation (cyPhotonMap.h:236-258) on the device (device_photon_build.h): d_in = n + 1 records in emission order, slot 0
// zero, powers scaled; *d_out = a new buffer with the balanced map in heap order (the caller owns it).
static int BalanceOnDevice(DeviceState *D, const DPhoton *d_in, uint32_t n, DPhoton **d_out)
{
    struct Bufs {
        uint32_t *id = nullptr, *la = nullptr, *lb = nullptr, *count = nullptr; float *key = nullptr; PbSeg *seg[2] = {nullptr, nullptr}; DPhoton *out = nullptr;
        ~Bufs() { (void)hipFree(id); (void)hipFree(la); (void)hipFree(lb); (void)hipFree(count); (void)hipFree(key); (void)hipFree(seg[0]); (void)hipFree(seg[1]); (void)hipFree(out); }
    } b;
    const size_t n2 = (size_t)n + 2, max_segs = (size_t)n / 2 + 2;
    HIP_CHECK(hipMalloc(&b.id, n2 * 4)); HIP_CHECK(hipMalloc(&b.la, n2 * 4)); HIP_CHECK(hipMalloc(&b.lb, n2 * 4)); HIP_CHECK(hipMalloc(&b.key, n2 * 4));
    HIP_CHECK(hipMalloc(&b.count, 4)); HIP_CHECK(hipMalloc(&b.seg[0], max_segs * sizeof(PbSeg))); HIP_CHECK(hipMalloc(&b.seg[1], max_segs * sizeof(PbSeg)));
    HIP_CHECK(hipMalloc(&b.out, ((size_t)n + 1) * sizeof(DPhoton)));
    HIP_CHECK(hipMemsetAsync(b.out, 0, ((size_t)n + 1) * sizeof(DPhoton), D->stream));
    hipLaunchKernelGGL(k_pb_root, dim3(1), dim3(1024), 0, D->stream, d_in, n, b.seg[0], b.id);
    uint32_t count = 1;
    int cur = 0;
    for (int level = 0; count > 0; level++) {
        if (level > 64) { SetError("photon balance: more than 64 levels"); return BHRT_ERR_HIP; }
        HIP_CHECK(hipMemsetAsync(b.count, 0, 4, D->stream));
        const uint64_t avg = (uint64_t)n / count; // segments of a level have nearly equal sizes: the workgroup that suits them
        if (avg > 2048) hipLaunchKernelGGL(k_pb_level<1024>, dim3(count), dim3(1024), 0, D->stream, d_in, b.id, b.key, b.la, b.lb, b.seg[cur], count, b.seg[cur ^ 1], b.count, b.out);
        else if (avg > 64) hipLaunchKernelGGL(k_pb_level<256>, dim3(count), dim3(256), 0, D->stream, d_in, b.id, b.key, b.la, b.lb, b.seg[cur], count, b.seg[cur ^ 1], b.count, b.out);
        else hipLaunchKernelGGL(k_pb_level<64>, dim3(count), dim3(64), 0, D->stream, d_in, b.id, b.key, b.la, b.lb, b.seg[cur], count, b.seg[cur ^ 1], b.count, b.out);
        HIP_CHECK(hipMemcpyAsync(&count, b.count, 4, hipMemcpyDeviceToHost, D->stream));
        HIP_CHECK(hipStreamSynchronize(D->stream));
        if (count > max_segs) { SetError("photon balance: segment list overflow"); return BHRT_ERR_HIP; }
        cur ^= 1;
    }
    *d_out = b.out;
    b.out = nullptr;
    return BHRT_OK;
}

// A balanced map (heap order, slot 0 unused) that lies in HBM -> installed for the gather: the 24-byte records stay where they are (the device
// owns them from here on), the decoded hot / cold copy the gather walks (PhotonMapDev), the direction bounds of the top levels and the bounds of
// the positions are formed on the device.  The host copy (bhrt_photon_get / _export) is fetched when somebody asks for it.
static int InstallPhotonMapDev(DeviceState *D, DPhoton *d_balanced, uint32_t n)
{
    if (D->d_photons && D->d_photons != d_balanced) (void)hipFree(D->d_photons);
    D->d_photons = d_balanced;
    D->n_photons = n;
    D->h_photons.clear();
    if (D->d_ph_hot) (void)hipFree(D->d_ph_hot);
    if (D->d_ph_cold) (void)hipFree(D->d_ph_cold);
    D->d_ph_hot = nullptr; D->d_ph_cold = nullptr;
    HIP_CHECK(hipMalloc(&D->d_ph_hot, ((size_t)n + 1) * sizeof(float4)));
    HIP_CHECK(hipMalloc(&D->d_ph_cold, ((size_t)n + 1) * 2 * sizeof(float4)));
    hipLaunchKernelGGL(k_photon_expand, dim3((n + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, D->d_photons, n, D->d_ph_hot, D->d_ph_cold);
    D->pm.hot = D->d_ph_hot; D->pm.cold = D->d_ph_cold; D->pm.n = (int)n; D->pm.half = (int)n / 2 - 1;
    { // PhotonMapDev::dbox, bottom up, one heap level per launch
        struct Tmp { float *lo = nullptr, *hi = nullptr, *b6 = nullptr; ~Tmp() { (void)hipFree(lo); (void)hipFree(hi); (void)hipFree(b6); } } t;
        HIP_CHECK(hipMalloc(&t.lo, ((size_t)n + 1) * 3 * sizeof(float)));
        HIP_CHECK(hipMalloc(&t.hi, ((size_t)n + 1) * 3 * sizeof(float)));
        HIP_CHECK(hipMalloc(&t.b6, 6 * sizeof(float)));
        int top = 0;
        while ((2ull << top) <= (uint64_t)n) top++; // level of node n
        for (int L = top; L >= 0; L--) {
            const uint32_t first = 1u << L, last = (uint32_t)std::min<uint64_t>((2ull << L) - 1, n);
            hipLaunchKernelGGL(k_pb_dbox_level, dim3((last - first + 256) / 256), dim3(256), 0, D->stream, D->d_photons, n, D->pm.half, first, last, t.lo, t.hi);
        }
        const uint32_t nb = std::min<uint32_t>(n + 1, 1u << BHRT_DBOX_LEVELS);
        if (D->d_ph_dbox) (void)hipFree(D->d_ph_dbox);
        D->d_ph_dbox = nullptr;
        HIP_CHECK(hipMalloc(&D->d_ph_dbox, (size_t)nb * 2 * sizeof(float4)));
        hipLaunchKernelGGL(k_pb_dbox_pack, dim3((nb + 255) / 256), dim3(256), 0, D->stream, t.lo, t.hi, nb, D->d_ph_dbox);
        D->pm.dbox = D->d_ph_dbox; D->pm.n_dbox = (int)nb;
        hipLaunchKernelGGL(k_pb_bounds, dim3(1), dim3(1024), 0, D->stream, D->d_photons, n, t.b6);
        float b6[6];
        HIP_CHECK(hipMemcpyAsync(b6, t.b6, sizeof b6, hipMemcpyDeviceToHost, D->stream));
        HIP_CHECK(hipStreamSynchronize(D->stream));
        for (int k = 0; k < 3; k++) { D->pm.lo[k] = b6[k]; D->pm.hi[k] = b6[3 + k]; }
    }
    return BHRT_OK;
}
// the same for a map that lies on the host (a file: bhrt_photon_import)
static int InstallPhotonMap(DeviceState *D)
{
    const uint32_t n = (uint32_t)D->h_photons.size() - 1;
    DPhoton *d = nullptr;
    HIP_CHECK(hipMalloc(&d, ((size_t)n + 1) * sizeof(DPhoton)));
    const hipError_t e = hipMemcpy(d, D->h_photons.data(), ((size_t)n + 1) * sizeof(DPhoton), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); HIP_CHECK(e); }
    std::vector<HostPhoton> keep;
    keep.swap(D->h_photons);
    const int rc = InstallPhotonMapDev(D, d, n);
    D->h_photons.swap(keep); // the host copy is already there
    return rc;
}
// host copy of the installed map, fetched on demand (bhrt_photon_get / bhrt_photon_export)
static int EnsureHostPhotons(DeviceState *D)
{
    if (!D->h_photons.empty() || !D->d_photons || !D->n_photons) return BHRT_OK;
    D->h_photons.assign((size_t)D->n_photons + 1, HostPhoton());
    HIP_CHECK(hipMemcpy(D->h_photons.data(), D->d_photons, ((size_t)D->n_photons + 1) * sizeof(DPhoton), hipMemcpyDeviceToHost));
    memset(&D->h_photons[0], 0, sizeof(HostPhoton));
    return BHRT_OK;
}
// the balance of n + 1 emission-order records in HBM: on the device, or (BHRT_PHOTON_BALANCE_HOST=1: the tests' second opinion) by photon_host.cpp
static int BalanceRecords(DeviceState *D, const DPhoton *d_in, uint32_t n, DPhoton **d_out)
{
    if (!D->knobs.balance_host) return BalanceOnDevice(D, d_in, n, d_out);
    std::vector<HostPhoton> h((size_t)n + 1);
    HIP_CHECK(hipMemcpy(h.data(), d_in, ((size_t)n + 1) * sizeof(DPhoton), hipMemcpyDeviceToHost));
    memset(&h[0], 0, sizeof(HostPhoton));
    BalancePhotons(h);
    DPhoton *d = nullptr;
    HIP_CHECK(hipMalloc(&d, ((size_t)n + 1) * sizeof(DPhoton)));
    const hipError_t e = hipMemcpy(d, h.data(), ((size_t)n + 1) * sizeof(DPhoton), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); HIP_CHECK(e); }
    *d_out = d;
    return BHRT_OK;
}

// Emission + stable compaction + ScalePhotonPowers + balance, all in HBM: *d_balanced = the balanced records (n + 1, slot 0 unused; the caller
// owns the buffer).  Per batch of emissions only two words reach the host (the largest count of a path, the photons of the batch).
// global_map: BuildPhotonMap / TracePhotonRay / RandomPhotonBounce instead of the caustic variants.
static int BuildPhotons(bhrt_scene *scene, const bhrt_opts *opts, uint32_t max_photons, bool global_map, DPhoton **d_balanced, uint32_t *n_out)
{
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!opts || max_photons == 0 || max_photons > (1u << 28)) { SetError("bad photon budget"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    const bhrt_flat_header *H = scene->flat.hdr();
    const bhrt_light *lights = (const bhrt_light *)(scene->flat.blob.data() + H->off_lights);
    // BuildCausticPhotonMap, Main.cpp:346-361: point lights sorted by Gray * GetSize() (int size, lights.h:76)
    std::vector<int32_t> pl;
    for (uint32_t i = 0; i < H->n_lights; i++)
        if (lights[i].type == BHRT_LIGHT_POINT) pl.push_back((int32_t)i);
    if (pl.empty()) { SetError("photon map: the scene has no point light (BuildCausticPhotonMap returns false)"); return BHRT_ERR_UNSUPPORTED; }
    auto key = [&](int32_t i) { return ((lights[i].intensity[0] + lights[i].intensity[1] + lights[i].intensity[2]) / 3.0f) * (int)lights[i].size; };
    std::sort(pl.begin(), pl.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
    float sum = 0;
    for (int32_t i : pl) sum += key(i);

    struct Bufs { // freed on every return path, HIP_CHECK's included
        DPhoton *out = nullptr, *tmp = nullptr; uint32_t *counts = nullptr, *offsets = nullptr, *sums = nullptr, *stats = nullptr; int32_t *pl = nullptr;
        ~Bufs() { (void)hipFree(out); (void)hipFree(tmp); (void)hipFree(counts); (void)hipFree(offsets); (void)hipFree(sums); (void)hipFree(stats); (void)hipFree(pl); }
    } bufs;
    DPhoton *&d_out = bufs.out, *&d_tmp = bufs.tmp;
    uint32_t *&d_counts = bufs.counts, *&d_offsets = bufs.offsets;
    int32_t *&d_pl = bufs.pl;
    HIP_CHECK(hipMalloc(&d_out, ((size_t)max_photons + 1) * sizeof(DPhoton)));
    HIP_CHECK(hipMemset(d_out, 0, ((size_t)max_photons + 1) * sizeof(DPhoton)));
    const uint32_t E = global_map ? 1u << 16 : 1u << 20; // emissions per batch (nearly every emission of the global map stores photons)
    uint32_t cap = 8;            // photons one path may store before the batch is redone with more room
    HIP_CHECK(hipMalloc(&d_counts, E * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&d_offsets, E * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&d_pl, pl.size() * sizeof(int32_t)));
    HIP_CHECK(hipMemcpy(d_pl, pl.data(), pl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc(&d_tmp, (size_t)E * cap * sizeof(DPhoton)));
    const uint32_t n_tiles = (E + kScanTile - 1) / kScanTile;
    HIP_CHECK(hipMalloc(&bufs.sums, n_tiles * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&bufs.stats, 2 * sizeof(uint32_t)));
    uint64_t e0 = 0, stored = 0;
    const uint64_t emission_budget = (uint64_t)max_photons * 4096ull + (1ull << 24);
    while (stored < max_photons && e0 < emission_budget) {
        if (global_map) hipLaunchKernelGGL(k_photon_emit<true>, dim3(E / kBlock), dim3(kBlock), 0, D->stream, D->S, opts->seed, e0, E, d_pl, (int)pl.size(), sum, d_tmp, cap, d_counts);
        else hipLaunchKernelGGL(k_photon_emit<false>, dim3(E / kBlock), dim3(kBlock), 0, D->stream, D->S, opts->seed, e0, E, d_pl, (int)pl.size(), sum, d_tmp, cap, d_counts);
        // where every path's photons go: exclusive prefix of the counts, on the device; two words come back
        HIP_CHECK(hipMemcpyAsync(d_offsets, d_counts, E * sizeof(uint32_t), hipMemcpyDeviceToDevice, D->stream));
        hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kScanBlock), 0, D->stream, d_offsets, E, bufs.sums);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, D->stream, bufs.sums, n_tiles);
        hipLaunchKernelGGL(k_scan_add, dim3(n_tiles), dim3(kScanBlock), 0, D->stream, d_offsets, E, bufs.sums);
        HIP_CHECK(hipMemsetAsync(bufs.stats, 0, 2 * sizeof(uint32_t), D->stream));
        hipLaunchKernelGGL(k_pb_batch_stats, dim3(64), dim3(256), 0, D->stream, d_counts, d_offsets, E, bufs.stats);
        uint32_t stats[2];
        HIP_CHECK(hipMemcpyAsync(stats, bufs.stats, sizeof stats, hipMemcpyDeviceToHost, D->stream));
        HIP_CHECK(hipStreamSynchronize(D->stream));
        if (stats[0] > cap) { // a path stored more than `cap` photons: redo this batch with room for all of them
            (void)hipFree(d_tmp);
            d_tmp = nullptr;
            cap = stats[0];
            HIP_CHECK(hipMalloc(&d_tmp, (size_t)E * cap * sizeof(DPhoton)));
            continue;
        }
        hipLaunchKernelGGL(k_photon_compact, dim3(E / kBlock), dim3(kBlock), 0, D->stream, d_tmp, cap, d_counts, d_offsets, (uint32_t)std::min<uint64_t>(stored, max_photons), E, max_photons, d_out);
        stored += stats[1];
        e0 += E;
    }
    const uint32_t n = (uint32_t)std::min<uint64_t>(stored, max_photons);
    if (n == 0) { SetError("photon map: no photon reached a photon surface"); return BHRT_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_photon_scale, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, d_out, n, 1.f / (float)(int)n); // Main.cpp:380
    *n_out = n;
    return BalanceRecords(D, d_out, n, d_balanced); // PrepareForIrradianceEstimation (cyPhotonMap.h:236-258)
}

// ---- multi-GPU photon build: emission is keyed by the emission index, so ranks emit disjoint index ranges,
// exchange the records (one all-gather per batch, bhraytracer_amd/dist.py::photon_build_sharded) and every rank installs the
// same map: the first max_photons records in emission order, exactly what bhrt_photon_build keeps.
int bhrt_photon_emit_range(bhrt_scene *scene, const bhrt_opts *opts, int global_map, uint64_t e0, uint32_t count, void *photons_out, uint32_t capacity,
                           uint32_t *n_photons)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!opts || !photons_out || !n_photons || count == 0 || count % kBlock != 0 || count > (1u << 24)) { SetError("photon emit range: bad arguments (count must be a multiple of 256)"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    const bhrt_flat_header *H = scene->flat.hdr();
    const bhrt_light *lights = (const bhrt_light *)(scene->flat.blob.data() + H->off_lights);
    std::vector<int32_t> pl; // BuildCausticPhotonMap, Main.cpp:346-361 (same order as BuildPhotons)
    for (uint32_t i = 0; i < H->n_lights; i++)
        if (lights[i].type == BHRT_LIGHT_POINT) pl.push_back((int32_t)i);
    if (pl.empty()) { SetError("photon map: the scene has no point light (BuildCausticPhotonMap returns false)"); return BHRT_ERR_UNSUPPORTED; }
    auto key = [&](int32_t i) { return ((lights[i].intensity[0] + lights[i].intensity[1] + lights[i].intensity[2]) / 3.0f) * (int)lights[i].size; };
    std::sort(pl.begin(), pl.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
    float sum = 0;
    for (int32_t i : pl) sum += key(i);
    struct Bufs {
        DPhoton *tmp = nullptr, *out = nullptr; uint32_t *counts = nullptr, *offsets = nullptr; int32_t *pl = nullptr;
        ~Bufs() { (void)hipFree(tmp); (void)hipFree(out); (void)hipFree(counts); (void)hipFree(offsets); (void)hipFree(pl); }
    } b;
    HIP_CHECK(hipMalloc(&b.counts, count * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&b.offsets, count * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&b.pl, pl.size() * sizeof(int32_t)));
    HIP_CHECK(hipMemcpy(b.pl, pl.data(), pl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<uint32_t> counts(count), offsets(count);
    uint32_t cap = 8;
    uint64_t total = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        (void)hipFree(b.tmp); b.tmp = nullptr;
        HIP_CHECK(hipMalloc(&b.tmp, (size_t)count * cap * sizeof(DPhoton)));
        if (global_map) hipLaunchKernelGGL(k_photon_emit<true>, dim3(count / kBlock), dim3(kBlock), 0, D->stream, D->S, opts->seed, e0, count, b.pl, (int)pl.size(), sum, b.tmp, cap, b.counts);
        else hipLaunchKernelGGL(k_photon_emit<false>, dim3(count / kBlock), dim3(kBlock), 0, D->stream, D->S, opts->seed, e0, count, b.pl, (int)pl.size(), sum, b.tmp, cap, b.counts);
        HIP_CHECK(hipMemcpyAsync(counts.data(), b.counts, count * sizeof(uint32_t), hipMemcpyDeviceToHost, D->stream));
        HIP_CHECK(hipStreamSynchronize(D->stream));
        uint32_t maxc = 0;
        for (uint32_t i = 0; i < count; i++) maxc = std::max(maxc, counts[i]);
        if (maxc <= cap) break;
        cap = maxc; // a path stored more photons than there was room for: once more with room for all
    }
    for (uint32_t i = 0; i < count; i++) { offsets[i] = (uint32_t)std::min<uint64_t>(total, capacity); total += counts[i]; }
    *n_photons = (uint32_t)std::min<uint64_t>(total, 0xffffffffull);
    if (total > capacity) { SetError("photon emit range: photons_out too small"); return BHRT_ERR_ARG; }
    if (total == 0) return BHRT_OK;
    HIP_CHECK(hipMalloc(&b.out, ((size_t)total + 1) * sizeof(DPhoton)));
    HIP_CHECK(hipMemcpyAsync(b.offsets, offsets.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, D->stream));
    hipLaunchKernelGGL(k_photon_compact, dim3(count / kBlock), dim3(kBlock), 0, D->stream, b.tmp, cap, b.counts, b.offsets, 0u, count, (uint32_t)total, b.out);
    HIP_CHECK(hipMemcpyAsync(photons_out, b.out + 1, (size_t)total * sizeof(DPhoton), hipMemcpyDefault, D->stream)); // host or device destination
    HIP_CHECK(hipStreamSynchronize(D->stream));
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

// n records in emission order with unscaled power -> ScalePhotonPowers(1 / n) (Main.cpp:380), balance, install for the gather
int bhrt_photon_install(bhrt_scene *scene, const void *records, uint32_t n)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!records || n == 0 || n > (1u << 28)) { SetError("photon install: bad arguments"); return BHRT_ERR_ARG; }
    DeviceState *D = scene->dev;
    DPhoton *d = nullptr;
    HIP_CHECK(hipMalloc(&d, ((size_t)n + 1) * sizeof(DPhoton)));
    hipError_t e = hipMemset(d, 0, sizeof(DPhoton));
    if (e == hipSuccess) e = hipMemcpy(d + 1, records, (size_t)n * sizeof(DPhoton), hipMemcpyDefault); // host or device source
    if (e != hipSuccess) { (void)hipFree(d); HIP_CHECK(e); }
    hipLaunchKernelGGL(k_photon_scale, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, D->stream, d, n, 1.f / (float)(int)n);
    DPhoton *bal = nullptr;
    rc = BalanceRecords(D, d, n, &bal); // the records stay in HBM from the caller's buffer to the installed map
    (void)hipFree(d);
    if (rc) return rc;
    return InstallPhotonMapDev(D, bal, n);
} catch (...) { return bhrt::AbiException(); }

int bhrt_photon_build(bhrt_scene *scene, const bhrt_opts *opts, uint32_t max_photons, uint32_t *n_stored)
try {
    if (!scene) { SetError("null scene"); return BHRT_ERR_ARG; }
    DPhoton *bal = nullptr;
    uint32_t n = 0;
    int rc = BuildPhotons(scene, opts, max_photons, false, &bal, &n);
    if (rc) return rc;
    DeviceState *D = scene->dev;
    rc = InstallPhotonMapDev(D, bal, n);
    if (rc) return rc;
    if (n_stored) *n_stored = D->n_photons;
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_photon_build_global(bhrt_scene *scene, const bhrt_opts *opts, uint32_t max_photons, void *photons_out, uint32_t capacity, uint32_t *n_stored,
                             const char *dat_path)
try {
    if (!scene) { SetError("null scene"); return BHRT_ERR_ARG; }
    DPhoton *bal = nullptr;
    uint32_t n = 0;
    int rc = BuildPhotons(scene, opts, max_photons, true, &bal, &n);
    if (rc) return rc;
    std::vector<HostPhoton> balanced((size_t)n + 1);
    const hipError_t ce = hipMemcpy(balanced.data(), bal, ((size_t)n + 1) * sizeof(DPhoton), hipMemcpyDeviceToHost);
    (void)hipFree(bal);
    HIP_CHECK(ce);
    if (n_stored) *n_stored = n;
    if (photons_out) {
        if (capacity < n) { SetError("photon buffer too small"); return BHRT_ERR_ARG; }
        memcpy(photons_out, &balanced[1], (size_t)n * sizeof(HostPhoton));
    }
    if (dat_path) { // Resource/photonmap.dat, Main.cpp:292-294
        FILE *fp = fopen(dat_path, "wb");
        if (!fp) { SetError(std::string("cannot write ") + dat_path); return BHRT_ERR_IO; }
        const bool ok = fwrite(&balanced[1], sizeof(HostPhoton), n, fp) == n;
        fclose(fp);
        if (!ok) { SetError("short write"); return BHRT_ERR_IO; }
    }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }


int bhrt_photon_gather_host_ex(bhrt_scene *scene, const float *p, const float *nrm, size_t cnt, float radius, int photon_exact, float *irrad, float *dir,
                               uint32_t *knn, uint32_t *knn_count, float *d2max)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    DeviceState *D = scene->dev;
    D->timers = 0;
    D->photon_exact = photon_exact;
    if (!D->d_photons) { SetError("photon map: call bhrt_photon_build first"); return BHRT_ERR_ARG; }
    if (!p || !nrm || !irrad || !dir) { SetError("null buffer"); return BHRT_ERR_ARG; }
    if (cnt == 0) return BHRT_OK;
    const uint32_t chunk = (knn || knn_count || d2max) ? 1u << 14 : 1u << 20;
    struct Bufs { float *f = nullptr; uint32_t *knn = nullptr; DeviceState *D; ~Bufs() { (void)hipFree(f); (void)hipFree(knn); D->d_knn = nullptr; } } b;
    b.D = D;
    HIP_CHECK(hipMalloc(&b.f, (size_t)chunk * 12 * sizeof(float)));
    float *d_buf = b.f;
    const size_t kw = BHRT_PHOTON_K + 2;
    std::vector<uint32_t> h_knn;
    if (knn || knn_count || d2max) { HIP_CHECK(hipMalloc(&b.knn, (size_t)chunk * kw * sizeof(uint32_t))); h_knn.resize((size_t)chunk * kw); }
    for (size_t c0 = 0; c0 < cnt; c0 += chunk) {
        const uint32_t m = (uint32_t)std::min<size_t>(chunk, cnt - c0);
        HIP_CHECK(hipMemcpy(d_buf, p + c0 * 3, (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_buf + (size_t)chunk * 3, nrm + c0 * 3, (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice));
        if (b.knn) HIP_CHECK(hipMemset(b.knn, 0, (size_t)m * kw * sizeof(uint32_t)));
        D->d_knn = b.knn;
        const GatherToArrays sink = {d_buf, d_buf + (size_t)chunk * 3, d_buf + (size_t)chunk * 6, d_buf + (size_t)chunk * 9};
        rc = RunGather(D, sink, 0, m, radius, nullptr);
        D->d_knn = nullptr;
        if (rc) return rc;
        HIP_CHECK(hipStreamSynchronize(D->stream));
        HIP_CHECK(hipMemcpy(irrad + c0 * 3, d_buf + (size_t)chunk * 6, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(dir + c0 * 3, d_buf + (size_t)chunk * 9, (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (b.knn) {
            HIP_CHECK(hipMemcpy(h_knn.data(), b.knn, (size_t)m * kw * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t *row = &h_knn[(size_t)i * kw];
                if (knn_count) knn_count[c0 + i] = row[0];
                if (d2max) memcpy(&d2max[c0 + i], &row[1], 4);
                if (knn) memcpy(knn + (c0 + i) * BHRT_PHOTON_K, row + 2, BHRT_PHOTON_K * sizeof(uint32_t));
            }
        }
    }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_photon_gather_host(bhrt_scene *scene, const float *p, const float *nrm, size_t cnt, float radius, float *irrad, float *dir)
{
    return bhrt_photon_gather_host_ex(scene, p, nrm, cnt, radius, 0, irrad, dir, nullptr, nullptr, nullptr);
}

int bhrt_photon_get(const bhrt_scene *scene, void *photons_out, uint32_t capacity, uint32_t *n)
try {
    if (!scene || !scene->dev || !scene->dev->n_photons) { SetError("photon map: nothing built"); return BHRT_ERR_ARG; }
    const int hrc = EnsureHostPhotons(scene->dev);
    if (hrc) return hrc;
    const uint32_t have = scene->dev->n_photons;
    if (n) *n = have;
    if (photons_out) {
        if (capacity < have) { SetError("photon buffer too small"); return BHRT_ERR_ARG; }
        memcpy(photons_out, &scene->dev->h_photons[1], (size_t)have * sizeof(HostPhoton));
    }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_photon_export(const bhrt_scene *scene, const char *dat_path)
try {
    if (!scene || !scene->dev || !scene->dev->n_photons || !dat_path) { SetError("photon map: nothing to export"); return BHRT_ERR_ARG; }
    const int hrc = EnsureHostPhotons(scene->dev);
    if (hrc) return hrc;
    FILE *fp = fopen(dat_path, "wb"); // fwrite(GetPhotons(), sizeof(Photon), NumPhotons(), fp), Main.cpp:383-385
    if (!fp) { SetError(std::string("cannot write ") + dat_path); return BHRT_ERR_IO; }
    const size_t n = scene->dev->n_photons;
    const bool ok = fwrite(&scene->dev->h_photons[1], sizeof(HostPhoton), n, fp) == n;
    fclose(fp);
    if (!ok) { SetError("short write"); return BHRT_ERR_IO; }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_photon_import(bhrt_scene *scene, const char *dat_path, int rebalance)
try {
    int rc = EnsureUploaded(scene);
    if (rc) return rc;
    if (!dat_path) { SetError("null path"); return BHRT_ERR_ARG; }
    FILE *fp = fopen(dat_path, "rb");
    if (!fp) { SetError(std::string("cannot open ") + dat_path); return BHRT_ERR_IO; }
    fseek(fp, 0, SEEK_END);
    const long bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (bytes <= 0 || bytes % (long)sizeof(HostPhoton) != 0 || (size_t)bytes / sizeof(HostPhoton) > (1u << 28)) {
        fclose(fp);
        SetError("photon file: size is not a positive multiple of the 24-byte record");
        return BHRT_ERR_IO;
    }
    const size_t n = (size_t)bytes / sizeof(HostPhoton);
    DeviceState *D = scene->dev;
    D->h_photons.assign(n + 1, HostPhoton());
    const bool ok = fread(&D->h_photons[1], sizeof(HostPhoton), n, fp) == n;
    fclose(fp);
    if (!ok) { D->h_photons.clear(); SetError("photon file: short read"); return BHRT_ERR_IO; }
    memset(&D->h_photons[0], 0, sizeof(HostPhoton));
    if (rebalance) { // InitializePhotonMapByFile runs PrepareForIrradianceEstimation again (cyPhotonMap.h:409-417): on the device like the build's
        DPhoton *d = nullptr, *bal = nullptr;
        HIP_CHECK(hipMalloc(&d, (n + 1) * sizeof(DPhoton)));
        hipError_t e = hipMemcpy(d, D->h_photons.data(), (n + 1) * sizeof(DPhoton), hipMemcpyHostToDevice);
        if (e == hipSuccess) rc = BalanceRecords(D, d, (uint32_t)n, &bal);
        (void)hipFree(d);
        HIP_CHECK(e);
        if (rc) return rc;
        return InstallPhotonMapDev(D, bal, (uint32_t)n); // the host copy is fetched again when somebody asks for it
    }
    return InstallPhotonMap(D);
} catch (...) { return bhrt::AbiException(); }

} // extern "C"
