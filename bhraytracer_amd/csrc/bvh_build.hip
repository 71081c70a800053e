// bvh_build.hip — cyBVH::Build (DataStructure/cyBVH.h:122-142,242-328) on the device, node for node.
//
// The reference builds a mesh's BVH by recursion: MeanSplit partitions a node's element list in place around the midpoint of
// the widest box axis (falling back to the other two axes), child boxes are the min / max of the children's element bounds,
// and ConvertTempData numbers the nodes depth-first with children allocated pairwise.  Node ids, leaf ranges and the element
// order are part of the hit record's meaning (the `prim` a ray reports, the order ties resolve in), so the device build
// reproduces all three exactly — level by level instead of node by node:
//
//   * MeanSplit's partition loop (`if (center <= split) i++; else swap(e[i], e[--j])`) has a closed form.  With k = number of
//     elements on the left side, f_1 < f_2 < ... the positions < k holding right-side elements and g_1 > g_2 > ... the positions
//     >= k holding left-side elements:  g_m -> f_m;  f_1 -> n-1, f_m -> g_(m-1) - 1;  a right-side element at position k goes
//     to g_last - 1 (n-1 if there is no g);  every other right-side element at p >= k moves to p - 1;  the rest stays.
//     (Checked against the loop itself for every flag pattern up to 13 elements.)  Ranks come from one exclusive scan of the
//     side flags over the whole element array; every node of a level is partitioned by the same three launches.
//   * an axis that puts every element on one side still permutes the list (all right-side: a rotation by one) before the next
//     axis is tried — so up to three partition rounds per level, each applied to the result of the previous one.
//   * child boxes: `if (b > v) b = v` keeps the FIRST element reaching the extreme (this decides between -0.0 and +0.0, which
//     OBJ exporters do write); 64-bit atomics on (ordered value, position) keep exactly that one.
//   * node ids: subtree sizes bottom-up, then ConvertTempData's numbering top-down (child pair at childIndex, the first
//     child's subtree allocates from childIndex + 2, the second child's after it).
//
// All of it is integer / compare work on a few arrays of N elements; nothing here is bound by anything but launch count at
// the mesh sizes the reference ships (<= 10^5 triangles), and by HBM streams beyond that.
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "bhrt.h"
#include "scene_internal.h"

namespace bhrt {

#define BVH_HIP_CHECK(expr)                                                                                    \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            SetError(std::string(#expr) + ": " + hipGetErrorString(e_));                                       \
            return BHRT_ERR_HIP;                                                                               \
        }                                                                                                      \
    } while (0)

namespace {

constexpr int kB = 256;
constexpr uint32_t kTileThreads = 1024, kPerThread = 4, kTile = kTileThreads * kPerThread;
enum : uint32_t { ST_NEW = 0, ST_ACTIVE = 1, ST_SPLIT = 2, ST_LEAF = 3 };

struct TNode { // temporary node, cyBVH.h:196-222
    uint32_t off, cnt, parent, child; // child = index of the first child (second = child + 1)
    uint32_t state, level, c1, size;
    uint32_t id, child_index;
    uint32_t sd[3];
    float split[3];
    float box[6];
    unsigned long long key[6]; // box reduction: (ordered value << 32) | position (min) / ~position (max)
};

struct Mesh { const float *v; const uint32_t *f; };

__device__ inline void elem_bounds(const Mesh &M, uint32_t e, float b[6]) // cyBVH.h:361-369
{
    const float *p = M.v + 3 * (size_t)M.f[3 * (size_t)e];
    b[0] = b[3] = p[0]; b[1] = b[4] = p[1]; b[2] = b[5] = p[2];
    for (int j = 1; j < 3; j++) {
        const float *q = M.v + 3 * (size_t)M.f[3 * (size_t)e + j];
        for (int k = 0; k < 3; k++) {
            if (b[k] > q[k]) b[k] = q[k];
            if (b[k + 3] < q[k]) b[k + 3] = q[k];
        }
    }
}
__device__ inline float elem_center(const Mesh &M, uint32_t e, uint32_t dim) // cyBVH.h:371-375
{
    const uint32_t *f = M.f + 3 * (size_t)e;
    return (M.v[3 * (size_t)f[0] + dim] + M.v[3 * (size_t)f[1] + dim] + M.v[3 * (size_t)f[2] + dim]) / 3.0f;
}
// order-preserving map float -> uint32 with -0.0 == +0.0 (the reference compares floats, so the two zeros tie)
__device__ inline uint32_t ord_key(float f)
{
    if (f == 0.f) f = 0.f;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- exclusive scan of the side flags over all N positions (S has N + 1 entries) -------------------------------------------
__device__ inline uint32_t block_scan_excl(uint32_t v, uint32_t *lds /* 17 */, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if ((int)lane >= off) incl += t; }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t w = 0; w < kTileThreads / 64; w++) { const uint32_t t = lds[w]; lds[w] = run; run += t; } lds[16] = run; }
    __syncthreads();
    const uint32_t r = lds[wave] + incl - v;
    total = lds[16];
    __syncthreads();
    return r;
}
// flags of one partition round, computed on the fly: 1 = the element goes to the first child (center <= split)
__global__ void __launch_bounds__(kTileThreads) k_bvh_scan_tiles(Mesh M, const uint32_t *elems, const uint32_t *enode, const TNode *nodes, uint32_t n, int round,
                                                                 uint32_t *S, uint32_t *tile_sums)
{
    __shared__ uint32_t lds[17];
    const uint32_t base = blockIdx.x * kTile + threadIdx.x * kPerThread;
    uint32_t v[kPerThread], sum = 0;
    for (uint32_t k = 0; k < kPerThread; k++) {
        const uint32_t i = base + k;
        uint32_t fl = 0;
        if (i < n) {
            const TNode &t = nodes[enode[i]];
            if (t.state == ST_ACTIVE) fl = elem_center(M, elems[i], t.sd[round]) <= t.split[round] ? 1u : 0u;
        }
        v[k] = sum;
        sum += fl;
    }
    uint32_t total;
    const uint32_t ex = block_scan_excl(sum, lds, total);
    for (uint32_t k = 0; k < kPerThread; k++) if (base + k < n) S[base + k] = ex + v[k];
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kTileThreads) k_bvh_scan_sums(uint32_t *tile_sums, uint32_t n_tiles, uint32_t *S, uint32_t n)
{
    __shared__ uint32_t lds[17];
    uint32_t carry = 0;
    for (uint32_t c = 0; c < n_tiles; c += kTileThreads) { // one workgroup, chunk after chunk
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_excl(v, lds, total);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) S[n] = carry;
}
__global__ void __launch_bounds__(kB) k_bvh_scan_add(uint32_t *S, uint32_t n, const uint32_t *tile_sums)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) S[i] += tile_sums[i / kTile];
}

// ---- one partition round ------------------------------------------------------------------------------------------------
// F[off + m] = position of the m-th misplaced element of the front part, G[off + m] = of the back part counted from the top
__global__ void __launch_bounds__(kB) k_bvh_rank(const uint32_t *enode, const TNode *nodes, const uint32_t *S, uint32_t n, uint32_t *F, uint32_t *G)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TNode &t = nodes[enode[i]];
    if (t.state != ST_ACTIVE) return;
    const uint32_t s0 = S[t.off], k = S[t.off + t.cnt] - s0, l = i - t.off, posL = S[i] - s0;
    const bool left = S[i + 1] != S[i];
    if (l < k && !left) F[t.off + (l - posL)] = i;
    if (l >= k && left) G[t.off + (k - posL - 1)] = i;
}
__global__ void __launch_bounds__(kB) k_bvh_permute(const uint32_t *enode, const TNode *nodes, const uint32_t *S, uint32_t n, const uint32_t *F, const uint32_t *G,
                                                    const uint32_t *elems_in, uint32_t *elems_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TNode &t = nodes[enode[i]];
    uint32_t dest = i;
    if (t.state == ST_ACTIVE) {
        const uint32_t s0 = S[t.off], k = S[t.off + t.cnt] - s0, l = i - t.off, posL = S[i] - s0;
        const bool left = S[i + 1] != S[i];
        if (l < k) {
            if (!left) { const uint32_t m = l - posL; dest = m == 0 ? t.off + t.cnt - 1 : G[t.off + m - 1] - 1; }
        } else if (left) dest = F[t.off + (k - posL - 1)];
        else if (l == k) {
            const uint32_t h = k - (S[t.off + k] - s0); // misplaced elements on either side
            dest = h > 0 ? G[t.off + h - 1] - 1 : t.off + t.cnt - 1;
        } else dest = i - 1;
    }
    elems_out[dest] = elems_in[i];
}
__global__ void __launch_bounds__(kB) k_bvh_round(TNode *nodes, uint32_t a, uint32_t b, const uint32_t *S)
{
    const uint32_t t = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    TNode &nd = nodes[t];
    if (nd.state != ST_ACTIVE) return;
    const uint32_t k = S[nd.off + nd.cnt] - S[nd.off];
    if (k > 0 && k < nd.cnt) { nd.c1 = k; nd.state = ST_SPLIT; } // cyBVH.h:321-324
}

// ---- per level ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kB) k_bvh_setup(TNode *nodes, uint32_t a, uint32_t b, uint32_t max_per) // MeanSplit head, cyBVH.h:297-303
{
    const uint32_t t = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    TNode &nd = nodes[t];
    if (nd.cnt <= max_per) { nd.state = ST_LEAF; return; }
    const float *box = nd.box;
    const float d[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
    uint32_t sd[3];
    sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
    sd[1] = (sd[0] + 1) % 3;
    sd[2] = (sd[0] + 2) % 3;
    if (d[sd[1]] < d[sd[2]]) { const uint32_t x = sd[1]; sd[1] = sd[2]; sd[2] = x; }
    for (int s = 0; s < 3; s++) { nd.sd[s] = sd[s]; nd.split[s] = 0.5f * (box[sd[s]] + box[sd[s] + 3]); }
    nd.state = ST_ACTIVE;
}
// after the three rounds: forced halving or leaf (cyBVH.h:250-260), then the two children (TempNode::Split, cyBVH.h:205-209)
__global__ void __launch_bounds__(kB) k_bvh_finish(TNode *nodes, uint32_t a, uint32_t b, uint32_t *n_nodes)
{
    const uint32_t t = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    TNode &nd = nodes[t];
    if (nd.state == ST_ACTIVE) {
        if (nd.cnt > 8u /* CY_BVH_MAX_ELEMENT_COUNT */) { nd.c1 = nd.cnt / 2; nd.state = ST_SPLIT; }
        else nd.state = ST_LEAF;
    }
    if (nd.state != ST_SPLIT) return;
    const uint32_t c = atomicAdd(n_nodes, 2u); // the order of the temporary nodes is free; ids are assigned from the tree's shape
    nd.child = c;
    for (uint32_t s = 0; s < 2; s++) {
        TNode &ch = nodes[c + s];
        ch.off = s == 0 ? nd.off : nd.off + nd.c1;
        ch.cnt = s == 0 ? nd.c1 : nd.cnt - nd.c1;
        ch.parent = t; ch.child = 0; ch.state = ST_NEW; ch.level = nd.level + 1; ch.c1 = 0; ch.size = 0; ch.id = 0; ch.child_index = 0;
        for (int k = 0; k < 3; k++) { ch.key[k] = ~0ull; ch.key[k + 3] = 0ull; }
    }
}
__device__ inline unsigned long long wave_min64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ inline unsigned long long wave_max64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, off);
        v = o > v ? o : v;
    }
    return v;
}
// elements move to their child node; the children's boxes gather (value, first position) extremes
__global__ void __launch_bounds__(kB) k_bvh_assign(Mesh M, const uint32_t *elems, uint32_t *enode, TNode *nodes, uint32_t n, uint32_t level)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t target = 0xffffffffu;
    float eb[6];
    if (i < n) {
        const TNode &t = nodes[enode[i]];
        if (t.state == ST_SPLIT && t.level == level) {
            target = t.child + ((i - t.off) < t.c1 ? 0u : 1u);
            enode[i] = target;
            elem_bounds(M, elems[i], eb);
        }
    }
    const uint32_t first = __shfl((int)target, 0);
    const bool uniform = __all(target == first);
    if (uniform && target == 0xffffffffu) return;
    unsigned long long kmin[3], kmax[3];
    for (int k = 0; k < 3; k++) {
        kmin[k] = target != 0xffffffffu ? (((unsigned long long)ord_key(eb[k]) << 32) | i) : ~0ull;
        kmax[k] = target != 0xffffffffu ? (((unsigned long long)ord_key(eb[k + 3]) << 32) | (0xffffffffu - i)) : 0ull;
    }
    if (uniform) { // whole wave feeds one child: six atomics per wave
        for (int k = 0; k < 3; k++) { kmin[k] = wave_min64(kmin[k]); kmax[k] = wave_max64(kmax[k]); }
        if (__lane_id() == 0)
            for (int k = 0; k < 3; k++) { atomicMin(&nodes[target].key[k], kmin[k]); atomicMax(&nodes[target].key[k + 3], kmax[k]); }
    } else if (target != 0xffffffffu) {
        for (int k = 0; k < 3; k++) { atomicMin(&nodes[target].key[k], kmin[k]); atomicMax(&nodes[target].key[k + 3], kmax[k]); }
    }
}
// Box::operator+= over the node's elements in list order (cyBVH.h:164,263-273): the first element reaching the extreme supplies the bits
__global__ void __launch_bounds__(kB) k_bvh_boxes(Mesh M, const uint32_t *elems, TNode *nodes, uint32_t a, uint32_t b)
{
    const uint32_t t = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    TNode &nd = nodes[t];
    for (int k = 0; k < 3; k++) {
        float eb[6];
        float lo = 1e30f, hi = -1e30f;
        if (nd.key[k] != ~0ull) { elem_bounds(M, elems[(uint32_t)nd.key[k]], eb); if (lo > eb[k]) lo = eb[k]; }
        if (nd.key[k + 3] != 0ull) { elem_bounds(M, elems[0xffffffffu - (uint32_t)nd.key[k + 3]], eb); if (hi < eb[k + 3]) hi = eb[k + 3]; }
        nd.box[k] = lo; nd.box[k + 3] = hi;
    }
}
// root: every element, position order
__global__ void __launch_bounds__(kB) k_bvh_root_keys(Mesh M, const uint32_t *elems, TNode *nodes, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float eb[6];
    if (i < n) elem_bounds(M, elems[i], eb);
    for (int k = 0; k < 3; k++) {
        unsigned long long kmin = i < n ? (((unsigned long long)ord_key(eb[k]) << 32) | i) : ~0ull;
        unsigned long long kmax = i < n ? (((unsigned long long)ord_key(eb[k + 3]) << 32) | (0xffffffffu - i)) : 0ull;
        kmin = wave_min64(kmin); kmax = wave_max64(kmax);
        if (__lane_id() == 0) { atomicMin(&nodes[0].key[k], kmin); atomicMax(&nodes[0].key[k + 3], kmax); }
    }
}
__global__ void k_bvh_init(TNode *nodes, uint32_t n, uint32_t *elems, uint32_t *enode, uint32_t *n_nodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { elems[i] = i; enode[i] = 0; }
    if (i == 0) {
        TNode &r = nodes[0];
        r.off = 0; r.cnt = n; r.parent = 0xffffffffu; r.child = 0; r.state = ST_NEW; r.level = 0; r.c1 = 0; r.size = 0; r.id = 1; r.child_index = 2;
        for (int k = 0; k < 3; k++) { r.key[k] = ~0ull; r.key[k + 3] = 0ull; }
        *n_nodes = 1;
    }
}

// ---- numbering (ConvertTempData, cyBVH.h:281-291) --------------------------------------------------------------------------
__global__ void __launch_bounds__(kB) k_bvh_sizes(TNode *nodes, uint32_t a, uint32_t b)
{
    const uint32_t t = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    TNode &nd = nodes[t];
    nd.size = nd.state == ST_SPLIT ? 1u + nodes[nd.child].size + nodes[nd.child + 1].size : 1u;
}
__global__ void __launch_bounds__(kB) k_bvh_ids(TNode *nodes, uint32_t a, uint32_t b)
{
    const uint32_t t = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    const TNode &nd = nodes[t];
    if (nd.state != ST_SPLIT) return;
    TNode &c1 = nodes[nd.child], &c2 = nodes[nd.child + 1];
    c1.id = nd.child_index; c1.child_index = nd.child_index + 2;
    c2.id = nd.child_index + 1; c2.child_index = nd.child_index + 1 + c1.size; // what ConvertTempData(child1) returns
}
__global__ void __launch_bounds__(kB) k_bvh_emit(const TNode *nodes, uint32_t n_nodes, bhrt_bvh_node *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes) return;
    const TNode &nd = nodes[t];
    bhrt_bvh_node o;
    for (int k = 0; k < 6; k++) o.b[k] = nd.box[k];
    o.parent = nd.parent == 0xffffffffu ? 0u : nodes[nd.parent].id;
    if (nd.state == ST_SPLIT) o.data = nd.child_index & 0x7fffffffu;                                      // SetInternalNode, cyBVH.h:181
    else o.data = (nd.off & ((1u << 28) - 1)) | ((nd.cnt - 1) << 28) | (1u << 31);                        // SetLeafNode, cyBVH.h:180
    out[nd.id] = o;
    if (t == 0) { bhrt_bvh_node z; memset(&z, 0, sizeof z); out[0] = z; }
}

struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t Alloc(T **p, size_t count)
    {
        const hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

} // namespace

// Host pointers in and out; runs on `device`.  nodes_out: at least 2 * nf entries (node 0 unused, root = 1).
int BuildBvhOnDevice(const float *vertices, uint32_t nv, const uint32_t *faces, uint32_t nf, uint32_t max_per, int device, bhrt_bvh_node *nodes_out,
                     uint32_t node_capacity, uint32_t *n_nodes_out, uint32_t *elems_out, uint32_t *depth_out)
{
    if (!vertices || !faces || !nodes_out || !elems_out || nf == 0 || nv == 0) { SetError("bvh build: bad arguments"); return BHRT_ERR_ARG; }
    if (nf >= (1u << 28)) { SetError("bvh build: too many faces for the 28-bit leaf offset"); return BHRT_ERR_UNSUPPORTED; }
    for (size_t k = 0; k < (size_t)nf * 3; k++)
        if (faces[k] >= nv) { SetError("bvh build: face index out of range"); return BHRT_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { SetError("no HIP device"); return BHRT_ERR_NO_DEVICE; }
    if (device < 0 || device >= count) { SetError("bvh build: bad device"); return BHRT_ERR_ARG; }
    BVH_HIP_CHECK(hipSetDevice(device));
    if (max_per > 8) max_per = 8; // cyBVH.h:127
    if (max_per == 0) max_per = 1;
    const uint32_t n = nf, max_nodes = 2 * n;
    Scratch scr;
    float *d_v; uint32_t *d_f, *d_elems[2], *d_enode, *d_S, *d_F, *d_G, *d_tiles, *d_count; TNode *d_nodes; bhrt_bvh_node *d_out;
    const uint32_t n_tiles = (n + kTile - 1) / kTile;
    BVH_HIP_CHECK(scr.Alloc(&d_v, (size_t)nv * 3)); BVH_HIP_CHECK(scr.Alloc(&d_f, (size_t)nf * 3));
    BVH_HIP_CHECK(scr.Alloc(&d_elems[0], n)); BVH_HIP_CHECK(scr.Alloc(&d_elems[1], n)); BVH_HIP_CHECK(scr.Alloc(&d_enode, n));
    BVH_HIP_CHECK(scr.Alloc(&d_S, (size_t)n + 1)); BVH_HIP_CHECK(scr.Alloc(&d_F, n)); BVH_HIP_CHECK(scr.Alloc(&d_G, n));
    BVH_HIP_CHECK(scr.Alloc(&d_tiles, n_tiles)); BVH_HIP_CHECK(scr.Alloc(&d_count, 1));
    BVH_HIP_CHECK(scr.Alloc(&d_nodes, max_nodes)); BVH_HIP_CHECK(scr.Alloc(&d_out, (size_t)max_nodes + 1));
    BVH_HIP_CHECK(hipMemcpy(d_v, vertices, (size_t)nv * 3 * sizeof(float), hipMemcpyHostToDevice));
    BVH_HIP_CHECK(hipMemcpy(d_f, faces, (size_t)nf * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
    const Mesh M = {d_v, d_f};
    const dim3 ge((n + kB - 1) / kB), be(kB);
    auto grid = [](uint32_t cnt) { return dim3((cnt + kB - 1) / kB); };
    hipStream_t st = nullptr; // the build is a load-time step: default stream
    hipLaunchKernelGGL(k_bvh_init, ge, be, 0, st, d_nodes, n, d_elems[0], d_enode, d_count);
    hipLaunchKernelGGL(k_bvh_root_keys, ge, be, 0, st, M, d_elems[0], d_nodes, n);
    hipLaunchKernelGGL(k_bvh_boxes, dim3(1), be, 0, st, M, d_elems[0], d_nodes, 0u, 1u);
    std::vector<uint32_t> level_start = {0, 1};
    int cur = 0;
    for (uint32_t level = 0;; level++) {
        const uint32_t a = level_start[level], b = level_start[level + 1];
        hipLaunchKernelGGL(k_bvh_setup, grid(b - a), be, 0, st, d_nodes, a, b, max_per);
        for (int round = 0; round < 3; round++) {
            hipLaunchKernelGGL(k_bvh_scan_tiles, dim3(n_tiles), dim3(kTileThreads), 0, st, M, d_elems[cur], d_enode, d_nodes, n, round, d_S, d_tiles);
            hipLaunchKernelGGL(k_bvh_scan_sums, dim3(1), dim3(kTileThreads), 0, st, d_tiles, n_tiles, d_S, n);
            hipLaunchKernelGGL(k_bvh_scan_add, ge, be, 0, st, d_S, n, d_tiles);
            hipLaunchKernelGGL(k_bvh_rank, ge, be, 0, st, d_enode, d_nodes, d_S, n, d_F, d_G);
            hipLaunchKernelGGL(k_bvh_permute, ge, be, 0, st, d_enode, d_nodes, d_S, n, d_F, d_G, d_elems[cur], d_elems[cur ^ 1]);
            hipLaunchKernelGGL(k_bvh_round, grid(b - a), be, 0, st, d_nodes, a, b, d_S);
            cur ^= 1;
        }
        hipLaunchKernelGGL(k_bvh_finish, grid(b - a), be, 0, st, d_nodes, a, b, d_count);
        uint32_t total = 0;
        BVH_HIP_CHECK(hipMemcpy(&total, d_count, sizeof total, hipMemcpyDeviceToHost)); // also orders the level's launches before the next
        if (total > max_nodes) { SetError("bvh build: node count exceeds 2 * faces"); return BHRT_ERR_OVERFLOW; }
        if (total == b) break; // no node of this level split
        hipLaunchKernelGGL(k_bvh_assign, ge, be, 0, st, M, d_elems[cur], d_enode, d_nodes, n, level);
        hipLaunchKernelGGL(k_bvh_boxes, grid(total - b), be, 0, st, M, d_elems[cur], d_nodes, b, total);
        level_start.push_back(total);
        if (level > 4096) { SetError("bvh build: runaway depth"); return BHRT_ERR_HIP; }
    }
    const uint32_t n_nodes = level_start.back(), n_levels = (uint32_t)level_start.size() - 1;
    for (uint32_t l = n_levels; l-- > 0;) hipLaunchKernelGGL(k_bvh_sizes, grid(level_start[l + 1] - level_start[l]), be, 0, st, d_nodes, level_start[l], level_start[l + 1]);
    for (uint32_t l = 0; l < n_levels; l++) hipLaunchKernelGGL(k_bvh_ids, grid(level_start[l + 1] - level_start[l]), be, 0, st, d_nodes, level_start[l], level_start[l + 1]);
    hipLaunchKernelGGL(k_bvh_emit, grid(n_nodes), be, 0, st, d_nodes, n_nodes, d_out);
    BVH_HIP_CHECK(hipGetLastError());
    if (n_nodes + 1 > node_capacity) { SetError("bvh build: nodes_out too small"); return BHRT_ERR_ARG; }
    BVH_HIP_CHECK(hipMemcpy(nodes_out, d_out, ((size_t)n_nodes + 1) * sizeof(bhrt_bvh_node), hipMemcpyDeviceToHost));
    BVH_HIP_CHECK(hipMemcpy(elems_out, d_elems[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_nodes_out) *n_nodes_out = n_nodes;
    if (depth_out) *depth_out = n_levels - 1;
    return BHRT_OK;
}

int BuildBvhDevice(HostMesh &m, unsigned max_per, int device)
{
    const uint32_t nf = (uint32_t)(m.f.size() / 3);
    if (nf == 0) { BuildBvh(m, max_per); return 0; } // nothing to partition: a single empty leaf
    m.bvh.assign(2 * (size_t)nf + 1, bhrt_bvh_node());
    m.elems.assign(nf, 0u);
    uint32_t n_nodes = 0, depth = 0;
    const int rc = BuildBvhOnDevice(m.v.data(), (uint32_t)(m.v.size() / 3), m.f.data(), nf, max_per, device, m.bvh.data(), (uint32_t)m.bvh.size(), &n_nodes, m.elems.data(), &depth);
    if (rc) return rc;
    m.bvh.resize((size_t)n_nodes + 1);
    m.bvh_depth = depth;
    return 0;
}

} // namespace bhrt

extern "C" int bhrt_bvh_build(const float *vertices, uint32_t n_vertices, const uint32_t *faces, uint32_t n_faces, uint32_t max_per_leaf, int device,
                              struct bhrt_bvh_node *nodes_out, uint32_t node_capacity, uint32_t *n_nodes, uint32_t *elems_out, uint32_t *depth)
try {
    return bhrt::BuildBvhOnDevice(vertices, n_vertices, faces, n_faces, max_per_leaf, device, nodes_out, node_capacity, n_nodes, elems_out, depth);
} catch (...) { return bhrt::AbiException(); }
