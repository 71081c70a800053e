// device_photon_build.h — PrepareForIrradianceEstimation / BalanceSegment (DataStructure/cyPhotonMap.h:236-328) on the device (round 3;
// SURVEY.md 2.2 "k_photon_build").  Included by kernels.hip.
//
// The balanced map is not a function of the photon SET: BalanceSegment picks its splitting axis from a box that is inherited, never
// recomputed (so deep segments of a caustic on a floor split along z, where every key is equal), and quick-selects the median with a Hoare
// partition whose result for equal keys depends on the order the elements are in — i.e. on every swap of every ancestor's partition.  A
// byte-identical map therefore needs the reference's swap sequence itself.  That sequence parallelises exactly: one partition pass over
// [left, right] with pivot v = key[right] stops its i-scan at the positions A (ascending) whose key is not < v and its j-scan at the positions
// B (descending, below `right`) whose key is not > v or that are `left`; it swaps A[m] with B[m] while A[m] < B[m] (K swaps: between two
// swaps the scans only cross elements no swap has touched), ends with i = A[K], or B[K-1] when the scan reaches the element the last swap
// put there first, and swaps i with `right`.  Two prefix counts, a count and a scatter: O(range) per pass for a whole workgroup.
//   k_pb_level: one workgroup per segment of a level runs the segment's whole quick-select (passes in parallel; ranges of <= kPbSeq elements by
//   one lane, the reference's loop as it stands), writes the median's record to its heap slot and hands the two sides to the next level.
// Comparisons are the reference's own (`<`, `>`: NaN and -0 behave as they do there).  The host routine (photon_host.cpp) stays as the
// second opinion of the tests (BHRT_PHOTON_BALANCE_HOST=1) and for maps loaded from a file.
#pragma once

namespace bhrt {

struct PbSeg {
    uint32_t heap, start, end;
    float bmin[3], bmax[3];
};
constexpr int kPbSeq = 32;

// bounding box over slots 0..n (the unused slot 0 included: cyPhotonMap.h:241-242) -> the root segment; one workgroup
__global__ void __launch_bounds__(1024) k_pb_root(const DPhoton *rec, uint32_t n, PbSeg *root, uint32_t *id)
{
    __shared__ float lo[3][16], hi[3][16];
    float l[3], h[3];
    for (int k = 0; k < 3; k++) { l[k] = rec[0].pos[k]; h[k] = rec[0].pos[k]; }
    for (uint32_t i = threadIdx.x; i <= n; i += 1024) {
        id[i] = i;
        for (int k = 0; k < 3; k++) { const float x = rec[i].pos[k]; if (l[k] > x) l[k] = x; if (h[k] < x) h[k] = x; }
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) { const float a = __shfl_xor(l[k], off), b = __shfl_xor(h[k], off); if (l[k] > a) l[k] = a; if (h[k] < b) h[k] = b; }
        if ((threadIdx.x & 63) == 0) { lo[k][threadIdx.x >> 6] = l[k]; hi[k][threadIdx.x >> 6] = h[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PbSeg s;
        s.heap = 1; s.start = 1; s.end = n;
        for (int k = 0; k < 3; k++) {
            float a = lo[k][0], b = hi[k][0];
            for (int w = 1; w < 16; w++) { if (a > lo[k][w]) a = lo[k][w]; if (b < hi[k][w]) b = hi[k][w]; }
            s.bmin[k] = a; s.bmax[k] = b;
        }
        *root = s;
    }
}

// the reference's loop on [left, right] (cyPhotonMap.h:278-293), one lane
__device__ inline void pb_select_seq(float *key, uint32_t *id, int left, int right, int median)
{
    while (right > left) {
        const float v = key[right];
        int i = left - 1, j = right;
        while (key[++i] < v) {}
        while (key[--j] > v && j > left) {}
        while (i < j) {
            { const float t = key[i]; key[i] = key[j]; key[j] = t; const uint32_t u = id[i]; id[i] = id[j]; id[j] = u; }
            while (key[++i] < v) {}
            while (key[--j] > v && j > left) {}
        }
        { const float t = key[i]; key[i] = key[right]; key[right] = t; const uint32_t u = id[i]; id[i] = id[right]; id[right] = u; }
        if (i >= median) right = i - 1;
        if (i <= median) left = i + 1;
    }
}
template <int T>
__global__ void __launch_bounds__(T) k_pb_level(const DPhoton *rec, uint32_t *id, float *key, uint32_t *listA, uint32_t *listB, const PbSeg *segs, uint32_t n_segs,
                                                 PbSeg *next, uint32_t *next_count, DPhoton *out)
{
    __shared__ uint32_t lds_a[T / 64], lds_b[T / 64];
    __shared__ int s_left, s_right;
    __shared__ uint32_t s_k;
    if (blockIdx.x >= n_segs) return;
    const PbSeg sg = segs[blockIdx.x];
    const int start = (int)sg.start, end = (int)sg.end;
    // the median of a left-balanced tree (cyPhotonMap.h:264-273)
    int median = 1;
    while ((4 * median) <= (end - start + 1)) median += median;
    if ((3 * median) <= (end - start + 1)) { median += median; median += start - 1; }
    else median = end - median + 1;
    int axis = 2; // the widest side of the INHERITED box (cyPhotonMap.h:275-281)
    const float dx = sg.bmax[0] - sg.bmin[0], dy = sg.bmax[1] - sg.bmin[1], dz = sg.bmax[2] - sg.bmin[2];
    if (dx > dy) { if (dx > dz) axis = 0; }
    else if (dy > dz) axis = 1;
    for (int p = start + (int)threadIdx.x; p <= end; p += T) key[p] = rec[id[p]].pos[axis];
    if (threadIdx.x == 0) { s_left = start; s_right = end; }
    __syncthreads();
    for (;;) {
        const int left = s_left, right = s_right;
        if (right <= left) break;
        if (right - left + 1 <= kPbSeq) {
            if (threadIdx.x == 0) pb_select_seq(key, id, left, right, median);
            break;
        }
        const float v = key[right];
        // A: ascending positions of [left, right] where the i-scan stops; B: descending positions of [left, right - 1] where the j-scan stops.
        // Every wave takes a contiguous span of the range (of the reversed range for B), counts its stops with ballots, the wave counts are
        // prefix-summed once per pass, and a second sweep writes the lists: coalesced reads, two workgroup barriers per pass.
        constexpr int W = T / 64;
        const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
        const int len = right - left + 1, S = (((len + W - 1) / W) + 63) & ~63;
        const int a0 = wave * S, a1 = min(a0 + S, len);         // offsets from `left` (A) / indices k of p = right - 1 - k (B), k < len - 1
        const int b1 = min(a0 + S, len - 1);
        uint32_t ca = 0, cb = 0;
        for (int base = a0; base < a1; base += 64) {
            const int o = base + lane;
            ca += (uint32_t)__popcll(__ballot(o < a1 && !(key[left + o] < v)));
        }
        for (int base = a0; base < b1; base += 64) {
            const int k = base + lane, p = right - 1 - k;
            cb += (uint32_t)__popcll(__ballot(k < b1 && (!(key[p] > v) || p == left)));
        }
        if (lane == 0) { lds_a[wave] = ca; lds_b[wave] = cb; }
        __syncthreads();
        uint32_t na = 0, nb = 0, ra = 0, rb = 0;
        for (int w = 0; w < W; w++) { if (w == wave) { ra = na; rb = nb; } na += lds_a[w]; nb += lds_b[w]; }
        const uint64_t lt = (1ull << lane) - 1ull;
        for (int base = a0; base < a1; base += 64) {
            const int o = base + lane;
            const bool f = o < a1 && !(key[left + o] < v);
            const uint64_t m = __ballot(f);
            if (f) listA[left + (int)ra + __popcll(m & lt)] = (uint32_t)(left + o);
            ra += (uint32_t)__popcll(m);
        }
        for (int base = a0; base < b1; base += 64) {
            const int k = base + lane, p = right - 1 - k;
            const bool f = k < b1 && (!(key[p] > v) || p == left);
            const uint64_t m = __ballot(f);
            if (f) listB[left + (int)rb + __popcll(m & lt)] = (uint32_t)p;
            rb += (uint32_t)__popcll(m);
        }
        __syncthreads();
        // K swaps: A[m] < B[m] (A ascends, B descends: the pairs that satisfy it are the first K)
        const uint32_t nm = na < nb ? na : nb;
        uint32_t mine = 0;
        for (uint32_t m = threadIdx.x; m < nm; m += T) mine += listA[left + (int)m] < listB[left + (int)m] ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (threadIdx.x == 0) s_k = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_k, mine);
        __syncthreads();
        const uint32_t K = s_k;
        for (uint32_t m = threadIdx.x; m < K; m += T) {
            const uint32_t a = listA[left + (int)m], b = listB[left + (int)m];
            const float t = key[a]; key[a] = key[b]; key[b] = t;
            const uint32_t u = id[a]; id[a] = id[b]; id[b] = u;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int i = (int)listA[left + (int)K]; // K <= na - 1: `right` itself is the last of A and no B lies at or above it
            if (K > 0 && (int)listB[left + (int)K - 1] < i) i = (int)listB[left + (int)K - 1];
            { const float t = key[i]; key[i] = key[right]; key[right] = t; const uint32_t u = id[i]; id[i] = id[right]; id[right] = u; }
            if (i >= median) s_right = i - 1;
            if (i <= median) s_left = i + 1;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) { // cyPhotonMap.h:295-327
        DPhoton m = rec[id[median]];
        m.planeAndDirZ = (uint8_t)((m.planeAndDirZ & 0x8) | (axis & 0x3));
        out[sg.heap] = m;
        if (median > start) {
            if (start < median - 1) {
                PbSeg c = sg;
                c.heap = 2 * sg.heap; c.start = (uint32_t)start; c.end = (uint32_t)(median - 1);
                c.bmax[axis] = m.pos[axis];
                next[atomicAdd(next_count, 1u)] = c;
            } else out[2 * sg.heap] = rec[id[start]];
        }
        if (median < end) {
            if (median + 1 < end) {
                PbSeg c = sg;
                c.heap = 2 * sg.heap + 1; c.start = (uint32_t)(median + 1); c.end = (uint32_t)end;
                c.bmin[axis] = m.pos[axis];
                next[atomicAdd(next_count, 1u)] = c;
            } else out[2 * sg.heap + 1] = rec[id[end]];
        }
    }
}

// one batch of emissions: prefix = exclusive prefix of counts (k_scan_* have made it) -> stats[0] = largest count (zeroed by the host), stats[1] = photons of the batch
__global__ void __launch_bounds__(256) k_pb_batch_stats(const uint32_t *counts, const uint32_t *prefix, uint32_t E, uint32_t *stats)
{
    uint32_t mx = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < E; i += gridDim.x * 256) mx = max(mx, counts[i]);
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&stats[0], mx);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[1] = prefix[E - 1] + counts[E - 1];
}

// PhotonMapDev::dbox, bottom up one heap level per launch: bounds of the DIRECTIONS of the photons LocatePhotons reaches below node i (it recurses
// only below `half`), itself included
__global__ void __launch_bounds__(256) k_pb_dbox_level(const DPhoton *ph, uint32_t n, int half, uint32_t first, uint32_t last, float *lo, float *hi)
{
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i > last || i > n) return;
    const V3 d = photon_direction(ph[i]);
    float l3[3] = {d.x, d.y, d.z}, h3[3] = {d.x, d.y, d.z};
    if ((int)i < half)
        for (uint64_t c = 2ull * i; c <= 2ull * i + 1 && c <= n; c++)
            for (int k = 0; k < 3; k++) { l3[k] = fminf(l3[k], lo[c * 3 + k]); h3[k] = fmaxf(h3[k], hi[c * 3 + k]); }
    for (int k = 0; k < 3; k++) { lo[(size_t)i * 3 + k] = l3[k]; hi[(size_t)i * 3 + k] = h3[k]; }
}
__global__ void __launch_bounds__(256) k_pb_dbox_pack(const float *lo, const float *hi, uint32_t nb, float4 *box)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    if (i == 0) { box[0] = make_float4(0, 0, 0, 0); box[1] = make_float4(0, 0, 0, 0); return; }
    box[2 * (size_t)i] = make_float4(lo[(size_t)i * 3], lo[(size_t)i * 3 + 1], lo[(size_t)i * 3 + 2], 0.f);
    box[2 * (size_t)i + 1] = make_float4(hi[(size_t)i * 3], hi[(size_t)i * 3 + 1], hi[(size_t)i * 3 + 2], 0.f);
}
// bounds of the photon positions 1..n (PhotonMapDev::lo / hi): one workgroup
__global__ void __launch_bounds__(1024) k_pb_bounds(const DPhoton *ph, uint32_t n, float *out6)
{
    __shared__ float lo[3][16], hi[3][16];
    float l[3] = {BHRT_BIGFLOAT, BHRT_BIGFLOAT, BHRT_BIGFLOAT}, h[3] = {-BHRT_BIGFLOAT, -BHRT_BIGFLOAT, -BHRT_BIGFLOAT};
    for (uint32_t i = 1 + threadIdx.x; i <= n; i += 1024)
        for (int k = 0; k < 3; k++) { l[k] = fminf(l[k], ph[i].pos[k]); h[k] = fmaxf(h[k], ph[i].pos[k]); }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) { l[k] = fminf(l[k], __shfl_xor(l[k], off)); h[k] = fmaxf(h[k], __shfl_xor(h[k], off)); }
        if ((threadIdx.x & 63) == 0) { lo[k][threadIdx.x >> 6] = l[k]; hi[k][threadIdx.x >> 6] = h[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = (int)threadIdx.x;
        float a = lo[k][0], b = hi[k][0];
        for (int w = 1; w < 16; w++) { a = fminf(a, lo[k][w]); b = fmaxf(b, hi[k][w]); }
        out6[k] = a; out6[3 + k] = b;
    }
}

} // namespace bhrt
