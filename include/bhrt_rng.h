/* bhrt_rng.h — the counter-based random stream that replaces libc rand() on this path.
 *
 * The reference draws every random number from glibc's global rand()
 * (Materials/Blinn/MtlBlinn.cpp:42-49, Main.cpp:136-137).  A global sequential
 * generator cannot be reproduced by a GPU wavefront, so this framework defines a
 * counter-based stream with the same *range* (31-bit ints, RAND_MAX = 2^31-1 as on
 * glibc) and two ways of addressing it:
 *
 *   sequential mode  one stream per (pixel, sample); the draw counter runs through the
 *                    whole depth-first Shade() recursion exactly where the reference
 *                    calls rand().  Used to pin the CPU oracle against the compiled
 *                    reference (whose rand() is interposed by the same stream).
 *   keyed mode       one stream per (pixel, sample, shade-call path code, section);
 *                    each stream has its own counter.  This is what the HIP wavefront
 *                    path and the oracle's keyed mode use, so both are bit-comparable.
 *
 * Plain C, integer only: results are identical on host and device by construction.
 */
#ifndef BHRT_RNG_H
#define BHRT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BHRT_HD __host__ __device__
#else
#define BHRT_HD
#endif

#define BHRT_RAND_MAX 2147483647 /* glibc RAND_MAX; the reference divides by it */

/* sections of one Shade() call (MtlBlinn.cpp:117,124,130) that own a sub-stream in keyed mode */
#define BHRT_SEC_REFRACTION 0u
#define BHRT_SEC_GI 1u
#define BHRT_SEC_DIRECT 2u

static inline BHRT_HD uint32_t bhrt_mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* key of the per-(pixel,sample) stream; pixel = j*W+i in the reference's row-major order */
static inline BHRT_HD uint32_t bhrt_sample_key(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    return bhrt_mix32(bhrt_mix32(seed * 0x9E3779B9U + pixel) + sample * 0x85EBCA6BU + 0x1B873593U);
}

/* key of one section of one Shade() call; path code: root = 1, refraction child = 2c, GI child = 2c+1 */
static inline BHRT_HD uint32_t bhrt_section_key(uint32_t sample_key, uint64_t path_code, uint32_t section)
{
    uint32_t lo = (uint32_t)path_code, hi = (uint32_t)(path_code >> 32);
    uint32_t k = bhrt_mix32(sample_key ^ bhrt_mix32(lo + 0x68E31DA4U));
    k = bhrt_mix32(k ^ bhrt_mix32(hi + 0xB5297A4DU));
    return bhrt_mix32(k + section * 0x1B56C4E9U + 0x7F4A7C15U);
}

/* photon pass (Main.cpp:342-386): the reference's emission loop is one long sequential rand() sequence.
 * sequential mode: ONE stream for the whole loop (key below with emission = all ones);
 * keyed mode: one stream per emission index, counter runs along that photon's path. */
static inline BHRT_HD uint32_t bhrt_photon_key_sequential(uint32_t seed) { return bhrt_sample_key(seed, 0xFFFFFFFFu, 0x50484F54u); }
static inline BHRT_HD uint32_t bhrt_photon_key(uint32_t seed, uint64_t emission)
{
    uint32_t k = bhrt_photon_key_sequential(seed);
    k = bhrt_mix32(k ^ bhrt_mix32((uint32_t)emission + 0x3C6EF372U));
    return bhrt_mix32(k ^ bhrt_mix32((uint32_t)(emission >> 32) + 0x9E3779B9U));
}

/* Keyed photon stream of a 64-bit emission index: (key, first counter).  A 32-bit key alone gives 2^32 streams, and the ~2e7
 * emissions of a 1 M-photon caustic map would then contain ~5e4 pairs that replay an identical path (birthday bound).  Emissions
 * [65536 m, 65536 m + 65536) therefore share key(m) and take disjoint 2^16-draw windows of its counter: a 1 M-photon build uses a
 * few hundred keys (collision probability ~1e-5), and two emissions never read the same (key, counter) pair: the counter of an emission
 * advances in its low 16 bits only (BHRT_PHOTON_WINDOW_MASK; kernels and oracle alike), so a path that draws more than 2^16 numbers — a
 * degenerate hit whose rejection loop runs to its cap — starts over in its own window instead of entering its successors'. */
#define BHRT_PHOTON_WINDOW_MASK 0xFFFFu
static inline BHRT_HD void bhrt_photon_stream(uint32_t seed, uint64_t emission, uint32_t *key, uint32_t *counter)
{
    *key = bhrt_photon_key(seed, emission >> 16);
    *counter = (uint32_t)(emission & 0xFFFFu) << 16;
}

/* the counter-th draw of stream `key`: an int in [0, BHRT_RAND_MAX] like rand() */
static inline BHRT_HD int32_t bhrt_rand31(uint32_t key, uint32_t counter)
{
    return (int32_t)(bhrt_mix32(key ^ bhrt_mix32(counter + 0x632BE5ABU)) >> 1);
}

#endif /* BHRT_RNG_H */
