/* bhrt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, never shipped).
 *
 * The oracle is a CPU restatement of the reference's per-pixel render path, evaluated on
 * the same flattened scene blob (include/bhrt_flat.h) the HIP kernels read.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (bhraytracer_amd/) never includes, links or calls anything under oracle/.
 *
 * Pinning: the reference has no tests or golden vectors of its own, so the
 * oracle is pinned against the reference itself, compiled in place into oracle/_ref
 * (oracle/Makefile, oracle/ref_harness/), and against the fixtures that binary generated
 * (tests/golden/, generator committed as tests/golden/make_golden.py).
 */
#ifndef BHRT_ORACLE_H
#define BHRT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_RNG_SEQUENTIAL = 0, ORACLE_RNG_KEYED = 1 };
enum { ORACLE_MATH_LIBM = 0, ORACLE_MATH_DEVICE = 1 };

typedef struct oracle_opts {
    int32_t spp;              /* PT_SampleCount, Main.cpp:141 */
    int32_t gi_bounces;       /* GIBounceCount, Main.cpp:130 */
    int32_t internal_bounces; /* INTERNAL_REFLECTION_BOUNCE, Main.cpp:41 */
    uint32_t seed;
    int32_t rng_mode;  /* ORACLE_RNG_* */
    int32_t math_mode; /* ORACLE_MATH_* */
    int32_t jitter;    /* 1: RandomPositionInPixel (Main.cpp:132-139); 0: ray through the pixel corner */
    int32_t x0, y0, x1, y1; /* pixel region [x0,x1) x [y0,y1); x1 <= 0 means the full image */
    int32_t threads;        /* OpenMP threads (0 = runtime default) */
    int32_t photon_gather;  /* 1: add the caustic photon-map term (MtlBlinn.cpp:329-342) */
} oracle_opts;

typedef struct oracle_stats {
    uint64_t closest_rays; /* top-level recursive() calls (Main.cpp:389) */
    uint64_t shadow_rays;  /* GenLight::Shadow calls (GenLight.cpp:10) */
    uint64_t shade_calls;  /* MtlBlinn::Shade calls */
    uint64_t samples;
    double seconds;
} oracle_stats;

/* hit attribute record: 16 floats = z, p[3], N[3], uvw[3], duvw0[3], duvw1[3] */
#define ORACLE_HIT_FLOATS 16

/* recursive() (Main.cpp:389-413) for n rays; rays = n x (o[3], d[3]).
 * node[i] = flattened node index or -1; face[i] = triangle id or -1; front[i] = 0/1. */
int oracle_trace_closest(const void *blob, const float *rays, int hit_side, size_t n, int32_t *node, int32_t *face,
                         int32_t *front, float *attrs);
/* GenLight::Shadow (GenLight.cpp:10-13): vis[i] = 0 (occluded) or 1 */
int oracle_trace_shadow(const void *blob, const float *rays, const float *tmax, size_t n, float *vis);
/* The frame loop (Main.cpp:143-172,204-234) over a pixel region.
 * samples: region_pixels*spp*3 per-sample radiance (may be NULL); radiance: region_pixels*3 averages
 * (pre-gamma); rgb8: region_pixels*3 after gamma + Color24.  Any output may be NULL. */
int oracle_render(const void *blob, const oracle_opts *opts, float *samples, float *radiance, uint8_t *rgb8,
                  oracle_stats *stats);
/* cyBVH build (cyBVH.h:122-142) restated independently of the front-end: nodes out as 8 x uint32 per node
 * (6 bounds bits, data, parent), elems out as nf x uint32.  Returns node count (incl. slot 0). */
int oracle_bvh_build(const float *v, const uint32_t *f, uint32_t nf, uint32_t max_per_leaf, uint32_t *nodes_out,
                     size_t nodes_cap, uint32_t *elems_out);
/* elementary functions in device-math mode (for tests/test_detmath.py); fn: 0 sin 1 cos 2 tan 3 acos 4 asin 5 atan2 6 pow */
int oracle_math_eval(int fn, int math_mode, const float *a, const float *b, size_t n, float *out);

/* caustic photon map (Main.cpp:342-386, cyPhotonMap.h) — see bhrt_oracle.cpp */
int oracle_photon_build(const void *blob, const oracle_opts *opts, uint32_t max_photons, void *photons_out /* 24 B each */,
                        uint32_t *n_stored, uint64_t *n_emitted);
/* images beside the colour image: first hit of every pixel's un-jittered camera ray (z: Main.cpp:231 / scene.h:532; normal and
 * albedo: the optional DenoiseImage inputs, Main.cpp:70-71), ComputeZBufferImage (scene.h:578-600), colorArray (Main.cpp:219-229) */
int oracle_first_hit(const void *blob, int math_mode, float *z, float *normal, float *albedo);
int oracle_zbuffer_image(const float *zbuffer, size_t size, uint8_t *zbufferImg);
int oracle_color_image(const float *radiance, size_t n_floats, int gamma, int math_mode, float *color);
/* global photon map, BuildPhotonMap (Main.cpp:251-317, MtlBlinn.cpp:140-202): balanced + emission-order records (24 B each) */
int oracle_photon_build_global(const void *blob, const oracle_opts *opts, uint32_t max_photons, void *photons_out, void *emitted_out,
                               uint32_t *n_stored, uint64_t *n_emitted);
int oracle_photon_unbalanced(void *out);                     /* the same photons in emission order (n_stored records) */
int oracle_photon_attach(const void *photons, uint32_t n);  /* n balanced (heap-order) records, e.g. from the HIP path */
int oracle_photon_balance(const void *emitted, uint32_t n, void *balanced_out); /* PrepareForIrradianceEstimation on n records */
int oracle_photon_gather(const float *p, const float *nrm, size_t cnt, float radius, float *irrad, float *dir);

/* the photons LocatePhotons (cyPhotonMap.h:421-498) ends up with: indices into the balanced map (1-based, ascending, 0-padded to 1000), their
 * number, and np.dist2[0] */
int oracle_photon_knn(const float *p, const float *nrm, size_t cnt, float radius, uint32_t *idx, uint32_t *count, float *d2max);

/* BeginRender() as the reference's whole program runs it with one thread (Main.cpp:178-242): one rand() stream for the photon build
 * (photon_budget > 0: BuildCausticPhotonMap with that budget, Main.cpp:342-386, and the gather in Shade) and the pixel loop in
 * column-major order, opts->spp samples per pixel.  rgb8: W*H*3 = RenderImage::GetPixels().  photons_out: the balanced map. */
int oracle_begin_render(const void *blob, const oracle_opts *opts, uint32_t photon_budget, uint8_t *rgb8, void *photons_out, uint32_t *n_stored,
                        uint64_t *n_emitted, uint64_t *draws);

const char *oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
