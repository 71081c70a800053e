/* bhrt.h — C ABI of the MI355X-native render path (libbhrt.so).
 *
 * Drop-in boundary for the per-pixel render path of BosonHBC/BHRayTracer.  The reference has
 * no FFI: its seam is three free functions plus `extern` globals shared between the UI and the
 * renderer (all paths relative to /root/reference/BHRayTracer):
 *     int  LoadScene(char const *filename);   Main.cpp:43, xmlload.cpp:65
 *     void BeginRender();                     Main.cpp:178   (called from viewport.cpp:425-449)
 *     void StopRender();                      Main.cpp:243
 *     globals rootNode, camera, renderImage, lights, materials, ...   Main.cpp:17-37
 * and, one level down, the plugin virtuals of Scenes/scene.h (Object::IntersectRay :256,
 * Light::Illuminate :268, Material::Shade :291, Texture::Sample :314) reached through
 * recursive() (Main.cpp:389) and GenLight::Shadow (Lights/GenLight.cpp:10).
 * This header exports the same units with explicit ownership (an opaque scene handle instead of
 * globals) and runtime options instead of the reference's compile-time #defines.
 *
 * Threading: calls on one bhrt_scene must not overlap; different scenes are independent.
 * Errors: every function returns 0 on success, non-zero otherwise; bhrt_last_error() gives the
 * message for the calling thread.  Compute entry points fail (never fall back to the CPU) when
 * no gfx950 device is usable.
 */
#ifndef BHRT_H
#define BHRT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bhrt_scene bhrt_scene;

/* BHRT_HIT_* as in Scenes/scene.h:57-60 */
#define BHRT_SIDE_FRONT 1
#define BHRT_SIDE_BACK 2
#define BHRT_SIDE_BOTH 3

enum {
    BHRT_OK = 0,
    BHRT_ERR_IO = 1,
    BHRT_ERR_PARSE = 2,
    BHRT_ERR_ARG = 3,
    BHRT_ERR_NO_DEVICE = 4,
    BHRT_ERR_HIP = 5,
    BHRT_ERR_UNSUPPORTED = 6,
    BHRT_ERR_OVERFLOW = 7
};

typedef struct bhrt_info { /* replaces reading the globals camera / rootNode / lights (Main.cpp:17-30) */
    int32_t width, height;
    uint32_t n_nodes, n_meshes, n_triangles, n_bvh_nodes, n_materials, n_lights, n_textures;
    uint32_t max_node_depth, max_bvh_depth;
    uint64_t flat_bytes;
    uint32_t n_warnings;
} bhrt_info;

/* Runtime form of the reference's compile-time knobs (SURVEY.md §5 "Config / flags"). */
typedef struct bhrt_opts {
    int32_t spp;              /* PT_SampleCount (Main.cpp:141), default 32 */
    int32_t gi_bounces;       /* GIBounceCount (Main.cpp:130), default 3 */
    int32_t internal_bounces; /* INTERNAL_REFLECTION_BOUNCE (Main.cpp:41), default 16 */
    uint32_t seed;            /* stream seed of include/bhrt_rng.h (the reference never calls srand) */
    int32_t jitter;           /* 1 = RandomPositionInPixel (Main.cpp:132-139); 0 = ray through the pixel corner */
    int32_t gamma;            /* USE_GamaCorrection (Main.cpp:128): 1 = pow(c, 1/2.2f) before Color24 */
    int32_t photon_map;       /* USE_PhotonMap (Main.cpp:51): 1 = gather from the caustic map built by bhrt_photon_build */
    /* tile partition: this process renders tiles t with t % world_size == rank */
    int32_t rank, world_size;
    int32_t tile_size;        /* square tile edge in pixels, default 32 */
    int32_t samples_per_pass; /* upper bound on camera samples in flight per wavefront pass.  Device memory per sample: ~1 KB (1.4 KB with the photon map)
                                 when six Shade() frames are provided per sample slot.  0 = choose: what the frame needs, at most 2^28 slots, halved until
                                 the workspace fits into 85 % of the free memory — with six frames per slot that is 2^27 (~138 GB / 186 GB); a frame too
                                 large for one such pass (config 4: 2.65e8 samples per GPU) provides the frames per slot its earlier passes needed + 30 %
                                 instead (1.7 for config 4: 2^28 slots in 180 GB) and is one pass from its second render on */
    int32_t timers;           /* HIP-event kernel timers of bhrt_stats: 0 = seconds_shade only (default; an event between two kernels
                               * idles the GPU ~6 us), 1 = all kernel groups, -1 = none */
    int32_t photon_exact;     /* caustic gather of queries with >= 1000 photons inside the radius: 0 (default) = the same photon SET as
                               * LocatePhotons (cyPhotonMap.h:421-498) found by a wave-cooperative selection, sums in a fixed order (irradiance
                               * equal to a few ulp); 1 = the reference's candidate-heap history replayed lane by lane, identical bits, ~5x slower */
    int32_t leaf_skip;        /* 1 = the mesh walks leave out a box-missed LEAF sibling (TriObj.cpp:245-248,263-266,286-300) when its visit provably
                               * accepts nothing (bhrt_flat.h: bhrt_mesh::skip_*; proof in scene_host.cpp::ComputeLeafSkip).  Same hit records either
                               * way; 0 (default): on the scenes measured the test costs more instructions than the visits it saves */
    float photon_radius;      /* gather radius of the caustic term, MAX_Area (MtlBlinn.cpp:29); 0 = the reference's 0.5.  The photon count of an estimate,
                               * MAX_PhotonCountInArea = 1000 (MtlBlinn.cpp:28), is a TEMPLATE argument in the reference too (EstimateIrradiance<1000>,
                               * MtlBlinn.cpp:333) and sizes the candidate lists of the kernels: compile-time here as there (device_photon.h: BHRT_PHOTON_K) */
    int32_t reserved[1];
} bhrt_opts;

typedef struct bhrt_stats {
    uint64_t closest_rays;  /* top-level recursive() equivalents traced (Main.cpp:389) */
    uint64_t shadow_rays;   /* GenLight::Shadow equivalents traced (GenLight.cpp:10) */
    uint64_t shade_calls;   /* MtlBlinn::Shade equivalents evaluated */
    uint64_t camera_samples;
    uint32_t passes, wave_iterations;
    double seconds_total;    /* wall clock of the call, scene already resident */
    double seconds_trace_closest, seconds_trace_shadow, seconds_shade, seconds_other; /* HIP-event kernel time */
    uint64_t launches_trace_closest, launches_trace_shadow;
    /* caustic gather (EstimateIrradiance<1000>, DataStructure/cyPhotonMap.h:332-382; called at MtlBlinn.cpp:334) */
    double seconds_photon_gather;  /* wall clock of the gather of every pass, HIP events */
    double seconds_photon_heavy;   /* ... of which in the pass for queries with >= 1000 photons inside the radius */
    uint64_t photon_queries;       /* Shade() frames that asked for the caustic term */
    uint64_t photon_heavy_queries; /* queries that met 1000 photons */
    uint64_t photon_wave_queries;  /* queries whose walk was handed to a whole wave */
    uint64_t photon_exact_queries; /* heavy queries answered by the exact replay of the reference's candidate heap */
    uint64_t photon_nodes_visited; /* kd-tree nodes whose photon was examined (24 B each: SURVEY.md 8d) */
    uint64_t deferred_rays;        /* rays parallel to a coordinate axis of the mesh they enter (Box.cpp:13-28 ignores that axis: a walk of
                                    * nearly the whole BVH): traced in wave steps of their own at the end of their pass */
    /* the lane pass of the gather (k_photon_gather_fast) alone: queries it answered, kd nodes it examined, and — only with the knob "gather_stats"
     * (bhrt_scene_knob: a statistics instantiation of the kernel, 7 % slower) — the photons those answers were made of */
    uint64_t photon_lane_queries, photon_lane_nodes, photon_found;
    double reserved[1];
} bhrt_stats;

/* compact hit record written by the trace kernel (SoA on the device: one array per field) */
typedef struct bhrt_hits {
    float *t;         /* HitInfo::z = ray parameter, BIGFLOAT on a miss */
    int32_t *node;    /* flattened node index (DFS pre-order of the Node tree), -1 on a miss */
    int32_t *prim;    /* triangle id for mesh hits, -1 otherwise */
    int32_t *front;   /* HitInfo::front */
} bhrt_hits;

const char *bhrt_last_error(void);
void bhrt_default_opts(bhrt_opts *opts);

/* ---- scene (= LoadScene + the globals it fills) ------------------------------------------------ */
int bhrt_scene_load_xml(const char *path, bhrt_scene **out);            /* xmlload.cpp:65 */
/* same, with the mesh BVHs (TriObj::Load -> cyBVHTriMesh::SetMesh(this, 4), objects.h:59) built on HIP device `bvh_device`
 * instead of by the host front-end; -1 = host.  The flattened scene is byte-identical either way. */
int bhrt_scene_load_xml_ex(const char *path, int bvh_device, bhrt_scene **out);
/* cyBVH::Build (DataStructure/cyBVH.h:122-142; SplitTempNode :242-278, ConvertTempData :281-291, MeanSplit :295-328) on the
 * device, node for node: ids, boxes, leaf ranges and element order equal the reference's recursive build.  Host pointers.
 * vertices: n_vertices xyz triples; faces: n_faces index triples; nodes_out (bhrt_bvh_node of include/bhrt_flat.h): capacity
 * >= 2 * n_faces + 1 is always enough (node 0 unused, root = 1; *n_nodes = nodes without slot 0); elems_out: n_faces. */
struct bhrt_bvh_node;
int bhrt_bvh_build(const float *vertices, uint32_t n_vertices, const uint32_t *faces, uint32_t n_faces, uint32_t max_per_leaf, int device,
                   struct bhrt_bvh_node *nodes_out, uint32_t node_capacity, uint32_t *n_nodes, uint32_t *elems_out, uint32_t *depth);
/* A second handle on the same loaded scene (host copy of the flattened scene, no device state): the reference's globals are
 * one per process; a host that drives several GPUs keeps one handle per device (bhrt_scene_upload) — see csrc/bhrt_main.cpp. */
int bhrt_scene_clone(const bhrt_scene *scene, bhrt_scene **out);
void bhrt_scene_free(bhrt_scene *scene);
int bhrt_scene_info(const bhrt_scene *scene, bhrt_info *info);
int bhrt_scene_warning(const bhrt_scene *scene, uint32_t i, const char **text); /* the reference printf()s these */
int bhrt_scene_flat(const bhrt_scene *scene, const void **blob, uint64_t *bytes); /* host copy of the HBM image (include/bhrt_flat.h) */

/* ---- device residency ---------------------------------------------------------------------------- */
int bhrt_scene_upload(bhrt_scene *scene, int device); /* copies the flat scene into HBM of `device`; idempotent */
/* knobs of an uploaded scene: they steer which internal path a render takes, never its result.  Test knobs "frame_cap", "gather_lane_budget" (0 = off) and
 * "gather_stats"; "shadow_overlap" (default 1; 0 = the any-hit kernels of a wave step run in front of the next step on the pass's own stream instead of
 * beside it on a second one: the kernel groups timed alone, bench.py's `frac_alone`).
 * The library reads its development switches (BHRT_STREAM_WAVES, BHRT_FUSED_CAMERA, BHRT_NO_SLOW_QUEUE, BHRT_DEBUG_*, BHRT_PHOTON_BALANCE_HOST,
 * BHRT_SHADOW_OVERLAP) from the environment once, at upload; the test knobs are not reachable from the environment at all. */
int bhrt_scene_knob(bhrt_scene *scene, const char *name, int value);
int bhrt_device_count(int *n);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* recursive(&rootNode, ray, hit, bHit, hitSide) for n rays (Main.cpp:389-413).
 * rays_soa: 6 arrays of n floats back to back (ox[n], oy[n], oz[n], dx[n], dy[n], dz[n]).
 * *_host variants take host pointers (copies included); *_dev take device pointers + a hipStream_t. */
int bhrt_trace_closest_host(bhrt_scene *scene, const float *rays_soa, int hit_side, size_t n, bhrt_hits out);
int bhrt_trace_closest_dev(bhrt_scene *scene, const float *d_rays_soa, int hit_side, size_t n, bhrt_hits d_out, void *stream);
/* GenLight::Shadow(ray, t_max) (Lights/GenLight.cpp:10-13): vis = 0 occluded / 1 visible */
int bhrt_trace_shadow_host(bhrt_scene *scene, const float *rays_soa, const float *tmax, size_t n, float *vis);
int bhrt_trace_shadow_dev(bhrt_scene *scene, const float *d_rays_soa, const float *d_tmax, size_t n, float *d_vis, void *stream);

/* BeginRender() (Main.cpp:178-242) without the UI and without the PNG write: renders this rank's tiles.
 * rgb8: W*H*3 bytes, row-major j*W+i like RenderImage::GetPixels (may be NULL);
 * radiance: W*H*3 floats, the per-pixel average BEFORE gamma (may be NULL).
 * Pixels of tiles owned by other ranks are left untouched.  Host pointers. */
int bhrt_render(bhrt_scene *scene, const bhrt_opts *opts, uint8_t *rgb8, float *radiance, bhrt_stats *stats);
/* Pinned (page-locked) host memory for the frame buffers handed to bhrt_render: the device-to-host copy then runs at the full PCIe
 * rate (a pageable destination: about half).  Any host pointer works; this is only faster. */
int bhrt_host_alloc(void **ptr, size_t bytes);
void bhrt_host_free(void *ptr);
/* same, but the outputs stay in HBM (device pointers; for timing and for the RCCL gather) */
int bhrt_render_dev(bhrt_scene *scene, const bhrt_opts *opts, uint8_t *d_rgb8, float *d_radiance, bhrt_stats *stats, void *stream);
/* per-sample radiance for a pixel region, keyed RNG (parity tests): out = region_pixels*spp*3 floats, host */
int bhrt_render_samples(bhrt_scene *scene, const bhrt_opts *opts, int x0, int y0, int x1, int y1, float *samples, bhrt_stats *stats);

/* ---- caustic photon map (Main.cpp:342-386, DataStructure/cyPhotonMap.h) -------------------------- */
int bhrt_photon_build(bhrt_scene *scene, const bhrt_opts *opts, uint32_t max_photons, uint32_t *n_stored);
/* Multi-GPU build of the caustic map: the emission loop of BuildCausticPhotonMap (Main.cpp:342-386) draws from
 * a stream keyed by the emission index, so ranks can run disjoint index ranges.  bhrt_photon_emit_range runs emissions
 * [e0, e0 + count) (count a multiple of 256) and returns the photons they store, in emission order, with unscaled power
 * (24-byte records; photons_out may be a host or a device pointer — the records of a multi-GPU build never need to touch the host;
 * *n_photons = how many, also when that is more than capacity: BHRT_ERR_ARG then, call again with room for them).  bhrt_photon_install takes records in
 * emission order (after the exchange: the first MAX_CausticPhotonCount of all ranks' records), applies ScalePhotonPowers(1/n)
 * (Main.cpp:380), balances and installs the map for bhrt_render* — the same map bhrt_photon_build makes alone. */
int bhrt_photon_emit_range(bhrt_scene *scene, const bhrt_opts *opts, int global_map, uint64_t e0, uint32_t count, void *photons_out, uint32_t capacity,
                           uint32_t *n_photons);
int bhrt_photon_install(bhrt_scene *scene, const void *records_emission_order /* host or device pointer */, uint32_t n);
/* The reference's second map, BuildPhotonMap (Main.cpp:251-295; TracePhotonRay Main.cpp:296-317, RandomPhotonBounce
 * MtlBlinn.cpp:140-202): photons that survive diffuse and specular bounces.  Its only call is commented out in the reference
 * (Main.cpp:196) and nothing gathers from it, so it is built on request and handed back: balanced 24-byte records into
 * photons_out (capacity records; may be NULL) and / or written like Resource/photonmap.dat (dat_path, may be NULL). */
int bhrt_photon_build_global(bhrt_scene *scene, const bhrt_opts *opts, uint32_t max_photons, void *photons_out, uint32_t capacity, uint32_t *n_stored,
                             const char *dat_path);
int bhrt_photon_gather_host(bhrt_scene *scene, const float *p, const float *n, size_t cnt, float radius, float *irrad, float *dir);
/* same with the choice of bhrt_opts.photon_exact, and (test hook, any pointer may be NULL) the photons the estimate used: knn = cnt x 1000
 * indices into the balanced map (1-based, unsorted, unused slots 0), knn_count = how many, d2max = np.dist2[0] at the end of LocatePhotons.
 * Filled for the queries the selection pass answers (>= 1000 photons in the radius, photon_exact = 0); knn_count = 0 otherwise. */
int bhrt_photon_gather_host_ex(bhrt_scene *scene, const float *p, const float *n, size_t cnt, float radius, int photon_exact, float *irrad, float *dir,
                               uint32_t *knn, uint32_t *knn_count, float *d2max);
int bhrt_photon_get(const bhrt_scene *scene, void *photons_out /* 24 B records, balanced order */, uint32_t capacity, uint32_t *n);
int bhrt_photon_export(const bhrt_scene *scene, const char *dat_path); /* 24-byte records, Main.cpp:383-385 */
/* Loads a map written by bhrt_photon_export or by the reference (Resource/causticPhotonMap.dat) instead of building it.
 * rebalance = 1: PhotonMap::InitializePhotonMapByFile (cyPhotonMap.h:409-417), which balances the records again;
 * rebalance = 0: the records are used in the order of the file (a balanced map as exported: the cached photon pass). */
int bhrt_photon_import(bhrt_scene *scene, const char *dat_path, int rebalance);

/* ---- multi-GPU framebuffer exchange (no counterpart in the single-process reference) ------------------
 * Tile t (row-major over ceil(W/tile) x ceil(H/tile) tiles) belongs to rank t mod world, the same rule bhrt_render*
 * applies through bhrt_opts.rank / world_size / tile_size.  A rank packs the tiles it rendered into ONE block
 * (float radiance section, then RGB8 section); one all-gather of the blocks (RCCL) and one unpack give every rank
 * the whole image.  Device pointers; `stream` is a hipStream_t (NULL = the default stream), the calls do not synchronise. */
size_t bhrt_tiles_block_bytes(int width, int height, int tile, int world); /* bytes of one rank's block */
int bhrt_tiles_pack_dev(const uint8_t *d_rgb8, const float *d_radiance, int width, int height, int tile, int rank, int world, void *d_block, void *stream);
int bhrt_tiles_unpack_dev(const void *d_blocks /* world blocks, rank-major */, int width, int height, int tile, int world, uint8_t *d_rgb8, float *d_radiance,
                          void *stream);

/* ---- images beside the colour image ------------------------------------------
 * First hit of the un-jittered camera ray of every pixel (the pixel corner: `1 / 2 == 0`, Main.cpp:145), row-major:
 *   z       W*H floats      HitInfo::z, BIGFLOAT where nothing is hit = RenderImage::GetZBuffer() (scene.h:532; its store is
 *                           commented out at Main.cpp:231)
 *   normal  W*H*3 floats    HitInfo::N in world space (zero on a miss)   \ the optional "normal" / "albedo" images of
 *   albedo  W*H*3 floats    diffuse.Sample(uvw, duvw) of the hit material / DenoiseImage (Main.cpp:70-71, commented out there)
 * Any pointer may be NULL. */
int bhrt_first_hit_dev(bhrt_scene *scene, float *d_z, float *d_normal, float *d_albedo, void *stream);
int bhrt_first_hit(bhrt_scene *scene, float *z, float *normal, float *albedo);
/* RenderImage::ComputeZBufferImage (scene.h:578-600): 8-bit depth image, 0 where nothing is hit.  Device pointers. */
int bhrt_zbuffer_image_dev(bhrt_scene *scene, const float *d_z, size_t n, uint8_t *d_img, void *stream);
/* colorArray of BeginRender (Main.cpp:202,219-229): pow(colour, 1/2.2f) as floats — the "color" image DenoiseImage is given
 * (Main.cpp:60-69).  d_radiance = the radiance image of bhrt_render_dev. */
int bhrt_color_image_dev(bhrt_scene *scene, const float *d_radiance, size_t n_pixels, int gamma, float *d_color, void *stream);

/* ---- test hook: csrc/bhrt_detmath.h evaluated on the device, to prove host and device produce the same bits.
 * fn: 0 sin 1 cos 2 tan 3 acos 4 asin 5 atan2(a,b) 6 pow(a,b) 7 rand_to_unit(bits of a) 8 a/b 9 sqrt(a); host pointers */
int bhrt_math_eval_dev(int fn, const float *a, const float *b, size_t n, float *out);

/* ---- image output (RenderImage::SaveImage, Scenes/scene.h:628-644) ------------------------------- */
int bhrt_save_png(const char *path, const uint8_t *rgb8, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* BHRT_H */
