/* bhrt_flat.h — the flattened scene: ONE relocatable blob that is (a) what the host
 * front-end produces from an XML scene, (b) exactly the bytes uploaded to HBM for the
 * HIP kernels, (c) what the CPU oracle evaluates.  All cross references are indices /
 * byte offsets from the start of the blob, so the blob can be memcpy'd anywhere.
 *
 * Mirrors the reference's data model (all citations relative to /root/reference/BHRayTracer):
 *   bhrt_node      <- Node + Transformation           Scenes/scene.h:208-246,426-502
 *   bhrt_mesh      <- TriObj / cyTriMesh / cyBVH      Objects/objects.h:46-67, Objects/TriObj/cyTriMesh.h:106-122,
 *                                                     DataStructure/cyBVH.h:187-200
 *   bhrt_material  <- MtlBlinn / MultiMtl             Materials/materials.h:20-83
 *   bhrt_light     <- Ambient/Direct/PointLight       Lights/lights.h:29-87
 *   bhrt_texmap    <- TextureMap, bhrt_texture <- TextureFile/TextureChecker  Scenes/scene.h:364-386, Textures/texture.h
 *   bhrt_camera    <- Camera + the frame BeginRender derives   Scenes/scene.h:506-524, Main.cpp:179-192
 *
 * Plain C, fixed-width types, 4-byte alignment everywhere.
 */
#ifndef BHRT_FLAT_H
#define BHRT_FLAT_H

#include <stdint.h>

#define BHRT_FLAT_MAGIC 0x54524842u /* "BHRT" */
#define BHRT_FLAT_VERSION 10u
#define BHRT_MAX_NODE_DEPTH 8 /* scene-graph depth below the root the kernels support */
#define BHRT_BIGFLOAT 1.0e30f  /* Scenes/scene.h:38 */

enum { BHRT_OBJ_NONE = 0, BHRT_OBJ_SPHERE = 1, BHRT_OBJ_PLANE = 2, BHRT_OBJ_MESH = 3 };
enum { BHRT_LIGHT_AMBIENT = 0, BHRT_LIGHT_DIRECT = 1, BHRT_LIGHT_POINT = 2 };
enum { BHRT_MTL_BLINN = 0, BHRT_MTL_WHITE = 1 /* empty MultiMtl: materials.h:71 */, BHRT_MTL_NONE = 2 /* node without material */ };
enum { BHRT_TEX_CHECKER = 0, BHRT_TEX_FILE = 1 };
enum { BHRT_HIT_NONE = 0, BHRT_HIT_FRONT = 1, BHRT_HIT_BACK = 2, BHRT_HIT_FRONT_AND_BACK = 3 }; /* scene.h:57-60 */

/* Transformation (scene.h:208-246): tm, pos, cached inverse itm; matrices column-major
 * like cyMatrix3 (cyMatrix.h:399-406): cell[col*3+row]. */
typedef struct bhrt_xform {
    float tm[9];
    float pos[3];
    float itm[9];
} bhrt_xform; /* 84 B */

/* One scene-graph node, in DFS pre-order of the reference's Node tree (root excluded). */
typedef struct bhrt_node {
    bhrt_xform xf;
    int32_t parent;    /* index of the parent node, -1 = child of rootNode */
    int32_t depth;     /* 1 = child of rootNode */
    int32_t obj_type;  /* BHRT_OBJ_* */
    int32_t mesh;      /* index into meshes[] when obj_type == MESH */
    int32_t material;  /* index into materials[], -1 = none */
    int32_t subtree_end; /* index one past the last descendant (pre-order) */
    int32_t pad[5];
} bhrt_node; /* 128 B */

/* BVH node: cyBVH's 28-byte node (6 bounds + data word, cyBVH.h:187-200) plus the parent
 * link the stackless traversal walks up with.  data: bit31 leaf; leaf: bits 28-30 = count-1,
 * bits 0-27 = element offset; inner: index of first child (second = first+1). Root id = 1. */
typedef struct bhrt_bvh_node {
    float b[6];
    uint32_t data;
    uint32_t parent;
} bhrt_bvh_node; /* 32 B */

/* Pre-gathered triangle: what TriObj::IntersectTriangle recomputes on every test, formed once on the host with the same float
 * operations, so the bits equal the reference's — vN = (v1-v0)x(v2-v0), |vN|, vN.v0 (TriObj.cpp:79-89) and the three vertices
 * already projected on the coordinate plane that the test picks from |vN| (TriObj.cpp:105-131; the choice is a function of
 * the triangle alone): axis 0 = (y, z), 1 = (x, z), 2 = (x, y), 3 = none of the three comparisons holds (NaN normal: the
 * reference then works on zeros).  48 B */
typedef struct bhrt_tri {
    /* first 20 bytes: what the plane part of the test reads (every triangle of a visited leaf); the rest only for a ray
     * whose plane hit lies in range */
    float vN[3];
    float vN_dot_v0;
    float vN_len;
    uint32_t face_axis; /* triangle id (index into f[]) | axis << 30 */
    float p0[2], p1[2], p2[2];
} bhrt_tri;

typedef struct bhrt_mesh {
    uint32_t nv, nf, nvn, nvt;
    uint32_t n_bvh_nodes; /* array length incl. unused slot 0 */
    uint32_t bvh_depth;   /* max depth of a node below the root (root = 0) */
    /* byte offsets from blob start */
    uint64_t off_v, off_vn, off_vt;    /* float[3] each */
    uint64_t off_f, off_fn, off_ft;    /* uint32[3] each (fn/ft always present: nvn/nvt > 0 is enforced) */
    uint64_t off_bvh;                  /* bhrt_bvh_node[n_bvh_nodes] */
    uint64_t off_elems;                /* uint32[nf] */
    uint64_t off_tris;                 /* bhrt_tri[nf], indexed by triangle id */
    uint64_t off_dbvh;                 /* bhrt_bvh_node[n_bvh_nodes]: the SAME tree renumbered breadth-first (root 1, children still adjacent,
                                          first child even), so the top levels are the lowest ids -> stageable in LDS.  Traversal uses this copy;
                                          off_bvh keeps cyBVH's own numbering (what the reference's node array looks like). */
    uint64_t off_leaf_tris;            /* bhrt_tri[nf] in BVH leaf order: entry k = triangle elems[k] (one load instead of two dependent ones) */
    float bound_min[3], bound_max[3];  /* cyTriMesh::ComputeBoundingBox */
    uint32_t bvh_nested;               /* 1 = every child box lies inside its parent's box, float for float (true of every tree cyBVH::Build makes:
                                          boxes are min / max over the same vertex coordinates).  Box::IntersectRay (Box.cpp:3-46) is monotone in the
                                          box bounds, rounding included, so a ray that misses a box misses every box nested in it: TraceBVHNode's
                                          visit of a box-missed INNER sibling (TriObj.cpp:245-248,263-266) then ends at that node's own two box
                                          tests, and the traversal kernels may leave it out (device_trace.h) */
    uint32_t pad0;
    uint64_t off_dparent;              /* uint32[n_bvh_nodes]: parent links of the breadth-first copy (the parent-link walks read them here) */
    /* Skip of box-missed LEAF siblings ("leaf skip"; device_trace.h::leaf_skip).  In the breadth-first copy the `parent` word of a
     * LEAF holds the leaf's packed normal cone instead (0 = the leaf is never skipped); these four are the mesh-wide constants of the test:
     * a ray with max |o_i| <= skip_omax whose line misses the leaf box inflated by skip_k0 + skip_k1 * max|o_i|, passes through the box
     * inflated by skip_big, and makes an angle with every triangle plane of the leaf that the cone bounds away from grazing cannot be
     * accepted by IntersectTriangle (TriObj.cpp:68-189) for any triangle of the leaf.  skip_omax = 0: no leaf of this mesh qualifies. */
    float skip_k0, skip_k1, skip_big, skip_omax;
} bhrt_mesh;

/* TextureMap = Transformation + texture (scene.h:364-386) */
typedef struct bhrt_texmap {
    bhrt_xform xf;
    int32_t texture; /* index into textures[], -1 = map exists but texture failed to load -> samples black */
} bhrt_texmap;

typedef struct bhrt_texture {
    int32_t type; /* BHRT_TEX_* */
    int32_t width, height;
    float color1[3], color2[3]; /* checker */
    uint64_t off_data;          /* RGB8 texels, row-major */
} bhrt_texture;

/* TexturedColor (scene.h:394-422) */
typedef struct bhrt_texcolor {
    float color[3];
    int32_t map; /* index into texmaps[], -1 = plain colour */
} bhrt_texcolor;

typedef struct bhrt_material {
    int32_t kind; /* BHRT_MTL_* */
    bhrt_texcolor diffuse, specular, refraction;
    float glossiness;
    float absorption[3];
    float ior;
    float refraction_glossiness; /* member refractionGlossiness (photon bounce only) */
} bhrt_material;

typedef struct bhrt_light {
    int32_t type; /* BHRT_LIGHT_* */
    float intensity[3];
    float vec[3]; /* position (point) or normalised direction (direct) */
    float size;   /* point light radius (float); GetSize() truncates to int, lights.h:76 */
} bhrt_light;

typedef struct bhrt_camera {
    float pos[3], dir[3], up[3];
    float fov, focaldist, dof;
    int32_t width, height;
    /* frame derived in BeginRender (Main.cpp:179-192) */
    float top_left[3], dd_x[3], dd_y[3];
} bhrt_camera;

typedef struct bhrt_flat_header {
    uint32_t magic, version;
    uint64_t total_bytes;
    uint32_t n_nodes, n_meshes, n_materials, n_lights, n_texmaps, n_textures;
    uint64_t off_nodes, off_meshes, off_materials, off_lights, off_texmaps, off_textures;
    bhrt_camera camera;
    bhrt_texcolor background, environment;
    float all_light_intensity; /* Main.cpp:116-123: sum of Gray() over the sorted lights */
    uint32_t max_node_depth;
    uint32_t reserved[6];
} bhrt_flat_header;

#endif /* BHRT_FLAT_H */
