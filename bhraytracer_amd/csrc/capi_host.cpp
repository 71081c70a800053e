// capi_host.cpp — host-only entry points of include/bhrt.h: scene loading (the LoadScene seam,
// xmlload.cpp:65), scene introspection and PNG output (RenderImage::SaveImage, scene.h:628).
#include <string.h>

#include <exception>
#include <new>

#include "png_io.h"
#include "scene_internal.h"

namespace bhrt {
static thread_local std::string g_error;
void SetError(const std::string &msg) { g_error = msg; }
int AbiException()
{
    try { throw; }
    catch (const std::bad_alloc &) { SetError("out of host memory"); return BHRT_ERR_IO; }
    catch (const std::exception &e) { SetError(std::string("internal error: ") + e.what()); return BHRT_ERR_UNSUPPORTED; }
    catch (...) { SetError("internal error"); return BHRT_ERR_UNSUPPORTED; }
}
} // namespace bhrt

extern "C" {

const char *bhrt_last_error(void) { return bhrt::g_error.c_str(); }

void bhrt_default_opts(bhrt_opts *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->spp = 32;              // Main.cpp:141
    o->gi_bounces = 3;        // Main.cpp:130
    o->internal_bounces = 16; // Main.cpp:41
    o->seed = 0;
    o->jitter = 1;
    o->gamma = 1; // Main.cpp:128
    o->photon_map = 0; // Main.cpp:51 (commented out in the reference)
    o->rank = 0;
    o->world_size = 1;
    o->tile_size = 32;
    o->samples_per_pass = 0;
}

int bhrt_scene_load_xml(const char *path, bhrt_scene **out) { return bhrt_scene_load_xml_ex(path, -1, out); }

int bhrt_scene_load_xml_ex(const char *path, int bvh_device, bhrt_scene **out)
try {
    if (!path || !out) { bhrt::SetError("bhrt_scene_load_xml: null argument"); return BHRT_ERR_ARG; }
    *out = nullptr;
    bhrt_scene *s = new bhrt_scene;
    std::string err;
    int rc = bhrt::LoadSceneXml(path, s->flat, err, bvh_device);
    if (rc) {
        bhrt::SetError(err);
        delete s;
        return rc == 1 ? BHRT_ERR_IO : (rc == 7 ? BHRT_ERR_HIP : BHRT_ERR_PARSE);
    }
    const bhrt_flat_header *H = s->flat.hdr();
    const bhrt_mesh *m = (const bhrt_mesh *)(s->flat.blob.data() + H->off_meshes);
    for (uint32_t i = 0; i < H->n_meshes; i++) {
        s->n_triangles += m[i].nf;
        s->n_bvh_nodes += m[i].n_bvh_nodes;
        if (m[i].bvh_depth > s->max_bvh_depth) s->max_bvh_depth = m[i].bvh_depth;
    }
    *out = s;
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_scene_clone(const bhrt_scene *src, bhrt_scene **out)
try {
    if (!src || !out) { bhrt::SetError("bhrt_scene_clone: null argument"); return BHRT_ERR_ARG; }
    bhrt_scene *s = new bhrt_scene;
    s->flat = src->flat; // the host copy only: the clone is uploaded to a device of its own (one handle per GPU)
    s->n_triangles = src->n_triangles; s->n_bvh_nodes = src->n_bvh_nodes; s->max_bvh_depth = src->max_bvh_depth;
    *out = s;
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

void bhrt_scene_free(bhrt_scene *s)
{
    if (!s) return;
    if (s->dev) bhrt::DestroyDeviceState(s->dev);
    delete s;
}

int bhrt_scene_info(const bhrt_scene *s, bhrt_info *info)
try {
    if (!s || !info) { bhrt::SetError("bhrt_scene_info: null argument"); return BHRT_ERR_ARG; }
    const bhrt_flat_header *H = s->flat.hdr();
    memset(info, 0, sizeof *info);
    info->width = H->camera.width; info->height = H->camera.height;
    info->n_nodes = H->n_nodes; info->n_meshes = H->n_meshes; info->n_triangles = s->n_triangles;
    info->n_bvh_nodes = s->n_bvh_nodes; info->n_materials = H->n_materials; info->n_lights = H->n_lights;
    info->n_textures = H->n_textures; info->max_node_depth = H->max_node_depth; info->max_bvh_depth = s->max_bvh_depth;
    info->flat_bytes = H->total_bytes; info->n_warnings = (uint32_t)s->flat.warnings.size();
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_scene_warning(const bhrt_scene *s, uint32_t i, const char **text)
try {
    if (!s || !text || i >= s->flat.warnings.size()) { bhrt::SetError("bhrt_scene_warning: bad index"); return BHRT_ERR_ARG; }
    *text = s->flat.warnings[i].c_str();
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_scene_flat(const bhrt_scene *s, const void **blob, uint64_t *bytes)
try {
    if (!s || !blob || !bytes) { bhrt::SetError("bhrt_scene_flat: null argument"); return BHRT_ERR_ARG; }
    *blob = s->flat.blob.data();
    *bytes = s->flat.blob.size();
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

int bhrt_save_png(const char *path, const uint8_t *rgb8, int w, int h)
try {
    if (!path || !rgb8) { bhrt::SetError("bhrt_save_png: null argument"); return BHRT_ERR_ARG; }
    if (!bhrt::SavePng(path, rgb8, w, h, 3)) { bhrt::SetError(std::string("cannot write ") + path); return BHRT_ERR_IO; }
    return BHRT_OK;
} catch (...) { return bhrt::AbiException(); }

} // extern "C"
