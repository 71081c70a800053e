// ref_harness.cpp — drives the UNMODIFIED reference (compiled in place from
// /root/reference, see oracle/Makefile) so that its own functions produce the golden
// vectors this repo's oracle and HIP path are pinned against.
//
// TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or linked into the product; the
// binary is written to oracle/_ref/ (git-ignored) and exists only in the development
// container (the reference cannot travel to the GPU box).
//
// How it reaches the reference: Main.cpp is included
// textually with main renamed, which exposes LoadScene, recursive(), the camera
// globals, CalculateLightsIntensity(), RandomPositionInPixel() and every
// Material::Shade.  libc rand() is interposed by the counter stream of
// include/bhrt_rng.h (sequential mode), reset per (pixel, sample).
//
// What this file restates (because the reference has it inline in BeginRender /
// PathTracing and it cannot be called separately): the camera frame
// (Main.cpp:179-192), the per-sample body of PathTracing (Main.cpp:145-168), the
// average / gamma / Color24 store (Main.cpp:170,220-230).  Everything else is the
// reference's own code.  The command `beginrender` restates nothing: it calls the
// reference's own BeginRender() (and, in the -DUSE_PhotonMap build, through it the
// reference's own BuildCausticPhotonMap()) and dumps RenderImage::GetPixels(); the
// oracle's whole-program mode (oracle_begin_render) is pinned against that, so the
// restated pieces above are themselves checked against the real functions.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <string>
#include <map>
#include <atomic>
#include <algorithm>
#include <iostream>
#include <sstream>
#include <limits>
#include <cassert>
#include <cstdint>
#include <omp.h>
#include <xmmintrin.h>
#include <emmintrin.h>

#include "../../include/bhrt_rng.h"

// read-only access to private state for the scene dump (layout is unaffected)
#define private public
#define protected public
#include "Scenes/scene.h"
#include "Objects/objects.h"
#include "Lights/lights.h"
#include "Materials/materials.h"
#include "Textures/texture.h"
#include "cyPhotonMap.h"
#undef private
#undef protected

#define main bhrt_reference_main
#include "Main.cpp"
#undef main

// ---------------------------------------------------------------- rand() interposition
static uint32_t g_key = 0, g_ctr = 0;
static unsigned long long g_draws = 0;
extern "C" int rand(void) { g_draws++; return bhrt_rand31(g_key, g_ctr++); }

// ---------------------------------------------------------------- helpers
static std::vector<const Node *> g_nodes; // DFS pre-order, root excluded
static std::map<const Node *, int> g_nodeIndex;
static std::vector<int> g_parent, g_depth;

static void Flatten(const Node *n, int parent, int depth)
{
    for (int i = 0; i < n->GetNumChild(); i++) {
        const Node *c = n->GetChild(i);
        int idx = (int)g_nodes.size();
        g_nodes.push_back(c);
        g_nodeIndex[c] = idx;
        g_parent.push_back(parent);
        g_depth.push_back(depth);
        Flatten(c, idx, depth + 1);
    }
}

template <class T> static void WriteFile(const std::string &path, const std::vector<T> &v)
{
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), fp);
    fclose(fp);
}
template <class T> static std::vector<T> ReadFile(const std::string &path)
{
    std::vector<T> v;
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(2); }
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    v.resize(n / sizeof(T));
    if (n) { size_t got = fread(v.data(), 1, n, fp); (void)got; }
    fclose(fp);
    return v;
}

static void SetupCameraFrame() // Main.cpp:179-192
{
    float aor = camera.imgWidth / (float)camera.imgHeight;
    float tan_h_pov = tan(camera.fov / 2 * PI / 180.0);
    float l = camera.focaldist;
    float h = 2 * l * tan_h_pov;
    float w = aor * h;
    camZAxis = -camera.dir;
    camYAxis = camera.up;
    camXAxis = camYAxis.Cross(camZAxis);
    topLeft = camera.pos - camZAxis * l + camYAxis * h / 2 - camXAxis * w / 2;
    dd_x = camXAxis * w / camera.imgWidth;
    dd_y = camYAxis * h / camera.imgHeight;
}

static void PushHit(std::vector<int> &oi, std::vector<float> &of, bool bHit, const HitInfo &h)
{
    oi.push_back(bHit && h.node ? g_nodeIndex[h.node] : -1);
    oi.push_back(h.front ? 1 : 0);
    of.push_back(h.z);
    of.push_back(h.p.x); of.push_back(h.p.y); of.push_back(h.p.z);
    of.push_back(h.N.x); of.push_back(h.N.y); of.push_back(h.N.z);
    of.push_back(h.uvw.x); of.push_back(h.uvw.y); of.push_back(h.uvw.z);
    for (int k = 0; k < 2; k++) { of.push_back(h.duvw[k].x); of.push_back(h.duvw[k].y); of.push_back(h.duvw[k].z); }
}

static int MaterialIndex(const Material *m)
{
    if (!m) return -1;
    for (size_t i = 0; i < materials.size(); i++) if (materials[i] == m) return (int)i;
    return -2;
}

static void DumpTexturedColor(std::vector<float> &f, const TexturedColor &tc)
{
    Color c = tc.GetColor();
    f.push_back(c.r); f.push_back(c.g); f.push_back(c.b);
    f.push_back(tc.GetTexture() ? 1.f : 0.f);
}

static void DumpScene(const std::string &prefix)
{
    std::vector<int> ni;
    std::vector<float> nf;
    std::vector<const TriObj *> meshes;
    for (size_t k = 0; k < g_nodes.size(); k++) {
        const Node *n = g_nodes[k];
        int type = 0, meshid = -1;
        const Object *o = n->GetNodeObj();
        if (o) {
            if (dynamic_cast<const Sphere *>(o)) type = 1;
            else if (dynamic_cast<const Plane *>(o)) type = 2;
            else if (const TriObj *t = dynamic_cast<const TriObj *>(o)) {
                type = 3;
                size_t m = 0;
                for (; m < meshes.size(); m++) if (meshes[m] == t) break;
                if (m == meshes.size()) meshes.push_back(t);
                meshid = (int)m;
            }
        }
        ni.push_back(g_parent[k]); ni.push_back(g_depth[k]); ni.push_back(type); ni.push_back(meshid);
        ni.push_back(MaterialIndex(n->GetMaterial()));
        const Matrix3f &tm = n->GetTransform();
        const Matrix3f &itm = n->GetInverseTransform();
        for (int i = 0; i < 9; i++) nf.push_back(tm.cell[i]);
        nf.push_back(n->GetPosition().x); nf.push_back(n->GetPosition().y); nf.push_back(n->GetPosition().z);
        for (int i = 0; i < 9; i++) nf.push_back(itm.cell[i]);
    }
    WriteFile(prefix + ".nodes_i32", ni);
    WriteFile(prefix + ".nodes_f32", nf);

    std::vector<float> cam = {camera.pos.x, camera.pos.y, camera.pos.z, camera.dir.x, camera.dir.y, camera.dir.z,
                              camera.up.x, camera.up.y, camera.up.z, camera.fov, camera.focaldist,
                              (float)camera.imgWidth, (float)camera.imgHeight,
                              topLeft.x, topLeft.y, topLeft.z, dd_x.x, dd_x.y, dd_x.z, dd_y.x, dd_y.y, dd_y.z};
    WriteFile(prefix + ".camera_f32", cam);

    std::vector<float> lf;
    for (size_t i = 0; i < lights.size(); i++) {
        Light *l = lights[i];
        if (AmbientLight *a = dynamic_cast<AmbientLight *>(l)) {
            lf.push_back(0); lf.push_back(a->intensity.r); lf.push_back(a->intensity.g); lf.push_back(a->intensity.b);
            lf.push_back(0); lf.push_back(0); lf.push_back(0); lf.push_back(0);
        } else if (DirectLight *d = dynamic_cast<DirectLight *>(l)) {
            lf.push_back(1); lf.push_back(d->intensity.r); lf.push_back(d->intensity.g); lf.push_back(d->intensity.b);
            lf.push_back(d->direction.x); lf.push_back(d->direction.y); lf.push_back(d->direction.z); lf.push_back(0);
        } else if (PointLight *p = dynamic_cast<PointLight *>(l)) {
            lf.push_back(2); lf.push_back(p->intensity.r); lf.push_back(p->intensity.g); lf.push_back(p->intensity.b);
            lf.push_back(p->position.x); lf.push_back(p->position.y); lf.push_back(p->position.z); lf.push_back(p->size);
        }
    }
    lf.push_back(allLightIntensity);
    WriteFile(prefix + ".lights_f32", lf);

    std::vector<float> mf;
    for (size_t i = 0; i < materials.size(); i++) {
        MtlBlinn *m = dynamic_cast<MtlBlinn *>(materials[i]);
        if (!m) { for (int k = 0; k < 26; k++) mf.push_back(-1.f); continue; }
        DumpTexturedColor(mf, m->diffuse);
        DumpTexturedColor(mf, m->specular);
        DumpTexturedColor(mf, m->refraction);
        DumpTexturedColor(mf, m->reflection);
        DumpTexturedColor(mf, m->emission);
        mf.push_back(m->glossiness);
        mf.push_back(m->absorption.r); mf.push_back(m->absorption.g); mf.push_back(m->absorption.b);
        mf.push_back(m->ior);
        mf.push_back(m->refractionGlossiness);
    }
    WriteFile(prefix + ".materials_f32", mf);

    for (size_t m = 0; m < meshes.size(); m++) {
        const TriObj *t = meshes[m];
        char tag[64];
        snprintf(tag, sizeof tag, ".mesh%d", (int)m);
        std::string mp = prefix + tag;
        std::vector<float> v, vn, vt;
        std::vector<unsigned> f, fn, ft;
        for (unsigned i = 0; i < t->NV(); i++) { v.push_back(t->V(i).x); v.push_back(t->V(i).y); v.push_back(t->V(i).z); }
        for (unsigned i = 0; i < t->NVN(); i++) { vn.push_back(t->VN(i).x); vn.push_back(t->VN(i).y); vn.push_back(t->VN(i).z); }
        for (unsigned i = 0; i < t->NVT(); i++) { vt.push_back(t->VT(i).x); vt.push_back(t->VT(i).y); vt.push_back(t->VT(i).z); }
        for (unsigned i = 0; i < t->NF(); i++) {
            for (int k = 0; k < 3; k++) f.push_back(t->F(i).v[k]);
            if (t->fn) for (int k = 0; k < 3; k++) fn.push_back(t->FN(i).v[k]);
            if (t->ft) for (int k = 0; k < 3; k++) ft.push_back(t->FT(i).v[k]);
        }
        WriteFile(mp + ".v_f32", v); WriteFile(mp + ".vn_f32", vn); WriteFile(mp + ".vt_f32", vt);
        WriteFile(mp + ".f_u32", f); WriteFile(mp + ".fn_u32", fn); WriteFile(mp + ".ft_u32", ft);
        // BVH: walk the node array from the root (id 1) to find its extent
        const cyBVHTriMesh &bvh = t->bvh;
        unsigned maxId = 1;
        std::vector<unsigned> stack = {1};
        while (!stack.empty()) {
            unsigned id = stack.back(); stack.pop_back();
            if (id > maxId) maxId = id;
            if (!bvh.IsLeafNode(id)) { stack.push_back(bvh.GetFirstChildNode(id)); stack.push_back(bvh.GetSecondChildNode(id)); }
        }
        std::vector<float> bb((maxId + 1) * 6, 0.f);
        std::vector<unsigned> bd(maxId + 1, 0u);
        for (unsigned id = 1; id <= maxId; id++) {
            const float *b = bvh.GetNodeBounds(id);
            for (int k = 0; k < 6; k++) bb[id * 6 + k] = b[k];
            bd[id] = bvh.nodes[id].data;
        }
        std::vector<unsigned> el(bvh.elements, bvh.elements + t->NF());
        WriteFile(mp + ".bvh_f32", bb); WriteFile(mp + ".bvh_u32", bd); WriteFile(mp + ".elems_u32", el);
        std::vector<float> bounds = {t->GetBoundMin().x, t->GetBoundMin().y, t->GetBoundMin().z,
                                     t->GetBoundMax().x, t->GetBoundMax().y, t->GetBoundMax().z};
        WriteFile(mp + ".bounds_f32", bounds);
    }
}

#ifdef USE_PhotonMap
// BuildCausticPhotonMap (Main.cpp:342-386) with the photon budget as a run-time value (the reference's
// MAX_CausticPhotonCount is an unguarded #define) and without the .dat write.  Everything it calls is the reference's own
// code: ComparePointLight, PointLight::GetProbability/RandomPhoton/GetPhotonIntensity, TraceCausticPhotonRay,
// PhotonMap::Resize/NumPhotons/ScalePhotonPowers/PrepareForIrradianceEstimation.
static unsigned long long g_emitted = 0;
static std::vector<unsigned char> g_unbalanced;
static bool BuildCausticPhotonMapN(int maxPhotons)
{
    causticPhotonMap = new PhotonMap();
    causticPhotonMap->Resize(maxPhotons);
    memset((void *)&causticPhotonMap->photons[0], 0, sizeof(cyPhotonMap::Photon) * (maxPhotons + 1)); // defined bytes for slot 0 / plane bits
    std::vector<PointLight *> pointLightList;
    for (auto it = lights.begin(); it != lights.end(); ++it)
        if (PointLight *ptr = reinterpret_cast<PointLight *>(*it)) pointLightList.push_back(ptr);
    if (pointLightList.size() == 0) return false;
    sort(pointLightList.begin(), pointLightList.end(), ComparePointLight);
    float sumOfPointLight = 0;
    for (int i = 0; i < pointLightList.size(); ++i) sumOfPointLight += pointLightList[i]->GetIntensity() * pointLightList[i]->GetSize();
    while (causticPhotonMap->NumPhotons() < maxPhotons) {
        float rnd = Rnd01();
        int i = 0;
        while (rnd > pointLightList[i]->GetProbability(sumOfPointLight) && i < pointLightList.size() - 1) i++;
        PointLight *thisLight = pointLightList[i];
        Ray ray = thisLight->RandomPhoton();
        Color bounceIntensity = thisLight->GetPhotonIntensity();
        TraceCausticPhotonRay(ray, bounceIntensity, true);
        g_emitted++;
    }
    causticPhotonMap->ScalePhotonPowers(1.f / causticPhotonMap->NumPhotons());
    g_unbalanced.assign((unsigned char *)causticPhotonMap->GetPhotons(), (unsigned char *)causticPhotonMap->GetPhotons() + sizeof(cyPhotonMap::Photon) * maxPhotons);
    causticPhotonMap->PrepareForIrradianceEstimation();
    return true;
}
// BuildPhotonMap (Main.cpp:251-295) restated the same way for the GLOBAL photon map: the loop body, TracePhotonRay and
// RandomPhotonBounce are the reference's own (Main.cpp:296-317, MtlBlinn.cpp:140-202).
static bool BuildPhotonMapN(int maxPhotons)
{
    photonMap = new PhotonMap();
    photonMap->Resize(maxPhotons);
    memset((void *)&photonMap->photons[0], 0, sizeof(cyPhotonMap::Photon) * (maxPhotons + 1));
    std::vector<PointLight *> pointLightList;
    for (auto it = lights.begin(); it != lights.end(); ++it)
        if (PointLight *ptr = reinterpret_cast<PointLight *>(*it)) pointLightList.push_back(ptr);
    if (pointLightList.size() == 0) return false;
    sort(pointLightList.begin(), pointLightList.end(), ComparePointLight);
    float sumOfPointLight = 0;
    for (int i = 0; i < pointLightList.size(); ++i) sumOfPointLight += pointLightList[i]->GetIntensity() * pointLightList[i]->GetSize();
    g_emitted = 0;
    while (photonMap->NumPhotons() < maxPhotons) {
        float rnd = Rnd01();
        int i = 0;
        while (rnd > pointLightList[i]->GetProbability(sumOfPointLight) && i < pointLightList.size() - 1) i++;
        PointLight *thisLight = pointLightList[i];
        Ray ray = thisLight->RandomPhoton();
        Color bounceIntensity = thisLight->GetPhotonIntensity();
        TracePhotonRay(ray, bounceIntensity, true);
        g_emitted++;
    }
    photonMap->ScalePhotonPowers(1.f / photonMap->NumPhotons());
    g_unbalanced.assign((unsigned char *)photonMap->GetPhotons(), (unsigned char *)photonMap->GetPhotons() + sizeof(cyPhotonMap::Photon) * maxPhotons);
    photonMap->PrepareForIrradianceEstimation();
    return true;
}
#endif

static void usage()
{
    fprintf(stderr,
            "usage: ref_harness <scene.xml> <out_prefix> [--spp N] [--gi G] [--bounce B] [--seed S]\n"
            "                   [--region x0 y0 x1 y1] [--rays file] [--shadow file] cmd...\n"
            "  cmds: dump primary rays shadow render  (photon build only: photons gather; --photons N --gather file)\n");
    exit(2);
}

int main(int argc, char **argv)
{
    if (argc < 4) usage();
    const char *scene = argv[1];
    std::string prefix = argv[2];
    int spp = 1, gi = GIBounceCount, bounce = INTERNAL_REFLECTION_BOUNCE;
    uint32_t seed = 0;
    int rx0 = 0, ry0 = 0, rx1 = -1, ry1 = -1;
    std::string raysFile, shadowFile, gatherFile;
    int nPhotons = 10000;
    std::vector<std::string> cmds;
    for (int a = 3; a < argc; a++) {
        std::string s = argv[a];
        if (s == "--spp") spp = atoi(argv[++a]);
        else if (s == "--gi") gi = atoi(argv[++a]);
        else if (s == "--bounce") bounce = atoi(argv[++a]);
        else if (s == "--seed") seed = (uint32_t)strtoul(argv[++a], 0, 10);
        else if (s == "--region") { rx0 = atoi(argv[a + 1]); ry0 = atoi(argv[a + 2]); rx1 = atoi(argv[a + 3]); ry1 = atoi(argv[a + 4]); a += 4; }
        else if (s == "--rays") raysFile = argv[++a];
        else if (s == "--photons") nPhotons = atoi(argv[++a]);
        else if (s == "--gather") gatherFile = argv[++a];
        else if (s == "--shadow") shadowFile = argv[++a];
        else cmds.push_back(s);
    }
    omp_set_num_threads(1);
    const double t_start = omp_get_wtime();
    if (!LoadScene(scene)) return 1;
    if (std::find(cmds.begin(), cmds.end(), "beginrender") != cmds.end()) {
        // The reference's OWN BeginRender() (Main.cpp:178-242), nothing restated: camera frame, [-DUSE_PhotonMap: its own
        // BuildCausticPhotonMap(), 1 M photons, which writes Resource/causticPhotonMap.dat under the working directory],
        // CalculateLightsIntensity(), the pixel loop (one OpenMP thread: columns outside, rows inside) with PT_SampleCount = 32,
        // GI depth 3, gamma, Color24, SaveImages().  rand() is interposed by ONE stream that is never reset.
        g_key = bhrt_photon_key_sequential(seed);
        g_ctr = 0;
        BeginRender();
        const int W = camera.imgWidth, H = camera.imgHeight;
        std::vector<unsigned char> px((unsigned char *)renderImage.GetPixels(), (unsigned char *)renderImage.GetPixels() + (size_t)W * H * 3);
        WriteFile(prefix + ".begin_rgb8", px);
        std::vector<float> cam = {topLeft.x, topLeft.y, topLeft.z, dd_x.x, dd_x.y, dd_x.z, dd_y.x, dd_y.y, dd_y.z, allLightIntensity};
        WriteFile(prefix + ".begin_camera_f32", cam);
        std::vector<unsigned long long> meta = {(unsigned long long)g_ctr, (unsigned long long)W, (unsigned long long)H};
        WriteFile(prefix + ".begin_meta", meta);
        fprintf(stderr, "ref_harness: BeginRender() done (%llu rand() draws)\n", g_draws);
        return 0;
    }
    SetupCameraFrame();
    CalculateLightsIntensity(); // Main.cpp:116-123 (sorts `lights`, sums allLightIntensity)
    Flatten(&rootNode, -1, 1);
    const int W = camera.imgWidth, H = camera.imgHeight;
    if (rx1 < 0) { rx1 = W; ry1 = H; }
    const double t_loaded = omp_get_wtime();

    for (const std::string &cmd : cmds) {
        if (cmd == "dump") {
            DumpScene(prefix);
        } else if (cmd == "primary") {
            // rays through the reference's pixel "centre" (= corner, Main.cpp:145), no jitter
            std::vector<int> oi; std::vector<float> of;
            for (int j = 0; j < H; j++)
                for (int i = 0; i < W; i++) {
                    Vec3f pixelCenter = topLeft + (i + 1 / 2) * dd_x - (j + 1 / 2) * dd_y;
                    Ray ray = Ray(camera.pos, pixelCenter - camera.pos);
                    bool bHit = false;
                    HitInfo h = HitInfo();
                    recursive(&rootNode, ray, h, bHit, HIT_FRONT);
                    PushHit(oi, of, bHit, h);
                }
            WriteFile(prefix + ".primary_i32", oi);
            WriteFile(prefix + ".primary_f32", of);
        } else if (cmd == "aux") {
            // the images RenderImage holds beside the colour: zbuffer as the commented-out store of Main.cpp:231 would fill it
            // (HitInfo::z of the pixel's un-jittered ray), then the reference's own ComputeZBufferImage; plus the hit normal and
            // diffuse.Sample(uvw, duvw) = the optional DenoiseImage inputs (Main.cpp:70-71)
            renderImage.Init(W, H);
            std::vector<float> nrm((size_t)W * H * 3, 0.f), alb((size_t)W * H * 3, 0.f);
            for (int j = 0; j < H; j++)
                for (int i = 0; i < W; i++) {
                    Vec3f pixelCenter = topLeft + (i + 1 / 2) * dd_x - (j + 1 / 2) * dd_y;
                    Ray ray = Ray(camera.pos, pixelCenter - camera.pos);
                    bool bHit = false;
                    HitInfo h = HitInfo();
                    recursive(&rootNode, ray, h, bHit, HIT_FRONT);
                    const size_t pix = (size_t)j * W + i;
                    renderImage.GetZBuffer()[pix] = h.z;
                    if (bHit) {
                        nrm[pix * 3] = h.N.x; nrm[pix * 3 + 1] = h.N.y; nrm[pix * 3 + 2] = h.N.z;
                        const Material *m = h.node->GetMaterial();
                        Color kd = Color::Black();
                        if (const MtlBlinn *b = dynamic_cast<const MtlBlinn *>(m)) kd = b->diffuse.Sample(h.uvw, h.duvw);
                        else if (m) kd = Color::White(); // MultiMtl::Shade of an empty list, materials.h:71
                        alb[pix * 3] = kd.r; alb[pix * 3 + 1] = kd.g; alb[pix * 3 + 2] = kd.b;
                    }
                }
            std::vector<float> zb(renderImage.GetZBuffer(), renderImage.GetZBuffer() + (size_t)W * H);
            renderImage.ComputeZBufferImage();
            std::vector<unsigned char> zi(renderImage.GetZBufferImage(), renderImage.GetZBufferImage() + (size_t)W * H);
            WriteFile(prefix + ".aux_z_f32", zb);
            WriteFile(prefix + ".aux_zimg_u8", zi);
            WriteFile(prefix + ".aux_normal_f32", nrm);
            WriteFile(prefix + ".aux_albedo_f32", alb);
        } else if (cmd == "rays") {
            std::vector<float> in = ReadFile<float>(raysFile); // N x 7: o, d, hitSide
            std::vector<int> oi; std::vector<float> of;
            for (size_t r = 0; r + 6 < in.size(); r += 7) {
                Ray ray(Vec3f(in[r], in[r + 1], in[r + 2]), Vec3f(in[r + 3], in[r + 4], in[r + 5]));
                bool bHit = false;
                HitInfo h = HitInfo();
                recursive(&rootNode, ray, h, bHit, (int)in[r + 6]);
                PushHit(oi, of, bHit, h);
            }
            WriteFile(prefix + ".rays_i32", oi);
            WriteFile(prefix + ".rays_f32", of);
        } else if (cmd == "shadow") {
            std::vector<float> in = ReadFile<float>(shadowFile); // N x 7: o, d, t_max
            std::vector<float> out;
            struct Probe : public GenLight { // GenLight::Shadow is protected/static (lights.h:24)
                Color Illuminate(Vec3f const &, Vec3f const &) const { return Color(0, 0, 0); }
                Vec3f Direction(Vec3f const &) const { return Vec3f(0, 0, 0); }
                float GetIntensity() const { return 0; }
                static float S(Ray r, float t) { return Shadow(r, t); }
            };
            for (size_t r = 0; r + 6 < in.size(); r += 7) {
                Ray ray(Vec3f(in[r], in[r + 1], in[r + 2]), Vec3f(in[r + 3], in[r + 4], in[r + 5]));
                out.push_back(Probe::S(ray, in[r + 6]));
            }
            WriteFile(prefix + ".shadow_f32", out);
#ifdef USE_PhotonMap
        } else if (cmd == "photons") {
            // one sequential stream for the whole emission loop, like the reference's global rand()
            g_key = bhrt_sample_key(seed, 0xFFFFFFFFu, 0x50484F54u);
            g_ctr = 0;
            if (!BuildCausticPhotonMapN(nPhotons)) { fprintf(stderr, "no point light\n"); return 3; }
            std::vector<unsigned char> bal((unsigned char *)causticPhotonMap->GetPhotons(), (unsigned char *)causticPhotonMap->GetPhotons() + sizeof(cyPhotonMap::Photon) * nPhotons);
            WriteFile(prefix + ".photons_emitted", g_unbalanced);
            WriteFile(prefix + ".photons_balanced", bal);
            std::vector<unsigned long long> meta = {(unsigned long long)causticPhotonMap->NumPhotons(), g_emitted, (unsigned long long)g_ctr, (unsigned long long)(long long)causticPhotonMap->halfStoredPhotons};
            WriteFile(prefix + ".photons_meta", meta);
        } else if (cmd == "gphotons") { // the global photon map (BuildPhotonMap), same stream convention
            g_key = bhrt_sample_key(seed, 0xFFFFFFFFu, 0x50484F54u);
            g_ctr = 0;
            if (!BuildPhotonMapN(nPhotons)) { fprintf(stderr, "no point light\n"); return 3; }
            std::vector<unsigned char> bal((unsigned char *)photonMap->GetPhotons(), (unsigned char *)photonMap->GetPhotons() + sizeof(cyPhotonMap::Photon) * nPhotons);
            WriteFile(prefix + ".gphotons_emitted", g_unbalanced);
            WriteFile(prefix + ".gphotons_balanced", bal);
            std::vector<unsigned long long> meta = {(unsigned long long)photonMap->NumPhotons(), g_emitted, (unsigned long long)g_ctr, (unsigned long long)(long long)photonMap->halfStoredPhotons};
            WriteFile(prefix + ".gphotons_meta", meta);
        } else if (cmd == "gather") {
            std::vector<float> in = ReadFile<float>(gatherFile); // N x 6: p, normal
            std::vector<float> out;
            for (size_t r = 0; r + 5 < in.size(); r += 6) {
                Vec3f pos(in[r], in[r + 1], in[r + 2]), nrm(in[r + 3], in[r + 4], in[r + 5]), dir;
                Color irr;
                causticPhotonMap->EstimateIrradiance<1000>(irr, dir, 0.5, pos, &nrm); // as called at MtlBlinn.cpp:334
                out.push_back(irr.r); out.push_back(irr.g); out.push_back(irr.b);
                out.push_back(dir.x); out.push_back(dir.y); out.push_back(dir.z);
            }
            WriteFile(prefix + ".gather_f32", out);
#endif
        } else if (cmd == "render") {
            // per-sample body of PathTracing (Main.cpp:145-168) with rand() reset per (pixel, sample)
            const int rw = rx1 - rx0, rh = ry1 - ry0;
            std::vector<float> samples((size_t)rw * rh * spp * 3);
            std::vector<float> radiance((size_t)rw * rh * 3);
            std::vector<unsigned char> rgb((size_t)rw * rh * 3);
            std::vector<float> colorArray((size_t)rw * rh * 3); // Main.cpp:202,229: what DenoiseImage is given
            std::vector<unsigned> draws((size_t)rw * rh * spp);
            const float pixelLen = dd_x.Length();
            for (int j = ry0; j < ry1; j++)
                for (int i = rx0; i < rx1; i++) {
                    Vec3f pixelCenter = topLeft + (i + 1 / 2) * dd_x - (j + 1 / 2) * dd_y;
                    Color colorSum = Color::Black();
                    size_t pix = (size_t)(j - ry0) * rw + (i - rx0);
                    for (int s = 0; s < spp; s++) {
                        g_key = bhrt_sample_key(seed, (uint32_t)(j * W + i), (uint32_t)s);
                        g_ctr = 0;
                        Ray ray = Ray(camera.pos, RandomPositionInPixel(pixelCenter, pixelLen) - camera.pos);
                        bool bHit = false;
                        HitInfo h = HitInfo();
                        recursive(&rootNode, ray, h, bHit, HIT_FRONT);
                        Color c;
                        if (bHit) c = h.node->GetMaterial()->Shade(ray, h, lights, bounce, gi);
                        else {
                            Vec3f bguvw = Vec3f((float)i / camera.imgWidth, (float)j / camera.imgHeight, 0.0f);
                            c = background.Sample(bguvw);
                        }
                        colorSum += c;
                        float *o = &samples[(pix * spp + s) * 3];
                        o[0] = c.r; o[1] = c.g; o[2] = c.b;
                        draws[pix * spp + s] = g_ctr;
                    }
                    Color outColor = colorSum / spp; // Main.cpp:170 (int divisor -> float)
                    radiance[pix * 3 + 0] = outColor.r; radiance[pix * 3 + 1] = outColor.g; radiance[pix * 3 + 2] = outColor.b;
                    Color g = Color::Black(); // Main.cpp:220-226
                    const float inverseGama = 1 / 2.2f;
                    g.r = pow(outColor.r, inverseGama);
                    g.g = pow(outColor.g, inverseGama);
                    g.b = pow(outColor.b, inverseGama);
                    colorArray[pix * 3 + 0] = g.r; colorArray[pix * 3 + 1] = g.g; colorArray[pix * 3 + 2] = g.b;
                    Color24 q = Color24(g); // Main.cpp:230
                    rgb[pix * 3 + 0] = q.r; rgb[pix * 3 + 1] = q.g; rgb[pix * 3 + 2] = q.b;
                }
            WriteFile(prefix + ".samples_f32", samples);
            WriteFile(prefix + ".radiance_f32", radiance);
            WriteFile(prefix + ".rgb8", rgb);
            WriteFile(prefix + ".color_f32", colorArray);
            WriteFile(prefix + ".draws_u32", draws);
        } else {
            usage();
        }
    }
    // bench.py's cpu_baseline reads this line: scene load (XML, OBJ, BVH build) apart from the commands themselves
    fprintf(stderr, "ref_harness: load %.6f s, commands %.6f s\n", t_loaded - t_start, omp_get_wtime() - t_loaded);
    fprintf(stderr, "ref_harness: done (%llu rand() draws)\n", g_draws);
    return 0;
}
