// scene_host.cpp — host front-end of the render path: parses the reference's XML scene
// schema, loads OBJ meshes / textures, builds the cyBVH-compatible
// hierarchy and flattens everything into the blob the HIP kernels consume.
//
// Written from scratch; behaviour follows (all relative to /root/reference/BHRayTracer):
//   xmlload.cpp:65-132   LoadScene       :172-271 LoadNode      :275-303 LoadTransform
//   xmlload.cpp:307-390  LoadMaterial    :394-474 LoadLight     :478-521 ReadVector/Color/Float
//   xmlload.cpp:525-582  ReadTexture     Scenes/scene.h:229-232 Transformation::Scale/Rotate/Translate
//   Objects/TriObj/cyTriMesh.h:263-547 OBJ/MTL rules, :229-261 bbox + normals
//   DataStructure/cyBVH.h:122-142,242-328 BVH build    Main.cpp:116-123,179-192 lights sort, camera frame
// Compile with -ffp-contract=off: transforms, normals and BVH bounds must be bit-identical
// to the reference's (they decide hit indices).
#include "scene_host.h"

#include <ctype.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <algorithm>
#include <map>
#include <memory>

#include "bhrt.h"
#include "mini_xml.h"
#include "png_io.h"
#include "vecmath.h"

namespace bhrt {

// ------------------------------------------------------------------------------------------------
// Transformation (scene.h:208-246) on the host
// ------------------------------------------------------------------------------------------------
namespace {

struct Mat3 {
    float c[9];
};
Mat3 MatIdentity() { Mat3 m = {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; return m; }
// cyMatrix.h:671-681: per result column i: a = col0*r[i], b = col1*r[i+1], c = col2*r[i+2]; rd = a+b+c
Mat3 MatMul(const Mat3 &l, const Mat3 &r)
{
    Mat3 o;
    for (int i = 0; i < 9; i += 3)
        for (int j = 0; j < 3; j++) {
            float a = l.c[j] * r.c[i], b = l.c[3 + j] * r.c[i + 1], c = l.c[6 + j] * r.c[i + 2];
            o.c[i + j] = a + b + c;
        }
    return o;
}
// cyMatrix.h:799-819
Mat3 MatInverse(const Mat3 &m)
{
    const float *c = m.c;
    Mat3 inv;
    inv.c[0] = (c[4] * c[8] - c[5] * c[7]);
    inv.c[1] = (c[2] * c[7] - c[1] * c[8]);
    inv.c[2] = (c[1] * c[5] - c[2] * c[4]);
    inv.c[3] = (c[5] * c[6] - c[3] * c[8]);
    inv.c[4] = (c[0] * c[8] - c[2] * c[6]);
    inv.c[5] = (c[2] * c[3] - c[0] * c[5]);
    inv.c[6] = (c[3] * c[7] - c[4] * c[6]);
    inv.c[7] = (c[1] * c[6] - c[0] * c[7]);
    inv.c[8] = (c[0] * c[4] - c[1] * c[3]);
    float det = c[0] * inv.c[0] + c[1] * inv.c[3] + c[2] * inv.c[6];
    for (int i = 0; i < 9; i++) inv.c[i] /= det;
    return inv;
}

struct Xform {
    Mat3 tm = MatIdentity(), itm = MatIdentity();
    V3 pos = {0, 0, 0};
    void Transform(const Mat3 &m) // scene.h:232
    {
        tm = MatMul(m, tm);
        pos = mat_mul(m.c, pos);
        itm = MatInverse(tm);
    }
    void Scale(float sx, float sy, float sz) // scene.h:231
    {
        Mat3 m = {{0, 0, 0, 0, 0, 0, 0, 0, 0}};
        m.c[0] = sx; m.c[4] = sy; m.c[8] = sz;
        Transform(m);
    }
    void Rotate(V3 axis, float degrees) // scene.h:230 + cyMatrix.h:522-533
    {
        float angle = degrees * (float)M_PI / 180.0f;
        float sinAngle = sinf(angle), cosAngle = cosf(angle);
        float t = 1.0f - cosAngle;
        V3 a = t * axis;
        float txy = a.x * axis.y, txz = a.x * axis.z, tyz = a.y * axis.z;
        V3 s = sinAngle * axis;
        Mat3 m;
        m.c[0] = a.x * axis.x + cosAngle; m.c[1] = txy + s.z; m.c[2] = txz - s.y;
        m.c[3] = txy - s.z; m.c[4] = a.y * axis.y + cosAngle; m.c[5] = tyz + s.x;
        m.c[6] = txz + s.y; m.c[7] = tyz - s.x; m.c[8] = a.z * axis.z + cosAngle;
        Transform(m);
    }
    void Translate(V3 p) { pos = pos + p; } // scene.h:229
    void Store(bhrt_xform &o) const
    {
        memcpy(o.tm, tm.c, sizeof o.tm);
        memcpy(o.itm, itm.c, sizeof o.itm);
        o.pos[0] = pos.x; o.pos[1] = pos.y; o.pos[2] = pos.z;
    }
};

struct Col {
    float r, g, b;
};

bool NameIs(const XmlElement *e, const char *s) { return strcasecmp(e->name.c_str(), s) == 0; } // xmlload.cpp:34-38

void ReadFloat(const XmlElement *e, float &f, const char *name = "value") // xmlload.cpp:516-521
{
    double d = (double)f;
    e->QueryDouble(name, &d);
    f = (float)d;
}
void ReadVector(const XmlElement *e, V3 &v) // xmlload.cpp:478-493
{
    double x = (double)v.x, y = (double)v.y, z = (double)v.z;
    e->QueryDouble("x", &x); e->QueryDouble("y", &y); e->QueryDouble("z", &z);
    v.x = (float)x; v.y = (float)y; v.z = (float)z;
    float f = 1;
    ReadFloat(e, f);
    v = v * f;
}
void ReadColor(const XmlElement *e, Col &c) // xmlload.cpp:497-512
{
    double r = (double)c.r, g = (double)c.g, b = (double)c.b;
    e->QueryDouble("r", &r); e->QueryDouble("g", &g); e->QueryDouble("b", &b);
    c.r = (float)r; c.g = (float)g; c.b = (float)b;
    float f = 1;
    ReadFloat(e, f);
    c.r *= f; c.g *= f; c.b *= f;
}
void LoadTransform(Xform &t, const XmlElement *e) // xmlload.cpp:275-303
{
    for (auto &ch : e->children) {
        const XmlElement *c = ch.get();
        if (NameIs(c, "scale")) {
            V3 s = {1, 1, 1};
            ReadVector(c, s);
            t.Scale(s.x, s.y, s.z);
        } else if (NameIs(c, "rotate")) {
            V3 s = {0, 0, 0};
            ReadVector(c, s);
            s = normalized(s);
            float a = 0; // the reference leaves `a` uninitialised when angle= is absent
            ReadFloat(c, a, "angle");
            t.Rotate(s, a);
        } else if (NameIs(c, "translate")) {
            V3 p = {0, 0, 0};
            ReadVector(c, p);
            t.Translate(p);
        }
    }
}

std::string NormalisePath(const char *p) // Windows separators in 9 of 19 shipped scenes
{
    std::string s = p ? p : "";
    for (auto &ch : s) if (ch == '\\') ch = '/';
    return s;
}

struct TexData {
    int type = BHRT_TEX_FILE;
    int w = 0, h = 0;
    Col c1 = {0, 0, 0}, c2 = {1, 1, 1};
    std::vector<uint8_t> rgb;
};
struct TexMapData {
    Xform xf;
    int texture = -1;
};
struct TexColorData {
    Col color = {0, 0, 0};
    int map = -1;
};
struct MaterialData {
    std::string name;
    int kind = BHRT_MTL_BLINN;
    TexColorData diffuse, specular, refraction;
    float glossiness = 20.0f;
    Col absorption = {0, 0, 0};
    float ior = 1;
    float refraction_glossiness = 0;
    MaterialData() // materials.h:23-25
    {
        diffuse.color = {0.5f, 0.5f, 0.5f};
        specular.color = {0.7f, 0.7f, 0.7f};
        refraction.color = {0, 0, 0};
    }
};
struct LightData {
    int type;
    Col intensity = {0, 0, 0};
    V3 vec = {0, 0, 0};
    float size = 0;
    float Gray() const { return (intensity.r + intensity.g + intensity.b) / 3.0f; } // cyColor.h Gray()
};
struct NodeData {
    Xform xf;
    int parent = -1, depth = 1, obj_type = BHRT_OBJ_NONE, mesh = -1, material = -1;
    int subtree_end = 0;
    std::string material_name;
    bool has_material_name = false;
};

struct Loader {
    int bvh_device = -1;
    std::string fatal;
    std::string xml_dir;
    FlatScene *out;
    std::vector<NodeData> nodes;
    std::vector<std::pair<int, std::string>> node_mtl_list; // xmlload.cpp:55-61
    std::vector<MaterialData> materials;
    std::vector<LightData> lights;
    std::vector<std::unique_ptr<HostMesh>> meshes;
    std::map<std::string, int> mesh_by_name; // objList.Find (scene.h:187)
    std::vector<TexData> textures;
    std::map<std::string, int> tex_by_name; // textureList.Find
    std::vector<TexMapData> texmaps;
    TexColorData background, environment;

    void Warn(const std::string &w) { out->warnings.push_back(w); }

    // The reference opens asset paths relative to the process cwd.  Same here; additionally the
    // scene file's own directory and $BHRT_ASSET_ROOT are tried so scenes are relocatable.
    std::string Resolve(const std::string &name)
    {
        std::string n = NormalisePath(name.c_str());
        std::vector<std::string> cands = {n};
        if (!xml_dir.empty()) cands.push_back(xml_dir + "/" + n);
        if (const char *root = getenv("BHRT_ASSET_ROOT")) cands.push_back(std::string(root) + "/" + n);
        for (auto &c : cands) {
            FILE *fp = fopen(c.c_str(), "rb");
            if (fp) { fclose(fp); return c; }
        }
        return n;
    }

    int FindMaterial(const std::string &name) // scene.h:305: first match
    {
        for (size_t i = 0; i < materials.size(); i++)
            if (materials[i].name == name) return (int)i;
        return -1;
    }

    int ReadTextureFile(const char *texName) // xmlload.cpp:562-582 + Texture.cpp:58-93
    {
        auto it = tex_by_name.find(texName);
        if (it != tex_by_name.end()) return it->second;
        std::string path = Resolve(texName);
        TexData t;
        t.type = BHRT_TEX_FILE;
        bool ok = false;
        size_t len = path.size();
        if (len >= 3) {
            char ext[4] = {(char)tolower(path[len - 3]), (char)tolower(path[len - 2]), (char)tolower(path[len - 1]), 0};
            if (!strcmp(ext, "png")) ok = LoadPngRgb(path.c_str(), t.rgb, t.w, t.h);
            else if (!strcmp(ext, "ppm")) ok = LoadPpm(path.c_str(), t);
        }
        if (!ok) {
            Warn(std::string("texture: error loading file \"") + texName + "\"");
            return -1; // not cached: the reference retries (and fails) every time
        }
        textures.push_back(std::move(t));
        tex_by_name[texName] = (int)textures.size() - 1;
        return (int)textures.size() - 1;
    }
    static bool LoadPpm(const char *path, TexData &t) // Texture.cpp:32-54 (binary P6, maxval line skipped)
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) return false;
        auto readLine = [&](char *buf, int size) {
            int i;
            for (i = 0; i < size; i++) {
                int c = fgetc(fp);
                buf[i] = (char)c;
                if (feof(fp) || buf[i] == '\n' || buf[i] == '\r') { buf[i] = 0; return; }
            }
            buf[size - 1] = 0;
        };
        char buf[1024];
        readLine(buf, 1024);
        if (buf[0] != 'P' && buf[1] != '6') { fclose(fp); return false; }
        readLine(buf, 1024);
        while (buf[0] == '#') readLine(buf, 1024);
        if (sscanf(buf, "%d %d", &t.w, &t.h) != 2 || t.w <= 0 || t.h <= 0 || t.w > 32768 || t.h > 32768 || (size_t)t.w * t.h > ((size_t)1 << 28)) { fclose(fp); return false; }
        readLine(buf, 1024);
        while (buf[0] == '#') readLine(buf, 1024);
        t.rgb.resize((size_t)t.w * t.h * 3);
        size_t got = fread(t.rgb.data(), 3, (size_t)t.w * t.h, fp);
        (void)got;
        fclose(fp);
        return true;
    }

    int ReadTexture(const XmlElement *e) // xmlload.cpp:525-558; returns texmap index or -1
    {
        const char *texName = e->Attribute("texture");
        if (!texName) return -1;
        int tex = -1;
        if (strcasecmp(texName, "checkerboard") == 0) {
            TexData t;
            t.type = BHRT_TEX_CHECKER;
            for (auto &ch : e->children) {
                if (NameIs(ch.get(), "color1")) { Col c = {0, 0, 0}; ReadColor(ch.get(), c); t.c1 = c; }
                else if (NameIs(ch.get(), "color2")) { Col c = {0, 0, 0}; ReadColor(ch.get(), c); t.c2 = c; }
            }
            textures.push_back(std::move(t));
            tex = (int)textures.size() - 1;
        } else {
            tex = ReadTextureFile(texName);
        }
        TexMapData m;
        m.texture = tex;
        LoadTransform(m.xf, e);
        texmaps.push_back(m);
        return (int)texmaps.size() - 1;
    }

    void LoadMaterial(const XmlElement *e) // xmlload.cpp:307-390
    {
        const char *name = e->Attribute("name");
        const char *type = e->Attribute("type");
        if (!type || strcasecmp(type, "blinn") != 0) return; // unknown type: not appended
        MaterialData m;
        m.name = name ? name : "";
        for (auto &chp : e->children) {
            const XmlElement *c = chp.get();
            Col col = {1, 1, 1};
            float f = 1;
            if (NameIs(c, "diffuse")) { ReadColor(c, col); m.diffuse.color = col; m.diffuse.map = ReadTexture(c); }
            else if (NameIs(c, "specular")) { ReadColor(c, col); m.specular.color = col; m.specular.map = ReadTexture(c); }
            else if (NameIs(c, "glossiness")) { ReadFloat(c, f); m.glossiness = f; }
            else if (NameIs(c, "emission")) { ReadTexture(c); /* parsed, never shaded */ }
            else if (NameIs(c, "reflection")) { ReadTexture(c); /* parsed, never shaded (Q8) */ }
            else if (NameIs(c, "refraction")) {
                ReadColor(c, col);
                m.refraction.color = col;
                ReadFloat(c, f, "index");
                m.ior = f;
                m.refraction.map = ReadTexture(c);
                f = 0;
                ReadFloat(c, f, "glossiness");
                m.refraction_glossiness = f;
            } else if (NameIs(c, "absorption")) { ReadColor(c, col); m.absorption = col; }
        }
        materials.push_back(m);
    }

    void LoadLight(const XmlElement *e) // xmlload.cpp:394-474
    {
        const char *type = e->Attribute("type");
        if (!type) return;
        LightData l;
        if (strcasecmp(type, "ambient") == 0) {
            l.type = BHRT_LIGHT_AMBIENT;
            for (auto &c : e->children)
                if (NameIs(c.get(), "intensity")) { Col col = {1, 1, 1}; ReadColor(c.get(), col); l.intensity = col; }
        } else if (strcasecmp(type, "direct") == 0) {
            l.type = BHRT_LIGHT_DIRECT;
            l.vec = {0, 0, 1}; // lights.h:48
            for (auto &c : e->children) {
                if (NameIs(c.get(), "intensity")) { Col col = {1, 1, 1}; ReadColor(c.get(), col); l.intensity = col; }
                else if (NameIs(c.get(), "direction")) { V3 v = {1, 1, 1}; ReadVector(c.get(), v); l.vec = normalized(v); }
            }
        } else if (strcasecmp(type, "point") == 0) {
            l.type = BHRT_LIGHT_POINT;
            for (auto &c : e->children) {
                if (NameIs(c.get(), "intensity")) { Col col = {1, 1, 1}; ReadColor(c.get(), col); l.intensity = col; }
                else if (NameIs(c.get(), "position")) { V3 v = {0, 0, 0}; ReadVector(c.get(), v); l.vec = v; }
                else if (NameIs(c.get(), "size")) { float f = 0; ReadFloat(c.get(), f); l.size = f; }
            }
        } else
            return;
        lights.push_back(l);
    }

    // MultiMtl built from a mesh's .mtl (xmlload.cpp:219-250).  Shade() dispatches on hInfo.mtlID,
    // which triangle hits never set -> sub-material 0 is the only one ever shaded.
    void AppendMultiMtl(const std::string &name, const HostMesh &mesh)
    {
        MaterialData m;
        m.name = name;
        if (mesh.mtls.empty()) { m.kind = BHRT_MTL_WHITE; materials.push_back(m); return; }
        const HostMesh::Mtl &s = mesh.mtls[0];
        m.diffuse.color = {s.Kd[0], s.Kd[1], s.Kd[2]};
        m.specular.color = {s.Ks[0], s.Ks[1], s.Ks[2]};
        m.glossiness = s.Ns;
        m.ior = s.Ni;
        // every mesh material's textures are loaded by the reference (side effect: texture list); sub-material 0's are used
        for (size_t i = 0; i < mesh.mtls.size(); i++) {
            const HostMesh::Mtl &mi = mesh.mtls[i];
            int mapKd = -1, mapKs = -1;
            if (!mi.map_Kd.empty()) { TexMapData t; t.texture = ReadTextureFile(mi.map_Kd.c_str()); texmaps.push_back(t); mapKd = (int)texmaps.size() - 1; }
            if (!mi.map_Ks.empty()) { TexMapData t; t.texture = ReadTextureFile(mi.map_Ks.c_str()); texmaps.push_back(t); mapKs = (int)texmaps.size() - 1; }
            if (i == 0) {
                if (mapKd >= 0) m.diffuse.map = mapKd;
                if (mapKs >= 0) m.diffuse.map = mapKs; // xmlload.cpp:230 sets the DIFFUSE texture from map_Ks (sic)
            }
        }
        if (s.illum > 2 && s.illum <= 7) {
            float gloss = acosf(powf(2, 1 / s.Ns));
            if (s.illum >= 6) {
                m.refraction.color = {1 - s.Tf[0], 1 - s.Tf[1], 1 - s.Tf[2]};
                m.refraction_glossiness = gloss;
            }
        }
        materials.push_back(m);
    }

    void LoadNode(int parent, int depth, const XmlElement *e) // xmlload.cpp:172-271
    {
        int idx = (int)nodes.size();
        nodes.emplace_back();
        nodes[idx].parent = parent;
        nodes[idx].depth = depth;
        const char *name = e->Attribute("name");
        const char *mtlName = e->Attribute("material");
        if (mtlName) node_mtl_list.emplace_back(idx, mtlName);
        const char *type = e->Attribute("type");
        if (type) {
            if (strcasecmp(type, "sphere") == 0) nodes[idx].obj_type = BHRT_OBJ_SPHERE;
            else if (strcasecmp(type, "plane") == 0) nodes[idx].obj_type = BHRT_OBJ_PLANE;
            else if (strcasecmp(type, "obj") == 0) {
                std::string key = name ? name : "";
                auto it = mesh_by_name.find(key);
                int mi = -1;
                if (it != mesh_by_name.end()) mi = it->second;
                else {
                    std::unique_ptr<HostMesh> hm(new HostMesh);
                    std::string err;
                    std::string path = Resolve(key);
                    if (!name || !LoadObj(path.c_str(), mtlName == nullptr, *hm, err)) {
                        Warn("ERROR: Cannot load file \"" + key + "\" (" + err + "); node keeps a null object");
                    } else {
                        if (hm->vn.empty()) ComputeNormals(*hm);
                        ComputeBoundingBox(*hm);
                        if (!hm->had_vt) Warn("mesh \"" + key + "\" has no vt lines: the reference dereferences null here (SURVEY.md Q12); uvw = 0 is used");
                        if (bvh_device >= 0) {
                            if (BuildBvhDevice(*hm, 4, bvh_device) != 0 && fatal.empty()) fatal = "device BVH build of \"" + key + "\" failed: " + bhrt_last_error();
                        } else BuildBvh(*hm, 4);
                        meshes.push_back(std::move(hm));
                        mi = (int)meshes.size() - 1;
                        mesh_by_name[key] = mi;
                        if (!meshes[mi]->mtls.empty() && FindMaterial(key) < 0) { // xmlload.cpp:219-250
                            AppendMultiMtl(key, *meshes[mi]);
                            node_mtl_list.emplace_back(idx, key);
                        }
                    }
                }
                if (mi >= 0) { nodes[idx].obj_type = BHRT_OBJ_MESH; nodes[idx].mesh = mi; }
            } else
                Warn(std::string("object: unknown type \"") + type + "\"");
        }
        for (auto &c : e->children)
            if (NameIs(c.get(), "object")) LoadNode(idx, depth + 1, c.get());
        nodes[idx].subtree_end = (int)nodes.size();
        LoadTransform(nodes[idx].xf, e);
    }
};

// ------------------------------------------------------------------------------------------------
// blob writer
// ------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------------------
// Leaf skip (DESIGN.md 4 "leaf skip"; the device side is device_trace.h::leaf_skip).  TraceBVHNode visits the sibling of a child that
// returned nothing even when the sibling's box was missed (TriObj.cpp:245-248,263-266), and TraceBVHShadow visits both children once
// either box is hit (TriObj.cpp:286-300).  When that box-missed node is a LEAF the visit runs IntersectTriangle (TriObj.cpp:68-189) on its
// triangles with no box test — and nothing ties that function's arithmetic to the slab test's: from far away its three signed areas are
// rounding noise and a triangle can be "hit" by a ray that passes it at a distance.  So a leaf may only be left out when a proof says
// that none of its triangles can be accepted for THIS ray.  What is computed here, per leaf, in double precision with every bound doubled:
//
// Notation: u = 2^-24; T a triangle of the leaf with the float normal n = vN, c = vN.v0 (float), projected on the plane that drops the axis
// of the largest |n_i| (slope = (|n_a| + |n_b|) / |n_axis| <= 2); Q = the 2-D point the test projects its hit point X = fl(o + fl(t d)) to.
//  (1) 2-D sign test (tri_areas).  Every computed area is the exact cross product of the two vertex differences as seen from Q, up to
//      |err| <= 4.1 u |p_i - Q| |p_j - Q| (two rounded differences, two products, one sum; halving is exact).  For Q outside T at distance y
//      (nearest point on an edge, or a vertex with interior angle alpha) one of the exact areas is negative with magnitude >= e s y
//      (e = shortest edge, s = sin(alpha_min / 2)) and, the three summing to the triangle's own |c_T|, one is >= |c_T| / 3; both are computed
//      with their true signs — mixed signs, i.e. REJECT — while  e s y > 4.1 u (y + E)^2  and  |c_T| / 3 > 4.1 u (y + E)^2  (E = longest
//      edge): for  delta_T < y < L_T  with delta_T, L_T the roots / bounds of those two inequalities.
//  (2) How far X can be.  X lies within rho of the ray's line and within a rounding error of T's plane; if the line passes through the leaf
//      box inflated by `big` (point P) and |cos(n, d)| >= kappa0 = 1/16 for every triangle of the leaf, then |X - v0| <= (1 + 1.001 / kappa0)
//      diag(box + 2 big) + 64 u (1024 + B + 1) / kappa0 + 1e-3 (for |o|, |box| <= 1024: the second term is what the roundings of t and X amount to
//      along the ray after the division by |cos|).  The leaf qualifies when that is below every L_T: the "far" branch of (1) cannot occur.
//  (3) Hence an accepted X has Q within delta_T of T's projection, its third coordinate within slope delta_T + G_T + 64 u (omax + B + 1) of
//      T's extent (G_T = how far T's own vertices are off the float plane), so X lies in T's box inflated by that, and the ray's line
//      passes through the LEAF box inflated by m = k0 + k1 omax  (k1 = 128 u;  k0 >= 2 max_T max(delta_T, slope delta_T + G_T) + 128 u (B + 1)).
//      A ray whose line MISSES the leaf box inflated by m cannot be accepted by any triangle of the leaf.
//  The angle condition of (2) is a cone: all unit normals of the leaf within omega of an axis a; a ray with |d.a| >= mu |d|,
//  mu = cos(acos(kappa0) - omega - margin), keeps |cos(n, d)| >= kappa0.  The cone is stored in the leaf's `parent` word of the breadth-first
//  copy as three 10-bit components of A = a / mu and a 2-bit exponent, rounded towards zero and VERIFIED after dequantising.
static uint32_t PackCone(const double A[3], double Aq[3])
{
    double mx = std::max(fabs(A[0]), std::max(fabs(A[1]), fabs(A[2])));
    if (!(mx > 0) || !(mx < 16.0)) return 0;
    int e = 0;
    while (e < 3 && mx >= ldexp(511.0, e - 8)) e++;
    if (mx >= ldexp(511.0, e - 8)) return 0;
    uint32_t w = (uint32_t)e << 30;
    for (int k = 0; k < 3; k++) {
        const int q = (int)trunc(ldexp(A[k], 8 - e)); // towards zero: |Aq| <= |A|
        Aq[k] = ldexp((double)q, e - 8);
        w |= ((uint32_t)q & 0x3ffu) << (10 * k);
    }
    return w;
}
static void ComputeLeafSkip(const HostMesh &m, const std::vector<bhrt_tri> &lt, std::vector<bhrt_bvh_node> &dn, std::vector<uint32_t> &dparent, bhrt_mesh &o)
{
    const double u = ldexp(1.0, -24), kappa0 = 1.0 / 16.0, c4 = 4.1 * u, tiny = ldexp(1.0, -100);
    dparent.assign(dn.size(), 0);
    for (size_t i = 0; i < dn.size(); i++) dparent[i] = dn[i].parent;
    o.skip_k0 = o.skip_k1 = o.skip_big = o.skip_omax = 0;
    for (size_t i = 1; i < dn.size(); i++)
        if (dn[i].data & 0x80000000u) dn[i].parent = 0; // a leaf's word holds its cone from here on: 0 = never skipped
    double B = 0;
    for (int k = 0; k < 3; k++) B = std::max(B, std::max(fabs((double)m.bound_min[k]), fabs((double)m.bound_max[k])));
    if (!(B <= 1024.0) || dn.size() < 2) return;
    double diag_sum = 0;
    size_t n_leaves = 0;
    for (size_t i = 1; i < dn.size(); i++)
        if (dn[i].data & 0x80000000u) {
            double s2 = 0;
            for (int k = 0; k < 3; k++) { const double ex = (double)dn[i].b[k + 3] - (double)dn[i].b[k]; s2 += ex * ex; }
            diag_sum += sqrt(s2);
            n_leaves++;
        }
    if (!n_leaves || !(diag_sum > 0)) return;
    // the near-miss radius: 2 mean leaf diagonals (of the 100 k-triangle mesh's 33 295 leaves 30 082 then pass the "far" condition below; with 4: 5 200, with 1: 32 314)
    const double big = 2.0 * diag_sum / (double)n_leaves;
    const double k1 = 128.0 * u, k0 = ldexp(B + 1.0, -13);
    size_t n_ok = 0;
    for (size_t i = 1; i < dn.size(); i++) {
        if (!(dn[i].data & 0x80000000u)) continue;
        const uint32_t cnt = ((dn[i].data >> 28) & 7u) + 1, off = dn[i].data & 0x0fffffffu;
        bool ok = true;
        double m0 = 0, Lmin = 1e300, axis_sum[3] = {0, 0, 0};
        double nrm[8][3];
        for (uint32_t t = 0; t < cnt && ok; t++) {
            const bhrt_tri &tr = lt[off + t];
            const uint32_t axis = tr.face_axis >> 30, face = tr.face_axis & 0x3fffffffu;
            const double n[3] = {tr.vN[0], tr.vN[1], tr.vN[2]}, c = tr.vN_dot_v0;
            const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (axis > 2 || !(nl > 0) || !std::isfinite(nl)) { ok = false; break; }
            for (int k = 0; k < 3; k++) { nrm[t][k] = n[k] / nl; axis_sum[k] += nrm[t][k]; }
            const double p[3][2] = {{tr.p0[0], tr.p0[1]}, {tr.p1[0], tr.p1[1]}, {tr.p2[0], tr.p2[1]}};
            double len[3], E = 0, e = 1e300;
            for (int k = 0; k < 3; k++) { // edge k = (p_{k+1}, p_{k+2})
                const double dx = p[(k + 2) % 3][0] - p[(k + 1) % 3][0], dy = p[(k + 2) % 3][1] - p[(k + 1) % 3][1];
                len[k] = sqrt(dx * dx + dy * dy);
                E = std::max(E, len[k]); e = std::min(e, len[k]);
            }
            const double cT = fabs((p[1][0] - p[0][0]) * (p[2][1] - p[0][1]) - (p[1][1] - p[0][1]) * (p[2][0] - p[0][0])) * (1.0 - 1e-9);
            if (!(cT > 0) || !(e > 0)) { ok = false; break; }
            double s = 1.0; // sin(alpha_min / 2); the angle at vertex k lies between edges k+1 and k+2 and has sin(alpha_k) = cT / (their lengths)
            for (int k = 0; k < 3; k++) {
                const double la = len[(k + 1) % 3], lb = len[(k + 2) % 3];
                const double ax = p[(k + 1) % 3][0] - p[k][0], ay = p[(k + 1) % 3][1] - p[k][1], bx = p[(k + 2) % 3][0] - p[k][0], by = p[(k + 2) % 3][1] - p[k][1];
                const double cosa = std::max(-1.0, std::min(1.0, (ax * bx + ay * by) / (lb * la) * (1.0 + 1e-9) + 1e-12)); // edges at vertex k have lengths len[k+2] and len[k+1]
                s = std::min(s, sqrt(std::max(0.0, (1.0 - cosa) / 2.0)));
            }
            s *= (1.0 - 1e-6);
            // e s y > c4 (y + E)^2 + tiny  <=>  c4 y^2 - (e s - 2 c4 E) y + c4 E^2 + tiny < 0
            const double bq = e * s - 2.0 * c4 * E, disc = bq * bq - 4.0 * c4 * (c4 * E * E + tiny);
            if (!(bq > 0) || !(disc > 0)) { ok = false; break; }
            const double y_lo = (bq - sqrt(disc)) / (2.0 * c4), y_hi = (bq + sqrt(disc)) / (2.0 * c4);
            const double L1 = sqrt(cT / (3.0 * c4)) - E;
            const double delta = 2.0 * y_lo + tiny, L = 0.5 * std::min(L1, y_hi);
            if (!(delta < L)) { ok = false; break; }
            Lmin = std::min(Lmin, L);
            const int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
            const double slope = (fabs(n[ia]) + fabs(n[ib])) / fabs(n[axis]);
            double G = 0;
            for (int k = 0; k < 3; k++) {
                const float *pv = &m.v[(size_t)m.f[(size_t)face * 3 + k] * 3];
                G = std::max(G, fabs(n[0] * pv[0] + n[1] * pv[1] + n[2] * pv[2] - c) / fabs(n[axis]));
            }
            m0 = std::max(m0, std::max(delta, slope * delta + G));
        }
        if (!ok) continue;
        if (!(2.0 * m0 + 128.0 * u * (B + 1.0) <= k0)) continue;
        double d2 = 0;
        for (int k = 0; k < 3; k++) { const double ex = (double)dn[i].b[k + 3] - (double)dn[i].b[k] + 2.0 * big; d2 += ex * ex; }
        // + the roundings of t and X seen along the ray: the point o + t d is off the plane by <= 11 u |n|_1 (omax + B + |t d|), which the division by
        // |cos| >= kappa0 turns into <= 18 u (omax + B + ...) / kappa0 along the line — 64 u (1024 + B + 1) / kappa0 covers it for every tame ray
        const double far_slack = 64.0 * u * (1024.0 + B + 1.0) / kappa0 + 1e-3;
        if (!((1.0 + 1.001 / kappa0) * sqrt(d2) + far_slack <= Lmin)) continue;
        const double al = sqrt(axis_sum[0] * axis_sum[0] + axis_sum[1] * axis_sum[1] + axis_sum[2] * axis_sum[2]);
        if (!(al > 0)) continue;
        double a[3] = {axis_sum[0] / al, axis_sum[1] / al, axis_sum[2] / al}, omega = 0;
        for (uint32_t t = 0; t < cnt; t++) omega = std::max(omega, acos(std::max(-1.0, std::min(1.0, nrm[t][0] * a[0] + nrm[t][1] * a[1] + nrm[t][2] * a[2]))));
        const double psi = acos(kappa0) - omega - 0.02;
        if (!(psi > 0.1)) continue;
        const double mu = cos(psi);
        double A[3] = {a[0] / mu, a[1] / mu, a[2] / mu}, Aq[3];
        const uint32_t w = PackCone(A, Aq);
        if (!w) continue;
        const double aql = sqrt(Aq[0] * Aq[0] + Aq[1] * Aq[1] + Aq[2] * Aq[2]);
        if (!(aql > 0)) continue;
        bool cone_ok = true; // the guarantee of the STORED cone: |d.Aq| >= |d| keeps every |cos(n_T, d)| >= kappa0
        for (uint32_t t = 0; t < cnt; t++) {
            const double om = acos(std::max(-1.0, std::min(1.0, (nrm[t][0] * Aq[0] + nrm[t][1] * Aq[1] + nrm[t][2] * Aq[2]) / aql)));
            if (!(acos(std::min(1.0, 1.0 / aql)) + om <= acos(kappa0) - 1e-3)) cone_ok = false;
        }
        if (!cone_ok) continue;
        dn[i].parent = w;
        n_ok++;
    }
    if (!n_ok) return;
    o.skip_k0 = (float)(k0 * (1.0 + 1e-6)); o.skip_k1 = (float)(k1 * (1.0 + 1e-6)); o.skip_big = (float)(big * (1.0 - 1e-6)); o.skip_omax = 1024.f;
}

struct BlobWriter {
    std::vector<uint8_t> &b;
    explicit BlobWriter(std::vector<uint8_t> &v) : b(v) {}
    uint64_t Append(const void *p, size_t n)
    {
        // every array starts a 128-byte line: a BVH node pair (64 B) is then one half line and never straddles two (16-byte starts
        // put the pairs of the mesh scenes at 112 mod 128: two sectors per pair, two lines for every other pair)
        b.resize((b.size() + 127) / 128 * 128, 0);
        uint64_t off = b.size();
        if (n) b.insert(b.end(), (const uint8_t *)p, (const uint8_t *)p + n);
        return off;
    }
};

void StoreTexColor(bhrt_texcolor &o, const TexColorData &t)
{
    o.color[0] = t.color.r; o.color[1] = t.color.g; o.color[2] = t.color.b;
    o.map = t.map;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// OBJ / MTL  (cyTriMesh.h:263-547)
// ------------------------------------------------------------------------------------------------
namespace {
// One logical line: leading blanks and '#' comment lines skipped, inner runs of blanks collapsed to a
// single space, cut at 1023 characters (Buffer::ReadLine, cyTriMesh.h:278-305).
struct LineReader {
    const std::string &s;
    size_t p = 0;
    bool eof = false;
    explicit LineReader(const std::string &text) : s(text) {}
    int Get()
    {
        if (p < s.size()) return (unsigned char)s[p++];
        eof = true;
        return -1;
    }
    int Read(std::string &line)
    {
        line.clear();
        int c = Get();
        while (!eof) {
            while (c >= 0 && isspace(c)) c = Get();
            if (c == '#') { while (!eof && c != '\n' && c != '\r' && c != 0) c = Get(); }
            else break;
        }
        bool inspace = false;
        while (line.size() < 1023) {
            if (eof || c == '\n' || c == '\r' || c == 0) break;
            if (isspace(c)) inspace = true;
            else {
                if (inspace) line.push_back(' ');
                inspace = false;
                line.push_back((char)c);
            }
            c = Get();
        }
        return (int)line.size();
    }
};
bool IsCommand(const std::string &l, const char *cmd)
{
    size_t n = strlen(cmd);
    if (l.compare(0, n, cmd) != 0) return false;
    return l.size() == n || l[n] == ' ';
}
void ReadVertex(const std::string &l, float *v)
{
    v[0] = v[1] = v[2] = 0;
    if (l.size() > 2) sscanf(l.c_str() + 2, "%f %f %f", &v[0], &v[1], &v[2]);
}
void ReadFloat3(const std::string &l, float *f)
{
    f[0] = f[1] = f[2] = 0;
    int n = l.size() > 2 ? sscanf(l.c_str() + 2, "%f %f %f", &f[0], &f[1], &f[2]) : 0;
    if (n == 1) f[2] = f[1] = f[0];
}
std::string CopyFrom(const std::string &l, size_t start)
{
    while (start < l.size() && l[start] <= ' ') start++;
    return start < l.size() ? l.substr(start) : std::string();
}
bool ReadWholeFile(const char *path, std::string &text)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
    fclose(fp);
    return true;
}
} // namespace

bool LoadObj(const char *path, bool load_mtl, HostMesh &m, std::string &err)
{
    std::string text;
    if (!ReadWholeFile(path, text)) { err = "cannot open file"; return false; }
    struct MtlData { std::string name; unsigned firstFace = 0, faceCount = 0; };
    std::vector<MtlData> mtlData;
    std::vector<std::string> mtlFiles;
    std::vector<int> faceMtlIndex;
    std::vector<float> _v, _vn, _vt;
    std::vector<uint32_t> _f, _fn, _ft;
    int currentMtl = -1;
    bool hasTextures = false, hasNormals = false;

    LineReader rd(text);
    std::string line;
    while (int rb = rd.Read(line)) {
        if (IsCommand(line, "v")) { float v[3]; ReadVertex(line, v); _v.insert(_v.end(), v, v + 3); }
        else if (IsCommand(line, "vt")) { float v[3]; ReadVertex(line, v); _vt.insert(_vt.end(), v, v + 3); hasTextures = true; }
        else if (IsCommand(line, "vn")) { float v[3]; ReadVertex(line, v); _vn.insert(_vn.end(), v, v + 3); hasNormals = true; }
        else if (IsCommand(line, "f")) {
            // fan triangulation + 1-based / negative indices (cyTriMesh.h:379-438)
            int facevert = -1;
            bool inspace = true, negative = false;
            int type = 0;
            unsigned index = 0;
            uint32_t face[3] = {0, 0, 0}, tface[3] = {0, 0, 0}, nface[3] = {0, 0, 0};
            size_t nFacesBefore = _f.size() / 3;
            auto push = [&]() {
                _f.insert(_f.end(), face, face + 3);
                if (hasTextures) _ft.insert(_ft.end(), tface, tface + 3);
                if (hasNormals) _fn.insert(_fn.end(), nface, nface + 3);
                faceMtlIndex.push_back(currentMtl);
            };
            for (int i = 2; i < rb; i++) {
                char ch = line[i];
                if (ch == ' ') { inspace = true; continue; }
                if (inspace) {
                    inspace = false; negative = false; type = 0; index = 0;
                    if (facevert < 2) {
                        if (facevert == -1) { face[0] = face[1] = face[2] = 0; tface[0] = tface[1] = tface[2] = 0; nface[0] = nface[1] = nface[2] = 0; }
                        facevert++;
                    } else {
                        push();
                        face[1] = face[2];
                        if (hasTextures) tface[1] = tface[2];
                        if (hasNormals) nface[1] = nface[2];
                    }
                }
                if (ch == '/') { type++; index = 0; }
                if (ch == '-') negative = true;
                if (ch >= '0' && ch <= '9') {
                    index = index * 10 + (unsigned)(ch - '0');
                    switch (type) {
                    case 0: face[facevert] = negative ? (unsigned)(_v.size() / 3) - index : index - 1; break;
                    case 1: tface[facevert] = negative ? (unsigned)(_vt.size() / 3) - index : index - 1; hasTextures = true; break;
                    case 2: nface[facevert] = negative ? (unsigned)(_vn.size() / 3) - index : index - 1; hasNormals = true; break;
                    }
                }
            }
            push();
            if (currentMtl >= 0) mtlData[currentMtl].faceCount += (unsigned)(_f.size() / 3 - nFacesBefore);
        } else if (load_mtl) {
            if (IsCommand(line, "usemtl")) {
                std::string nm = line.size() > 7 ? line.substr(7) : std::string();
                if (nm.empty()) currentMtl = 0;
                else {
                    int found = -1;
                    for (size_t i = 0; i < mtlData.size(); i++) if (mtlData[i].name == nm) { found = (int)i; break; }
                    if (found < 0) { MtlData d; d.name = nm; d.firstFace = (unsigned)(_f.size() / 3); mtlData.push_back(d); found = (int)mtlData.size() - 1; }
                    currentMtl = found;
                }
            }
            if (IsCommand(line, "mtllib")) mtlFiles.push_back(line.size() > 7 ? line.substr(7) : std::string());
        }
        if (rd.eof) break;
    }
    const size_t nf = _f.size() / 3;
    if (nf == 0) { err = "no faces"; return false; } // the reference would go on to crash on an empty BVH
    if ((!_ft.empty() && _ft.size() != _f.size()) || (!_fn.empty() && _fn.size() != _f.size())) {
        err = "faces mix vertex formats (v, v/vt, v//vn) in a way the reference mis-handles";
        return false;
    }
    for (size_t i = 0; i < _f.size(); i++)
        if (_f[i] >= _v.size() / 3) { err = "face vertex index out of range"; return false; }
    for (size_t i = 0; i < _ft.size(); i++)
        if (_ft[i] >= _vt.size() / 3) { err = "face texture index out of range"; return false; }
    for (size_t i = 0; i < _fn.size(); i++)
        if (!_vn.empty() && _fn[i] >= _vn.size() / 3) { err = "face normal index out of range"; return false; }

    m.v = _v; m.vt = _vt; m.vn = _vn;
    m.f.assign(_f.size(), 0);
    m.ft.assign(_vt.empty() ? 0 : _f.size(), 0);
    m.fn.assign(_vn.empty() ? 0 : _f.size(), 0);
    bool useFt = !m.ft.empty() && !_ft.empty(), useFn = !m.fn.empty() && !_fn.empty();
    auto copyFace = [&](size_t dst, size_t src) {
        for (int k = 0; k < 3; k++) {
            m.f[dst * 3 + k] = _f[src * 3 + k];
            if (useFt) m.ft[dst * 3 + k] = _ft[src * 3 + k];
            if (useFn) m.fn[dst * 3 + k] = _fn[src * 3 + k];
        }
    };
    if (load_mtl) { m.mtls.resize(mtlData.size()); m.mcfc.assign(mtlData.size(), 0); }
    if (!mtlData.empty()) { // faces regrouped by material (cyTriMesh.h:461-487)
        size_t fid = 0;
        for (size_t mi = 0; mi < mtlData.size(); mi++) {
            for (size_t i = mtlData[mi].firstFace, j = 0; j < mtlData[mi].faceCount && i < nf; i++)
                if (faceMtlIndex[i] == (int)mi) { copyFace(fid++, i); j++; }
            m.mcfc[mi] = (int)fid;
        }
        if (fid < nf)
            for (size_t i = 0; i < nf; i++)
                if (faceMtlIndex[i] < 0) copyFace(fid++, i);
    } else
        for (size_t i = 0; i < nf; i++) copyFace(i, i);

    if (load_mtl) { // .mtl files (cyTriMesh.h:498-544)
        std::string p = path, dir;
        size_t slash = p.find_last_of("/\\");
        if (slash != std::string::npos) dir = p.substr(0, slash + 1);
        for (auto &lib : mtlFiles) {
            std::string mt;
            if (!ReadWholeFile((dir + lib).c_str(), mt)) continue;
            LineReader mr(mt);
            int id = -1;
            while (mr.Read(line)) {
                if (IsCommand(line, "newmtl")) {
                    std::string nm = line.size() > 7 ? line.substr(7) : std::string();
                    id = -1;
                    for (size_t i = 0; i < mtlData.size(); i++) if (mtlData[i].name == nm) { id = (int)i; break; }
                    if (id >= 0) m.mtls[id].name = CopyFrom(line, 7);
                } else if (id >= 0) {
                    HostMesh::Mtl &M = m.mtls[id];
                    if (IsCommand(line, "Ka")) ReadFloat3(line, M.Ka);
                    else if (IsCommand(line, "Kd")) ReadFloat3(line, M.Kd);
                    else if (IsCommand(line, "Ks")) ReadFloat3(line, M.Ks);
                    else if (IsCommand(line, "Tf")) ReadFloat3(line, M.Tf);
                    else if (IsCommand(line, "Ns")) sscanf(line.c_str() + 2, "%f", &M.Ns);
                    else if (IsCommand(line, "Ni")) sscanf(line.c_str() + 2, "%f", &M.Ni);
                    else if (IsCommand(line, "illum")) sscanf(line.c_str() + 5, "%d", &M.illum);
                    else if (IsCommand(line, "map_Kd")) M.map_Kd = CopyFrom(line, 7);
                    else if (IsCommand(line, "map_Ks")) M.map_Ks = CopyFrom(line, 7);
                }
                if (mr.eof) break;
            }
        }
    }
    m.had_vt = !m.vt.empty();
    if (!m.had_vt) { // the reference crashes; define uvw = 0
        m.vt = {0, 0, 0};
        m.ft.assign(m.f.size(), 0);
    }
    return true;
}

void ComputeNormals(HostMesh &m) // cyTriMesh.h:248-261 (area-weighted, counter-clockwise)
{
    size_t nv = m.v.size() / 3, nf = m.f.size() / 3;
    std::vector<V3> vn(nv, v3(0, 0, 0));
    auto V = [&](uint32_t i) { return v3(m.v[i * 3], m.v[i * 3 + 1], m.v[i * 3 + 2]); };
    m.fn.assign(m.f.size(), 0);
    for (size_t i = 0; i < nf; i++) {
        uint32_t a = m.f[i * 3], b = m.f[i * 3 + 1], c = m.f[i * 3 + 2];
        V3 N = cross(V(b) - V(a), V(c) - V(a));
        vn[a] = vn[a] + N; vn[b] = vn[b] + N; vn[c] = vn[c] + N; // same vertex twice in a face accumulates twice, like +=
        for (int k = 0; k < 3; k++) m.fn[i * 3 + k] = m.f[i * 3 + k];
    }
    m.vn.resize(nv * 3);
    for (size_t i = 0; i < nv; i++) {
        V3 n = normalized(vn[i]);
        m.vn[i * 3] = n.x; m.vn[i * 3 + 1] = n.y; m.vn[i * 3 + 2] = n.z;
    }
}

void ComputeBoundingBox(HostMesh &m) // cyTriMesh.h:229-246
{
    size_t nv = m.v.size() / 3;
    if (!nv) return;
    for (int k = 0; k < 3; k++) m.bound_min[k] = m.bound_max[k] = m.v[k];
    for (size_t i = 1; i < nv; i++)
        for (int k = 0; k < 3; k++) {
            float x = m.v[i * 3 + k];
            if (m.bound_min[k] > x) m.bound_min[k] = x;
            if (m.bound_max[k] < x) m.bound_max[k] = x;
        }
}

// ------------------------------------------------------------------------------------------------
// BVH build (cyBVH.h:122-142, 242-328, 356-375)
// ------------------------------------------------------------------------------------------------
namespace {
struct Bx {
    float b[6];
    Bx() { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }
    void Add(const Bx &o)
    {
        for (int i = 0; i < 3; i++) {
            if (b[i] > o.b[i]) b[i] = o.b[i];
            if (b[i + 3] < o.b[i + 3]) b[i + 3] = o.b[i + 3];
        }
    }
};
struct TempNode {
    int child1 = -1, child2 = -1;
    Bx box;
    unsigned count = 0, offset = 0;
};
struct BvhBuilder {
    const HostMesh &m;
    std::vector<uint32_t> elems;
    std::vector<TempNode> t;
    unsigned maxPer;
    BvhBuilder(const HostMesh &mesh, unsigned mp) : m(mesh), maxPer(mp) {}
    Bx ElemBounds(unsigned i) const
    {
        Bx x;
        const float *p = &m.v[m.f[i * 3] * 3];
        x.b[0] = x.b[3] = p[0]; x.b[1] = x.b[4] = p[1]; x.b[2] = x.b[5] = p[2];
        for (int j = 1; j < 3; j++) {
            const float *q = &m.v[m.f[i * 3 + j] * 3];
            for (int k = 0; k < 3; k++) {
                if (x.b[k] > q[k]) x.b[k] = q[k];
                if (x.b[k + 3] < q[k]) x.b[k + 3] = q[k];
            }
        }
        return x;
    }
    float ElemCenter(unsigned i, int dim) const // cyBVH.h:371-375
    {
        return (m.v[m.f[i * 3] * 3 + dim] + m.v[m.f[i * 3 + 1] * 3 + dim] + m.v[m.f[i * 3 + 2] * 3 + dim]) / 3.0f;
    }
    unsigned MeanSplit(unsigned n, uint32_t *e, const float *box) // cyBVH.h:295-328
    {
        if (n <= maxPer) return 0;
        float d[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
        unsigned sd[3];
        sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
        sd[1] = (sd[0] + 1) % 3;
        sd[2] = (sd[0] + 2) % 3;
        if (d[sd[1]] < d[sd[2]]) std::swap(sd[1], sd[2]);
        unsigned c1 = 0;
        for (int s = 0; s < 3; s++) {
            unsigned dim = sd[s];
            float splitPos = 0.5f * (box[dim] + box[dim + 3]);
            unsigned i = 0, j = n;
            while (i < j) {
                if (ElemCenter(e[i], dim) <= splitPos) i++;
                else { j--; std::swap(e[i], e[j]); }
            }
            if (i < n && i > 0) { c1 = i; break; }
        }
        return c1;
    }
    void Split(int ti) // cyBVH.h:242-278
    {
        uint32_t *e = &elems[t[ti].offset];
        unsigned n = t[ti].count;
        unsigned c1 = MeanSplit(n, e, t[ti].box.b);
        if (c1 == 0 || c1 >= n) {
            if (n > 8) c1 = n / 2; // CY_BVH_MAX_ELEMENT_COUNT
            else return;
        }
        Bx b1, b2;
        for (unsigned i = 0; i < c1; i++) b1.Add(ElemBounds(e[i]));
        for (unsigned i = c1; i < n; i++) b2.Add(ElemBounds(e[i]));
        TempNode n1, n2;
        n1.count = c1; n1.offset = t[ti].offset; n1.box = b1;
        n2.count = n - c1; n2.offset = t[ti].offset + c1; n2.box = b2;
        int i1 = (int)t.size();
        t.push_back(n1);
        int i2 = (int)t.size();
        t.push_back(n2);
        t[ti].child1 = i1; t[ti].child2 = i2;
        Split(i1);
        Split(i2);
    }
    unsigned NumNodes(int ti) const { return t[ti].child1 < 0 ? 1 : 1 + NumNodes(t[ti].child1) + NumNodes(t[ti].child2); }
    unsigned Convert(std::vector<bhrt_bvh_node> &out, unsigned id, int ti, unsigned childIndex, unsigned parent, unsigned depth, unsigned &maxDepth)
    { // cyBVH.h:281-291
        bhrt_bvh_node &o = out[id];
        memcpy(o.b, t[ti].box.b, sizeof o.b);
        o.parent = parent;
        if (depth > maxDepth) maxDepth = depth;
        if (t[ti].child1 < 0) {
            o.data = (t[ti].offset & ((1u << 28) - 1)) | ((t[ti].count - 1) << 28) | (1u << 31);
            return childIndex;
        }
        o.data = childIndex & 0x7fffffffu;
        unsigned next = Convert(out, childIndex, t[ti].child1, childIndex + 2, id, depth + 1, maxDepth);
        return Convert(out, childIndex + 1, t[ti].child2, next, id, depth + 1, maxDepth);
    }
};
} // namespace

void BuildBvh(HostMesh &m, unsigned maxPer)
{
    unsigned nf = (unsigned)(m.f.size() / 3);
    if (maxPer > 8) maxPer = 8;
    BvhBuilder B(m, maxPer);
    B.elems.resize(nf);
    for (unsigned i = 0; i < nf; i++) B.elems[i] = i;
    Bx box;
    for (unsigned i = 0; i < nf; i++) box.Add(B.ElemBounds(i));
    TempNode root;
    root.count = nf; root.offset = 0; root.box = box;
    B.t.reserve(2 * (size_t)nf + 2);
    B.t.push_back(root);
    B.Split(0);
    unsigned n = B.NumNodes(0);
    m.bvh.assign(n + 1, bhrt_bvh_node());
    memset(m.bvh.data(), 0, sizeof(bhrt_bvh_node));
    unsigned maxDepth = 0;
    B.Convert(m.bvh, 1, 0, 2, 0, 0, maxDepth);
    m.bvh_depth = maxDepth;
    m.elems = B.elems;
}

// ------------------------------------------------------------------------------------------------
// LoadSceneXml
// ------------------------------------------------------------------------------------------------
int LoadSceneXml(const char *path, FlatScene &out, std::string &err, int bvh_device)
{
    out.blob.clear();
    out.warnings.clear();
    XmlDocument doc;
    if (!doc.LoadFile(path)) { err = std::string("Failed to load the file \"") + path + "\": " + doc.error; return 1; }
    const XmlElement *xml = doc.FirstChild("xml");
    if (!xml) { err = "No \"xml\" tag found."; return 2; }
    const XmlElement *scene = xml->FirstChild("scene");
    if (!scene) { err = "No \"scene\" tag found."; return 3; }
    const XmlElement *cam = xml->FirstChild("camera");
    if (!cam) { err = "No \"camera\" tag found."; return 4; }

    Loader L;
    L.out = &out;
    L.bvh_device = bvh_device;
    {
        std::string p = path;
        size_t slash = p.find_last_of("/\\");
        L.xml_dir = slash == std::string::npos ? std::string(".") : p.substr(0, slash);
    }
    for (auto &chp : scene->children) { // xmlload.cpp:140-168
        const XmlElement *c = chp.get();
        if (NameIs(c, "background")) { Col col = {1, 1, 1}; ReadColor(c, col); L.background.color = col; L.background.map = L.ReadTexture(c); }
        else if (NameIs(c, "environment")) { Col col = {1, 1, 1}; ReadColor(c, col); L.environment.color = col; L.environment.map = L.ReadTexture(c); }
        else if (NameIs(c, "object")) L.LoadNode(-1, 1, c);
        else if (NameIs(c, "material")) L.LoadMaterial(c);
        else if (NameIs(c, "light")) L.LoadLight(c);
    }
    for (auto &nm : L.node_mtl_list) { // xmlload.cpp:101-107
        int mi = L.FindMaterial(nm.second);
        if (mi >= 0) L.nodes[nm.first].material = mi;
    }

    bhrt_camera C;
    memset(&C, 0, sizeof C);
    { // xmlload.cpp:109-127 + scene.h:513-523
        V3 pos = {0, 0, 0}, dir = {0, 0, -1}, up = {0, 1, 0};
        float fov = 40, focaldist = 1, dof = 0;
        int w = 200, h = 150;
        dir = dir + pos;
        for (auto &chp : cam->children) {
            const XmlElement *c = chp.get();
            if (NameIs(c, "position")) ReadVector(c, pos);
            else if (NameIs(c, "target")) ReadVector(c, dir);
            else if (NameIs(c, "up")) ReadVector(c, up);
            else if (NameIs(c, "fov")) ReadFloat(c, fov);
            else if (NameIs(c, "focaldist")) ReadFloat(c, focaldist);
            else if (NameIs(c, "dof")) ReadFloat(c, dof);
            else if (NameIs(c, "width")) c->QueryInt("value", &w);
            else if (NameIs(c, "height")) c->QueryInt("value", &h);
        }
        dir = dir - pos;
        dir = normalized(dir);
        V3 x = cross(dir, up);
        up = normalized(cross(x, dir));
        if (w <= 0 || h <= 0) { err = "camera width/height must be positive"; return 5; }
        C.pos[0] = pos.x; C.pos[1] = pos.y; C.pos[2] = pos.z;
        C.dir[0] = dir.x; C.dir[1] = dir.y; C.dir[2] = dir.z;
        C.up[0] = up.x; C.up[1] = up.y; C.up[2] = up.z;
        C.fov = fov; C.focaldist = focaldist; C.dof = dof; C.width = w; C.height = h;
        // BeginRender's camera frame, Main.cpp:179-192 (tan in double, PI = 3.14159265)
        float aor = w / (float)h;
        float tan_h_pov = (float)tan(fov / 2 * 3.14159265 / 180.0);
        float l = focaldist;
        float hh = 2 * l * tan_h_pov;
        float ww = aor * hh;
        V3 camZ = -dir, camY = up, camX = cross(camY, camZ);
        V3 topLeft = pos - camZ * l + camY * hh / 2 - camX * ww / 2;
        V3 ddx = camX * ww / (float)w; // int -> float promotion in Vec3 / T
        V3 ddy = camY * hh / (float)h;
        C.top_left[0] = topLeft.x; C.top_left[1] = topLeft.y; C.top_left[2] = topLeft.z;
        C.dd_x[0] = ddx.x; C.dd_x[1] = ddx.y; C.dd_x[2] = ddx.z;
        C.dd_y[0] = ddy.x; C.dd_y[1] = ddy.y; C.dd_y[2] = ddy.z;
    }

    // CalculateLightsIntensity (Main.cpp:116-123): std::sort ascending by Gray(), then a float sum
    std::sort(L.lights.begin(), L.lights.end(), [](const LightData &a, const LightData &b) { return a.Gray() < b.Gray(); });
    float allLight = 0;
    for (auto &l : L.lights) allLight += l.Gray();

    unsigned maxDepth = 0;
    for (auto &n : L.nodes) if ((unsigned)n.depth > maxDepth) maxDepth = (unsigned)n.depth;
    if (maxDepth > BHRT_MAX_NODE_DEPTH) { err = "scene graph deeper than BHRT_MAX_NODE_DEPTH"; return 6; }
    if (!L.fatal.empty()) { err = L.fatal; return 7; }

    // ---------------- flatten
    std::vector<uint8_t> &blob = out.blob;
    blob.assign(sizeof(bhrt_flat_header), 0);
    BlobWriter W(blob);
    bhrt_flat_header H;
    memset(&H, 0, sizeof H);
    H.magic = BHRT_FLAT_MAGIC; H.version = BHRT_FLAT_VERSION;
    H.camera = C;
    StoreTexColor(H.background, L.background);
    StoreTexColor(H.environment, L.environment);
    H.all_light_intensity = allLight;
    H.max_node_depth = maxDepth;

    std::vector<bhrt_node> nodes(L.nodes.size());
    for (size_t i = 0; i < L.nodes.size(); i++) {
        bhrt_node &o = nodes[i];
        memset(&o, 0, sizeof o);
        L.nodes[i].xf.Store(o.xf);
        o.parent = L.nodes[i].parent; o.depth = L.nodes[i].depth; o.obj_type = L.nodes[i].obj_type;
        o.mesh = L.nodes[i].mesh; o.material = L.nodes[i].material; o.subtree_end = L.nodes[i].subtree_end;
    }
    H.n_nodes = (uint32_t)nodes.size();
    H.off_nodes = W.Append(nodes.data(), nodes.size() * sizeof(bhrt_node));

    std::vector<bhrt_mesh> meshes(L.meshes.size());
    for (size_t i = 0; i < L.meshes.size(); i++) {
        const HostMesh &m = *L.meshes[i];
        bhrt_mesh &o = meshes[i];
        memset(&o, 0, sizeof o);
        o.nv = (uint32_t)(m.v.size() / 3); o.nf = (uint32_t)(m.f.size() / 3);
        o.nvn = (uint32_t)(m.vn.size() / 3); o.nvt = (uint32_t)(m.vt.size() / 3);
        o.n_bvh_nodes = (uint32_t)m.bvh.size(); o.bvh_depth = m.bvh_depth;
        o.off_v = W.Append(m.v.data(), m.v.size() * 4);
        o.off_vn = W.Append(m.vn.data(), m.vn.size() * 4);
        o.off_vt = W.Append(m.vt.data(), m.vt.size() * 4);
        o.off_f = W.Append(m.f.data(), m.f.size() * 4);
        o.off_fn = W.Append(m.fn.data(), m.fn.size() * 4);
        o.off_ft = W.Append(m.ft.data(), m.ft.size() * 4);
        o.off_bvh = W.Append(m.bvh.data(), m.bvh.size() * sizeof(bhrt_bvh_node));
        o.off_elems = W.Append(m.elems.data(), m.elems.size() * 4);
        std::vector<bhrt_bvh_node> dbvh;
        { // breadth-first renumbering for the device (children stay adjacent, first child even, like cyBVH.h:281-291)
            std::vector<bhrt_bvh_node> dn(m.bvh.size());
            memset(dn.data(), 0, sizeof(bhrt_bvh_node) * dn.size());
            std::vector<uint32_t> order = {1};     // old ids in BFS order
            std::vector<uint32_t> newid(m.bvh.size(), 0);
            newid[1] = 1;
            uint32_t next = 2;
            for (size_t k = 0; k < order.size(); k++) {
                const uint32_t oldn = order[k];
                const bhrt_bvh_node &src = m.bvh[oldn];
                bhrt_bvh_node &dst = dn[newid[oldn]];
                dst = src;
                dst.parent = src.parent ? newid[src.parent] : 0;
                if (!(src.data & 0x80000000u)) {
                    const uint32_t c1 = src.data & 0x7fffffffu;
                    newid[c1] = next; newid[c1 + 1] = next + 1;
                    dst.data = next;
                    next += 2;
                    order.push_back(c1); order.push_back(c1 + 1);
                }
            }
            dbvh.swap(dn);
        }
        std::vector<bhrt_tri> tris(o.nf);
        for (uint32_t f = 0; f < o.nf; f++) {
            memset(&tris[f], 0, sizeof(bhrt_tri));
            const float *pa = &m.v[m.f[f * 3] * 3], *pb = &m.v[m.f[f * 3 + 1] * 3], *pc = &m.v[m.f[f * 3 + 2] * 3];
            // TriObj.cpp:79,85,89: the same float operations the reference performs per intersection test
            const V3 a = v3(pa[0], pa[1], pa[2]), b = v3(pb[0], pb[1], pb[2]), c = v3(pc[0], pc[1], pc[2]);
            const V3 vN = cross(b - a, c - a);
            tris[f].vN[0] = vN.x; tris[f].vN[1] = vN.y; tris[f].vN[2] = vN.z;
            tris[f].vN_len = length(vN);
            tris[f].vN_dot_v0 = dot(vN, a);
            // TriObj.cpp:105-131: projection plane from the largest |vN| component, first comparison that holds
            const float ax = fabsf(vN.x), ay = fabsf(vN.y), az = fabsf(vN.z);
            uint32_t axis = 3;
            if (ax >= ay && ax >= az) axis = 0;
            else if (ay >= ax && ay >= az) axis = 1;
            else if (az >= ay && az >= ax) axis = 2;
            const int iu = axis == 0 ? 1 : 0, iv = axis == 2 ? 1 : 2;
            if (axis < 3) {
                tris[f].p0[0] = pa[iu]; tris[f].p0[1] = pa[iv];
                tris[f].p1[0] = pb[iu]; tris[f].p1[1] = pb[iv];
                tris[f].p2[0] = pc[iu]; tris[f].p2[1] = pc[iv];
            }
            tris[f].face_axis = f | (axis << 30);
        }
        o.off_tris = W.Append(tris.data(), tris.size() * sizeof(bhrt_tri));
        std::vector<bhrt_tri> leaf_tris(o.nf);
        for (uint32_t k = 0; k < o.nf; k++) leaf_tris[k] = tris[m.elems[k]];
        o.off_leaf_tris = W.Append(leaf_tris.data(), leaf_tris.size() * sizeof(bhrt_tri));
        memcpy(o.bound_min, m.bound_min, 12);
        memcpy(o.bound_max, m.bound_max, 12);
        {
            std::vector<uint32_t> dparent;
            ComputeLeafSkip(m, leaf_tris, dbvh, dparent, o); // the leaves' `parent` words become their cones; the links move to an array of their own
            o.off_dbvh = W.Append(dbvh.data(), dbvh.size() * sizeof(bhrt_bvh_node));
            o.off_dparent = W.Append(dparent.data(), dparent.size() * sizeof(uint32_t));
        }
        o.bvh_nested = 1; // see bhrt_flat.h: checked, not assumed
        for (uint32_t n = 1; n < o.n_bvh_nodes && o.bvh_nested; n++) {
            const bhrt_bvh_node &pn = m.bvh[n];
            if (pn.data & 0x80000000u) continue;
            for (uint32_t c = pn.data & 0x7fffffffu, e = c + 2; c < e && c < o.n_bvh_nodes; c++)
                for (int k = 0; k < 3; k++)
                    if (!(m.bvh[c].b[k] >= pn.b[k] && m.bvh[c].b[k + 3] <= pn.b[k + 3])) o.bvh_nested = 0;
        }
    }
    H.n_meshes = (uint32_t)meshes.size();

    std::vector<bhrt_texture> textures(L.textures.size());
    for (size_t i = 0; i < L.textures.size(); i++) {
        bhrt_texture &o = textures[i];
        memset(&o, 0, sizeof o);
        const TexData &t = L.textures[i];
        o.type = t.type; o.width = t.w; o.height = t.h;
        o.color1[0] = t.c1.r; o.color1[1] = t.c1.g; o.color1[2] = t.c1.b;
        o.color2[0] = t.c2.r; o.color2[1] = t.c2.g; o.color2[2] = t.c2.b;
        o.off_data = W.Append(t.rgb.data(), t.rgb.size());
    }
    H.n_textures = (uint32_t)textures.size();
    H.off_textures = W.Append(textures.data(), textures.size() * sizeof(bhrt_texture));

    std::vector<bhrt_texmap> texmaps(L.texmaps.size());
    for (size_t i = 0; i < L.texmaps.size(); i++) {
        memset(&texmaps[i], 0, sizeof(bhrt_texmap));
        L.texmaps[i].xf.Store(texmaps[i].xf);
        texmaps[i].texture = L.texmaps[i].texture;
    }
    H.n_texmaps = (uint32_t)texmaps.size();
    H.off_texmaps = W.Append(texmaps.data(), texmaps.size() * sizeof(bhrt_texmap));

    std::vector<bhrt_material> mats(L.materials.size());
    for (size_t i = 0; i < L.materials.size(); i++) {
        bhrt_material &o = mats[i];
        memset(&o, 0, sizeof o);
        const MaterialData &m = L.materials[i];
        o.kind = m.kind;
        StoreTexColor(o.diffuse, m.diffuse); StoreTexColor(o.specular, m.specular); StoreTexColor(o.refraction, m.refraction);
        o.glossiness = m.glossiness;
        o.absorption[0] = m.absorption.r; o.absorption[1] = m.absorption.g; o.absorption[2] = m.absorption.b;
        o.ior = m.ior; o.refraction_glossiness = m.refraction_glossiness;
    }
    H.n_materials = (uint32_t)mats.size();
    H.off_materials = W.Append(mats.data(), mats.size() * sizeof(bhrt_material));

    std::vector<bhrt_light> lights(L.lights.size());
    for (size_t i = 0; i < L.lights.size(); i++) {
        bhrt_light &o = lights[i];
        memset(&o, 0, sizeof o);
        o.type = L.lights[i].type;
        o.intensity[0] = L.lights[i].intensity.r; o.intensity[1] = L.lights[i].intensity.g; o.intensity[2] = L.lights[i].intensity.b;
        o.vec[0] = L.lights[i].vec.x; o.vec[1] = L.lights[i].vec.y; o.vec[2] = L.lights[i].vec.z;
        o.size = L.lights[i].size;
    }
    H.n_lights = (uint32_t)lights.size();
    H.off_lights = W.Append(lights.data(), lights.size() * sizeof(bhrt_light));
    H.off_meshes = W.Append(meshes.data(), meshes.size() * sizeof(bhrt_mesh));
    while (blob.size() % 16) blob.push_back(0);
    H.total_bytes = blob.size();
    memcpy(blob.data(), &H, sizeof H);
    return 0;
}

} // namespace bhrt
