// scene_host.h — host front-end: XML scene -> flattened scene blob (include/bhrt_flat.h).
// Mirrors the reference's xmlload surface (xmlload.cpp:65-582) with its own reader.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "bhrt_flat.h"

namespace bhrt {

struct FlatScene {
    std::vector<uint8_t> blob;
    std::vector<std::string> warnings; // the reference printf()s and carries on (xmlload.cpp:212-214,570-573)
    const bhrt_flat_header *hdr() const { return reinterpret_cast<const bhrt_flat_header *>(blob.data()); }
};

// Returns 0 on success (like LoadScene returning 1), non-zero + err on failure.
// bvh_device >= 0: the mesh BVHs are built on that HIP device (bvh_build.hip) instead of by BuildBvh below; same result.
int LoadSceneXml(const char *path, FlatScene &out, std::string &err, int bvh_device = -1);

// OBJ mesh -> arrays, same face/index rules as cyTriMesh::LoadFromFileObj (cyTriMesh.h:263-547)
struct HostMesh {
    std::vector<float> v, vn, vt;       // xyz triples
    std::vector<uint32_t> f, fn, ft;    // index triples
    struct Mtl {
        std::string name;
        float Ka[3] = {0, 0, 0}, Kd[3] = {1, 1, 1}, Ks[3] = {0, 0, 0}, Tf[3] = {0, 0, 0};
        float Ns = 0, Ni = 1;
        int illum = 2;
        std::string map_Kd, map_Ks;
    };
    std::vector<Mtl> mtls;
    std::vector<int> mcfc; // material cumulative face count
    float bound_min[3] = {1, 1, 1}, bound_max[3] = {0, 0, 0};
    // BVH (cyBVH.h:122-142), node 0 unused, root = 1
    std::vector<bhrt_bvh_node> bvh;
    std::vector<uint32_t> elems;
    uint32_t bvh_depth = 0;
    bool had_vt = true;
};
bool LoadObj(const char *path, bool load_mtl, HostMesh &m, std::string &err);
void ComputeNormals(HostMesh &m);     // cyTriMesh.h:248-261
void ComputeBoundingBox(HostMesh &m); // cyTriMesh.h:229-246
void BuildBvh(HostMesh &m, unsigned max_elems_per_node = 4); // cyBVH.h:122-142, objects.h:59
int BuildBvhDevice(HostMesh &m, unsigned max_elems_per_node, int device); // the same tree from the device build (bvh_build.hip); 0 = ok

} // namespace bhrt
