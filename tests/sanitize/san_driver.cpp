// san_driver.cpp — the CPU sanitizer job (SURVEY.md 5 "Race detection / sanitizers").  TEST INFRASTRUCTURE.
// The host code of the product that parses untrusted input — scene_host.cpp (the xmlload schema, OBJ / MTL rules of cyTriMesh.h:263-547, PPM
// textures, BVH build, flattening, leaf-skip bounds), mini_xml.h, png_io.cpp, photon_host.cpp, capi_host.cpp — and the oracle's restatement
// (oracle/bhrt_oracle.cpp) are compiled with -fsanitize=address,undefined (tests/sanitize/Makefile) into this program; the two device entry
// points the host objects reference are stubbed out below (there is no GPU code in this build).
//   san_driver <file>...      every .xml is loaded through the C ABI (bhrt_scene_load_xml -> info -> warnings -> flat blob -> free); a scene
//                             that loads is also run through the oracle on its own blob (a few pixels, a few rays, a small photon build)
//   san_driver --png <file>   PNG reader on its own
//   san_driver --photons ...  also a 16-photon caustic build in the oracle for every scene that follows
// A malformed input has to end in an error code or a warning, never in a sanitizer report: the program exits 0 and prints one line per
// file; AddressSanitizer / UBSan abort it (-fno-sanitize-recover) with their report on stderr.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "bhrt.h"
#include "bhrt_oracle.h"
#include "png_io.h"
#include "scene_host.h"
#include "scene_internal.h"

namespace bhrt {
struct DeviceState;
int BuildBvhDevice(HostMesh &, unsigned, int) { return 1; }   // bvh_build.hip is not part of this build
void DestroyDeviceState(DeviceState *) {}                      // kernels.hip neither
} // namespace bhrt

static int run_oracle(const void *blob, const bhrt_info &info, bool photons)
{
    oracle_opts o;
    memset(&o, 0, sizeof o);
    o.spp = 1; o.gi_bounces = 1; o.internal_bounces = 2; // small: a mutated scene (coordinates of 2^31, NaN colours) can send every sample through the capped rejection loops
    o.seed = 1; o.rng_mode = ORACLE_RNG_KEYED; o.math_mode = ORACLE_MATH_DEVICE; o.jitter = 1;
    o.x0 = info.width / 2 - 1 < 0 ? 0 : info.width / 2 - 1; o.y0 = info.height / 2 - 1 < 0 ? 0 : info.height / 2 - 1;
    o.x1 = o.x0 + 2 > info.width ? info.width : o.x0 + 2; o.y1 = o.y0 + 2 > info.height ? info.height : o.y0 + 2;
    o.threads = 1;
    const size_t npx = (size_t)(o.x1 - o.x0) * (size_t)(o.y1 - o.y0);
    std::vector<float> samples(npx * o.spp * 3), rad(npx * 3);
    std::vector<uint8_t> rgb(npx * 3);
    oracle_stats st;
    int rc = oracle_render(blob, &o, samples.data(), rad.data(), rgb.data(), &st);
    const float rays[12] = {0, -30, 10, 0, 1, -0.3f, 1e30f, 0, 0, 0, 0, 0};
    int32_t node[2], face[2], front[2];
    float attrs[2 * ORACLE_HIT_FLOATS], tmax[2] = {1.f, 1.f}, vis[2];
    rc |= oracle_trace_closest(blob, rays, 1, 2, node, face, front, attrs);
    rc |= oracle_trace_shadow(blob, rays, tmax, 2, vis);
    if (photons && info.n_lights > 0) { // not for mutated scenes: a scene without a working caustic path runs through the whole emission budget
        std::vector<uint8_t> ph(16 * 24);
        uint32_t ns = 0;
        uint64_t ne = 0;
        (void)oracle_photon_build(blob, &o, 16, ph.data(), &ns, &ne); // a scene without a caustic path reports that: not an error of the job
    }
    return rc;
}

int main(int argc, char **argv)
{
    int n_ok = 0, n_err = 0;
    bool photons = false;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "--photons")) { photons = true; continue; }
        if (!strcmp(argv[a], "--png") && a + 1 < argc) {
            std::vector<uint8_t> px;
            int w = 0, h = 0;
            const bool ok = bhrt::LoadPngRgb(argv[++a], px, w, h);
            printf("%s: png %s %dx%d\n", argv[a], ok ? "ok" : "rejected", w, h);
            ok ? n_ok++ : n_err++;
            continue;
        }
        bhrt_scene *sc = nullptr;
        const int rc = bhrt_scene_load_xml(argv[a], &sc);
        if (rc) { printf("%s: error %d: %s\n", argv[a], rc, bhrt_last_error()); n_err++; continue; }
        bhrt_info info;
        bhrt_scene_info(sc, &info);
        for (uint32_t i = 0; i < info.n_warnings; i++) { const char *w = nullptr; bhrt_scene_warning(sc, i, &w); }
        const void *blob = nullptr;
        uint64_t bytes = 0;
        bhrt_scene_flat(sc, &blob, &bytes);
        const int orc = run_oracle(blob, info, photons);
        printf("%s: ok %dx%d, %u nodes, %u triangles, %u warnings, blob %llu bytes, oracle %s\n", argv[a], info.width, info.height, info.n_nodes, info.n_triangles,
               info.n_warnings, (unsigned long long)bytes, orc ? oracle_last_error() : "ok");
        bhrt_scene_free(sc);
        n_ok++;
    }
    printf("sanitize job: %d loaded, %d rejected, no sanitizer report\n", n_ok, n_err);
    return 0;
}
