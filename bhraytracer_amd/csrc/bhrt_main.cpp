// bhrt_main.cpp — C++ host program above the C ABI: the headless equivalent of the reference's main()
// (Main.cpp:418-431: LoadScene -> ShowViewport -> BeginRender -> SaveImages), with the reference's compile-time
// constants (scene path Main.cpp:423, output path Main.cpp:416, PT_SampleCount :141, GIBounceCount :130,
// INTERNAL_REFLECTION_BOUNCE :41) as command-line options.
//
//   bhrt render <scene.xml> [-o out.png] [--spp N] [--gi N] [--bounces N] [--seed S] [--no-jitter] [--no-gamma]
//               [--device D] [--rank R --world N --tile T] [--radiance out.f32]
//               [--photons N] [--photon-file map.dat] [--photon-out map.dat]     (USE_PhotonMap, Main.cpp:51,53,194,383)
//   bhrt info   <scene.xml>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "bhrt.h"

static int fail(const char *what)
{
    fprintf(stderr, "bhrt: %s: %s\n", what, bhrt_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 3 || (strcmp(argv[1], "render") && strcmp(argv[1], "info"))) {
        fprintf(stderr, "usage: bhrt render|info <scene.xml> [options]\n");
        return 2;
    }
    const bool render = !strcmp(argv[1], "render");
    const char *scene_path = argv[2];
    std::string out = "out.png", radiance_out, photon_file, photon_out;
    uint32_t photons = 0;
    bhrt_opts o;
    bhrt_default_opts(&o);
    int device = 0;
    for (int a = 3; a < argc; a++) {
        std::string s = argv[a];
        auto next = [&]() -> const char * { if (a + 1 >= argc) { fprintf(stderr, "bhrt: %s needs a value\n", s.c_str()); exit(2); } return argv[++a]; };
        if (s == "-o") out = next();
        else if (s == "--spp") o.spp = atoi(next());
        else if (s == "--gi") o.gi_bounces = atoi(next());
        else if (s == "--bounces") o.internal_bounces = atoi(next());
        else if (s == "--seed") o.seed = (uint32_t)strtoul(next(), nullptr, 10);
        else if (s == "--no-jitter") o.jitter = 0;
        else if (s == "--no-gamma") o.gamma = 0;
        else if (s == "--device") device = atoi(next());
        else if (s == "--rank") o.rank = atoi(next());
        else if (s == "--world") o.world_size = atoi(next());
        else if (s == "--tile") o.tile_size = atoi(next());
        else if (s == "--radiance") radiance_out = next();
        else if (s == "--photons") photons = (uint32_t)strtoul(next(), nullptr, 10);
        else if (s == "--photon-file") photon_file = next();
        else if (s == "--photon-out") photon_out = next();
        else { fprintf(stderr, "bhrt: unknown option %s\n", s.c_str()); return 2; }
    }
    bhrt_scene *scene = nullptr;
    if (bhrt_scene_load_xml(scene_path, &scene)) return fail("LoadScene");
    bhrt_info info;
    bhrt_scene_info(scene, &info);
    for (uint32_t i = 0; i < info.n_warnings; i++) {
        const char *w = nullptr;
        if (!bhrt_scene_warning(scene, i, &w)) fprintf(stderr, "bhrt: warning: %s\n", w);
    }
    printf("Render image width: %d\nRender image height: %d\n", info.width, info.height); // Main.cpp:426-427
    printf("nodes %u, meshes %u (%u triangles, %u BVH nodes), materials %u, lights %u, textures %u, scene blob %llu bytes\n", info.n_nodes,
           info.n_meshes, info.n_triangles, info.n_bvh_nodes, info.n_materials, info.n_lights, info.n_textures, (unsigned long long)info.flat_bytes);
    if (!render) { bhrt_scene_free(scene); return 0; }
    if (bhrt_scene_upload(scene, device)) return fail("upload");
    std::vector<uint8_t> rgb((size_t)info.width * info.height * 3, 0);
    std::vector<float> rad(radiance_out.empty() ? 0 : (size_t)info.width * info.height * 3, 0.f);
    if (!photon_file.empty()) { // a cached photon pass
        if (bhrt_photon_import(scene, photon_file.c_str(), 0)) return fail("photon import");
        o.photon_map = 1;
    } else if (photons) { // BuildCausticPhotonMap, Main.cpp:194
        uint32_t stored = 0;
        if (bhrt_photon_build(scene, &o, photons, &stored)) return fail("BuildCausticPhotonMap");
        printf("caustic photon map: %u photons\n", stored);
        o.photon_map = 1;
    }
    if (o.photon_map && !photon_out.empty() && bhrt_photon_export(scene, photon_out.c_str())) return fail("photon export");
    bhrt_stats st;
    if (bhrt_render(scene, &o, rgb.data(), rad.empty() ? nullptr : rad.data(), &st)) return fail("BeginRender");
    const double rays = (double)st.closest_rays + (double)st.shadow_rays;
    printf("rendered %llu camera samples, %.0f rays (%llu closest + %llu shadow), %u wave steps in %u pass(es): %.3f s, %.1f Mrays/s\n",
           (unsigned long long)st.camera_samples, rays, (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays, st.wave_iterations,
           st.passes, st.seconds_total, rays / st.seconds_total / 1e6);
    if (bhrt_save_png(out.c_str(), rgb.data(), info.width, info.height)) return fail("SaveImage");
    if (!radiance_out.empty()) {
        FILE *fp = fopen(radiance_out.c_str(), "wb");
        if (!fp) { fprintf(stderr, "bhrt: cannot write %s\n", radiance_out.c_str()); return 1; }
        fwrite(rad.data(), sizeof(float), rad.size(), fp);
        fclose(fp);
    }
    printf("wrote %s\n", out.c_str());
    bhrt_scene_free(scene);
    return 0;
}
