// bhrt_bridge.cpp — the drop-in binding of INTEGRATION.md as a build product: the reference program with the bodies of
// BeginRender / StopRender (Main.cpp:178-245) replaced by calls into libbhrt.so.  Everything else is the reference's own code,
// compiled where it lies: its Main.cpp is included below as text (globals, LoadScene's declaration, recursive(), SaveImages(),
// main()), its other translation units are the objects of oracle/_ref, and viewport_stub.cpp stands in for the GLUT UI exactly as
// in ref_harness — ShowViewport() calls BeginRender(), which is the one defined HERE.
//
// Test infrastructure (oracle/Makefile target `bridge`, output oracle/_ref/bhrt_bridge; tests/test_bridge.py): proves that the
// C ABI of include/bhrt.h is enough to replace the reference's render path without touching any other line of it.
//
//   bhrt_bridge                              the reference's main() as it stands (scene and output paths of Main.cpp:416,423)
//   bhrt_bridge scene.xml out.png [spp]      the same program on another scene file (the reference hard-codes its paths)
// Exit code: 0, or the bhrt_status BeginRender() ended with (BHRT_ERR_NO_DEVICE = 4 on a machine without a GPU).
#define BeginRender BeginRender_ref   /* the reference's bodies stay in the binary under other names, unused */
#define StopRender StopRender_ref
#define main reference_main
#include "Main.cpp"
#undef BeginRender
#undef StopRender
#undef main

#include "bhrt.h"
#include <stdio.h>
#include <stdlib.h>

static bhrt_scene *g_scene = nullptr;
static const char *g_sceneFile = "Resource/Data/proj12_backfaceTest.xml";   // Main.cpp:423
static const char *g_outFile = "Resource/Result/proj12_backfaceTest.png";   // Main.cpp:416
static int g_spp = 0;                                                       // 0: PT_SampleCount (Main.cpp:128) = bhrt_default_opts
static int g_status = BHRT_OK;

void BeginRender()   // called by the viewport on key press (viewport.cpp:425-449); here by viewport_stub.cpp::ShowViewport
{
    if (!g_scene && (g_status = bhrt_scene_load_xml(g_sceneFile, &g_scene))) { printf("bhrt: %s\n", bhrt_last_error()); return; }
    bhrt_opts o;
    bhrt_default_opts(&o);                 // 32 spp, GI 3, 16 internal bounces, gamma on — the reference's #defines
    if (g_spp > 0) o.spp = g_spp;
    bhrt_stats st;
    // Color24 is three uint8_t (cyColor.h): GetPixels() is exactly the W*H*3 row-major buffer bhrt_render fills
    if ((g_status = bhrt_render(g_scene, &o, &renderImage.GetPixels()[0].r, nullptr, &st))) { printf("bhrt: %s\n", bhrt_last_error()); return; }
    renderImage.IncrementNumRenderPixel(renderImage.GetWidth() * renderImage.GetHeight());   // scene.h:576: viewport progress
    printf("%.1f Mrays/s\n", (st.closest_rays + st.shadow_rays) / st.seconds_total / 1e6);
    renderImage.SaveImage(g_outFile);                                                         // Main.cpp:416
}
void StopRender() {}

int main(int argc, char **argv)
{
    if (argc < 3) { reference_main(); return g_status; }   // Main.cpp:418-431 as it stands: LoadScene, ShowViewport -> BeginRender above
    g_sceneFile = argv[1];
    g_outFile = argv[2];
    if (argc > 3) g_spp = atoi(argv[3]);
    omp_set_num_threads(16);
    if (!LoadScene(g_sceneFile)) return BHRT_ERR_IO;       // the reference's loader: camera and renderImage for the viewport's own needs
    ShowViewport();
    if (g_scene) bhrt_scene_free(g_scene);
    return g_status;
}
