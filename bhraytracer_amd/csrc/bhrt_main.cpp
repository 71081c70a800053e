// bhrt_main.cpp — C++ host program above the C ABI: the headless equivalent of the reference's main()
// (Main.cpp:418-431: LoadScene -> ShowViewport -> BeginRender -> SaveImages; the seam viewport.cpp:425-449 calls BeginRender
// synchronously), with the reference's compile-time constants (scene path Main.cpp:423, output path Main.cpp:416, PT_SampleCount
// :141, GIBounceCount :130, INTERNAL_REFLECTION_BOUNCE :41) as command-line options.
//
//   bhrt render <scene.xml> [-o out.png] [--spp N] [--gi N] [--bounces N] [--seed S] [--no-jitter] [--no-gamma]
//               [--device D | --gpus N [--rehearse]] [--rank R --world N] [--tile T] [--radiance out.f32] [--leaf-skip] [--photon-exact]
//               [--photons N] [--photon-file map.dat] [--photon-out map.dat]     (USE_PhotonMap, Main.cpp:51,53,194,383)
//   bhrt info   <scene.xml>
//
// --gpus N: ONE process drives N GPUs of the node (the reference's one process drives 16 OpenMP threads, Main.cpp:422): the
// image is cut into interleaved tile x tile squares (tile t -> GPU t mod N), one host thread per GPU calls
// bhrt_render_dev for its tiles, packs them (bhrt_tiles_pack_dev), ONE ncclAllGather over the RCCL communicator of the N devices
// (xGMI) moves every GPU's block to all of them, bhrt_tiles_unpack_dev rebuilds the frame, GPU 0's copy is saved.  The caustic
// photon map is emitted in disjoint emission-index ranges on the N GPUs (bhrt_photon_emit_range) and installed on every one
// (bhrt_photon_install): the same map as on one GPU.  --gpus 1 runs the same code over a one-device communicator.
// --gpus N --rehearse: the same N threads, rendezvous points, per-rank renders, packs and unpacks with every rank on device 0 and the
// ncclAllGather replaced by N device-to-device copies — what a one-GPU box can run of the N > 1 control flow (tests/test_cli.py).
// Without --gpus: one device, no RCCL involved (--rank / --world then render that rank's tiles only, for process-per-GPU launchers).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "bhrt.h"

static int fail(const char *what)
{
    fprintf(stderr, "bhrt: %s: %s\n", what, bhrt_last_error());
    return 1;
}
#define HOST_CHECK(expr, what)                                                                  \
    do {                                                                                        \
        if ((expr) != 0) { fprintf(stderr, "bhrt: GPU %d: %s failed (%s)\n", r, what, #expr); \
            failed.store(true); return; }                                                       \
    } while (0)

struct Args {
    std::string scene, out = "out.png", radiance_out, photon_file, photon_out;
    uint32_t photons = 0;
    bhrt_opts o;
    int device = 0, gpus = 0;
    bool rehearse = false; // --rehearse: the N ranks of --gpus N all on device 0, the all-gather as N device-to-device copies (no RCCL)
};

// Everything render_multi owns besides the caller's scene: released on every way out (the early returns included).
struct MultiGuard {
    std::vector<ncclComm_t> comms;
    std::vector<bhrt_scene *> clones; // scenes[1..N-1]
    bool aborted = false;
    ~MultiGuard()
    {
        for (ncclComm_t c : comms)
            if (c) { if (aborted) ncclCommAbort(c); else ncclCommDestroy(c); }
        for (bhrt_scene *s : clones)
            if (s) bhrt_scene_free(s);
    }
};

// BeginRender over N GPUs of this node; rgb / rad: the whole frame on the host (rad may be empty)
static int render_multi(bhrt_scene *first, const Args &A, const bhrt_info &info, std::vector<uint8_t> &rgb, std::vector<float> &rad, std::vector<bhrt_stats> &stats,
                        double &gather_seconds)
{
    const int N = A.gpus, W = info.width, H = info.height, tile = A.o.tile_size > 0 ? A.o.tile_size : 32;
    int have = 0;
    if (bhrt_device_count(&have) || have < (A.rehearse ? 1 : N)) { fprintf(stderr, "bhrt: --gpus %d but %d device(s) visible\n", N, have); return 1; }
    std::vector<int> devs(N);
    for (int r = 0; r < N; r++) devs[r] = A.rehearse ? 0 : r;
    MultiGuard G;
    if (!A.rehearse) {
        G.comms.assign(N, nullptr);
        if (ncclCommInitAll(G.comms.data(), N, devs.data()) != ncclSuccess) { fprintf(stderr, "bhrt: ncclCommInitAll over %d devices failed\n", N); return 1; }
    }
    std::vector<bhrt_scene *> scenes(N, nullptr);
    scenes[0] = first;
    G.clones.assign(N, nullptr);
    for (int r = 1; r < N; r++) {
        if (bhrt_scene_clone(first, &scenes[r])) return fail("scene clone");
        G.clones[r] = scenes[r];
    }
    const size_t bb = bhrt_tiles_block_bytes(W, H, tile, N), npx = (size_t)W * H;
    stats.assign(N, bhrt_stats());
    std::atomic<bool> failed(false);

    // ---- caustic photon map over the N GPUs (BuildCausticPhotonMap, Main.cpp:342-386): batches of 2^20 emissions — the batch of the
    // single-GPU build, cut like bhraytracer_amd/dist.py::photon_build_sharded cuts it (slice r = blocks [256 B r / N, 256 B (r + 1) / N) of
    // the batch's B = 4096 blocks of 256), so that every N stops after the same number of emissions when a scene runs out of its emission
    // budget —, GPU r emits the r-th slice of every batch, the host strings the slices together in emission order and every GPU installs the
    // first `photons` records.
    if (!A.photon_file.empty()) {
        for (int r = 0; r < N; r++)
            if (bhrt_scene_upload(scenes[r], devs[r]) || bhrt_photon_import(scenes[r], A.photon_file.c_str(), 0)) return fail("photon import");
    } else if (A.photons) {
        const uint32_t batch = 1u << 20, blocks = batch / 256;
        if ((uint32_t)N > blocks) { fprintf(stderr, "bhrt: --gpus %d: more ranks than emission blocks per batch\n", N); return 1; }
        std::vector<uint8_t> kept;
        uint64_t e0 = 0, total = 0;
        const uint64_t budget = (uint64_t)A.photons * 4096ull + (1ull << 24);
        while (total < A.photons && e0 < budget) {
            std::vector<std::vector<uint8_t>> part(N);
            std::vector<uint32_t> cnt(N, 0);
            std::vector<std::thread> th;
            for (int r = 0; r < N; r++)
                th.emplace_back([&, r]() {
                    HOST_CHECK(bhrt_scene_upload(scenes[r], devs[r]), "upload");
                    const uint32_t lo = 256u * (uint32_t)((uint64_t)blocks * (uint64_t)r / (uint64_t)N), hi = 256u * (uint32_t)((uint64_t)blocks * (uint64_t)(r + 1) / (uint64_t)N);
                    uint32_t cap = std::max(4 * (hi - lo), 4096u);
                    for (int attempt = 0; attempt < 2; attempt++) { // "photons_out too small" reports the size it needs
                        part[r].resize((size_t)cap * 24);
                        if (bhrt_photon_emit_range(scenes[r], &A.o, 0, e0 + lo, hi - lo, part[r].data(), cap, &cnt[r]) == 0) return;
                        if (cnt[r] <= cap) break;
                        cap = cnt[r];
                    }
                    fprintf(stderr, "bhrt: GPU %d: photon emission: %s\n", r, bhrt_last_error());
                    failed.store(true);
                });
            for (auto &t : th) t.join();
            if (failed.load()) return 1;
            for (int r = 0; r < N; r++) { kept.insert(kept.end(), part[r].begin(), part[r].begin() + (size_t)cnt[r] * 24); total += cnt[r]; }
            e0 += batch;
        }
        if (total == 0) { fprintf(stderr, "bhrt: photon map: no photon reached a photon surface\n"); return 1; }
        const uint32_t n = (uint32_t)std::min<uint64_t>(total, A.photons);
        for (int r = 0; r < N; r++)
            if (bhrt_photon_install(scenes[r], kept.data(), n)) return fail("photon install");
        printf("caustic photon map: %u photons from %llu emissions on %d GPU(s)\n", n, (unsigned long long)e0, N);
    }
    if ((A.o.photon_map || A.photons || !A.photon_file.empty()) && !A.photon_out.empty() && bhrt_photon_export(scenes[0], A.photon_out.c_str())) return fail("photon export");

    // ---- the frame: one host thread per GPU.  The threads agree on failure at three rendezvous points — after set-up, after the render, after
    // the pack — so either all of them enter the collective or none does: a rank missing from an all-gather would hang the others.  Inside the
    // collective a thread never blocks in a stream synchronise: it polls its stream and the failure flag, and when a peer has failed it
    // aborts its communicator (ncclCommAbort) and leaves.
    std::vector<double> gather_s(N, 0.0);
    std::vector<uint8_t *> mine_of(N, nullptr); // --rehearse: where every rank's packed block lies (all on device 0)
    std::atomic<int> ready(0), rendered(0), packed(0), exchanged(0);
    auto rendezvous = [&](std::atomic<int> &c) { c.fetch_add(1); while (c.load() < N) std::this_thread::yield(); return !failed.load(); };
    std::vector<std::thread> th;
    for (int r = 0; r < N; r++)
        th.emplace_back([&, r]() {
            uint8_t *d_rgb = nullptr, *d_mine = nullptr, *d_all = nullptr;
            float *d_rad = nullptr;
            hipStream_t s = nullptr;
            auto setup = [&]() {
                HOST_CHECK(bhrt_scene_upload(scenes[r], devs[r]), "upload");
                HOST_CHECK(hipSetDevice(devs[r]), "hipSetDevice");
                HOST_CHECK(hipStreamCreate(&s), "stream");
                HOST_CHECK(hipMalloc(&d_rgb, npx * 3), "hipMalloc");
                HOST_CHECK(hipMalloc(&d_rad, npx * 3 * sizeof(float)), "hipMalloc");
                HOST_CHECK(hipMalloc(&d_mine, bb), "hipMalloc");
                HOST_CHECK(hipMalloc(&d_all, bb * N), "hipMalloc");
                HOST_CHECK(hipMemsetAsync(d_rgb, 0, npx * 3, s), "memset");
                HOST_CHECK(hipMemsetAsync(d_rad, 0, npx * 3 * sizeof(float), s), "memset");
                HOST_CHECK(hipStreamSynchronize(s), "sync");
            };
            setup();
            auto release = [&]() {
                (void)hipFree(d_rgb); (void)hipFree(d_rad); (void)hipFree(d_mine); (void)hipFree(d_all);
                if (s) (void)hipStreamDestroy(s);
            };
            if (!rendezvous(ready)) { release(); return; }
            bhrt_opts o = A.o;
            o.rank = r; o.world_size = N; o.tile_size = tile;
            if (A.photons || !A.photon_file.empty()) o.photon_map = 1;
            if (bhrt_render_dev(scenes[r], &o, d_rgb, d_rad, &stats[r], nullptr)) { fprintf(stderr, "bhrt: GPU %d: BeginRender: %s\n", r, bhrt_last_error()); failed.store(true); }
            if (!rendezvous(rendered)) { release(); return; }
            const auto t0 = std::chrono::steady_clock::now();
            auto pack = [&]() {
                HOST_CHECK(bhrt_tiles_pack_dev(d_rgb, d_rad, W, H, tile, r, N, d_mine, s), "pack");
                HOST_CHECK(hipStreamSynchronize(s), "sync");
                mine_of[r] = d_mine;
            };
            pack();
            if (getenv("BHRT_TEST_FAIL_PACK") && atoi(getenv("BHRT_TEST_FAIL_PACK")) == r) { fprintf(stderr, "bhrt: GPU %d: pack failed (test knob)\n", r); failed.store(true); }
            if (!rendezvous(packed)) { release(); return; } // a rank whose pack failed keeps everybody out of the collective
            // waits for the stream without blocking in it: a peer's failure is seen within a poll
            auto wait_stream = [&]() -> bool {
                for (;;) {
                    const hipError_t q = hipStreamQuery(s);
                    if (q == hipSuccess) return true;
                    if (q != hipErrorNotReady) { fprintf(stderr, "bhrt: GPU %d: stream: %s\n", r, hipGetErrorString(q)); failed.store(true); }
                    if (failed.load()) {
                        if (!A.rehearse) { ncclCommAbort(G.comms[r]); G.comms[r] = nullptr; }
                        return false;
                    }
                    std::this_thread::yield();
                }
            };
            auto exchange = [&]() {
                if (A.rehearse) { // the all-gather of N ranks on ONE device: every rank copies every block into its own receive buffer
                    for (int k = 0; k < N; k++) HOST_CHECK(hipMemcpyAsync(d_all + (size_t)k * bb, mine_of[k], bb, hipMemcpyDeviceToDevice, s), "copy of a peer's block");
                } else if (ncclAllGather(d_mine, d_all, bb, ncclUint8, G.comms[r], s) != ncclSuccess) {
                    fprintf(stderr, "bhrt: GPU %d: ncclAllGather failed\n", r);
                    failed.store(true);
                }
                if (!wait_stream()) return;
                HOST_CHECK(bhrt_tiles_unpack_dev(d_all, W, H, tile, N, d_rgb, d_rad, s), "unpack");
                HOST_CHECK(hipStreamSynchronize(s), "sync");
                gather_s[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (r == 0) {
                    HOST_CHECK(hipMemcpy(rgb.data(), d_rgb, npx * 3, hipMemcpyDeviceToHost), "copy");
                    if (!rad.empty()) HOST_CHECK(hipMemcpy(rad.data(), d_rad, npx * 3 * sizeof(float), hipMemcpyDeviceToHost), "copy");
                }
            };
            exchange();
            rendezvous(exchanged); // --rehearse: nobody frees a block a peer may still be copying
            release();
        });
    for (auto &t : th) t.join();
    gather_seconds = *std::max_element(gather_s.begin(), gather_s.end());
    G.aborted = failed.load();
    return failed.load() ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc < 3 || (strcmp(argv[1], "render") && strcmp(argv[1], "info"))) {
        fprintf(stderr, "usage: bhrt render|info <scene.xml> [options]\n");
        return 2;
    }
    const bool render = !strcmp(argv[1], "render");
    Args A;
    A.scene = argv[2];
    bhrt_opts &o = A.o;
    bhrt_default_opts(&o);
    for (int a = 3; a < argc; a++) {
        std::string s = argv[a];
        auto next = [&]() -> const char * { if (a + 1 >= argc) { fprintf(stderr, "bhrt: %s needs a value\n", s.c_str()); exit(2); } return argv[++a]; };
        if (s == "-o") A.out = next();
        else if (s == "--spp") o.spp = atoi(next());
        else if (s == "--gi") o.gi_bounces = atoi(next());
        else if (s == "--bounces") o.internal_bounces = atoi(next());
        else if (s == "--seed") o.seed = (uint32_t)strtoul(next(), nullptr, 10);
        else if (s == "--no-jitter") o.jitter = 0;
        else if (s == "--no-gamma") o.gamma = 0;
        else if (s == "--leaf-skip") o.leaf_skip = 1;
        else if (s == "--photon-exact") o.photon_exact = 1;
        else if (s == "--device") A.device = atoi(next());
        else if (s == "--gpus") A.gpus = atoi(next());
        else if (s == "--rehearse") A.rehearse = true;
        else if (s == "--rank") o.rank = atoi(next());
        else if (s == "--world") o.world_size = atoi(next());
        else if (s == "--tile") o.tile_size = atoi(next());
        else if (s == "--radiance") A.radiance_out = next();
        else if (s == "--photons") A.photons = (uint32_t)strtoul(next(), nullptr, 10);
        else if (s == "--photon-file") A.photon_file = next();
        else if (s == "--photon-out") A.photon_out = next();
        else { fprintf(stderr, "bhrt: unknown option %s\n", s.c_str()); return 2; }
    }
    if (A.gpus < 0 || A.gpus > 64 || (A.gpus > 0 && (o.rank != 0 || o.world_size != 1))) { fprintf(stderr, "bhrt: --gpus N drives all N ranks itself (no --rank / --world)\n"); return 2; }
    bhrt_scene *scene = nullptr;
    if (bhrt_scene_load_xml(A.scene.c_str(), &scene)) return fail("LoadScene");
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
    std::vector<uint8_t> rgb((size_t)info.width * info.height * 3, 0);
    std::vector<float> rad(A.radiance_out.empty() ? 0 : (size_t)info.width * info.height * 3, 0.f);
    bhrt_stats st;
    memset(&st, 0, sizeof st);
    if (A.gpus > 0) {
        std::vector<bhrt_stats> per;
        double gather_s = 0;
        const auto t0 = std::chrono::steady_clock::now();
        if (render_multi(scene, A, info, rgb, rad, per, gather_s)) return 1;
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double slowest = 0;
        for (int r = 0; r < A.gpus; r++) {
            st.closest_rays += per[r].closest_rays; st.shadow_rays += per[r].shadow_rays; st.camera_samples += per[r].camera_samples;
            st.wave_iterations = std::max(st.wave_iterations, per[r].wave_iterations); st.passes = std::max(st.passes, per[r].passes);
            slowest = std::max(slowest, per[r].seconds_total);
            printf("GPU %d: %llu camera samples, %llu rays, %.3f s\n", r, (unsigned long long)per[r].camera_samples,
                   (unsigned long long)(per[r].closest_rays + per[r].shadow_rays), per[r].seconds_total);
        }
        st.seconds_total = slowest;
        printf("%d GPU(s): frame %.3f s on the slowest GPU, %s tile gather %.4f s, %.3f s wall incl. set-up\n", A.gpus, slowest, A.rehearse ? "rehearsed (device-to-device)" : "RCCL", gather_s, wall);
    } else {
        if (bhrt_scene_upload(scene, A.device)) return fail("upload");
        if (!A.photon_file.empty()) { // a cached photon pass
            if (bhrt_photon_import(scene, A.photon_file.c_str(), 0)) return fail("photon import");
            o.photon_map = 1;
        } else if (A.photons) { // BuildCausticPhotonMap, Main.cpp:194
            uint32_t stored = 0;
            if (bhrt_photon_build(scene, &o, A.photons, &stored)) return fail("BuildCausticPhotonMap");
            printf("caustic photon map: %u photons\n", stored);
            o.photon_map = 1;
        }
        if (o.photon_map && !A.photon_out.empty() && bhrt_photon_export(scene, A.photon_out.c_str())) return fail("photon export");
        if (bhrt_render(scene, &o, rgb.data(), rad.empty() ? nullptr : rad.data(), &st)) return fail("BeginRender");
    }
    const double rays = (double)st.closest_rays + (double)st.shadow_rays;
    printf("rendered %llu camera samples, %.0f rays (%llu closest + %llu shadow), %u wave steps in %u pass(es): %.3f s, %.1f Mrays/s\n",
           (unsigned long long)st.camera_samples, rays, (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays, st.wave_iterations,
           st.passes, st.seconds_total, rays / st.seconds_total / 1e6);
    if (bhrt_save_png(A.out.c_str(), rgb.data(), info.width, info.height)) return fail("SaveImage");
    if (!A.radiance_out.empty()) {
        FILE *fp = fopen(A.radiance_out.c_str(), "wb");
        if (!fp) { fprintf(stderr, "bhrt: cannot write %s\n", A.radiance_out.c_str()); return 1; }
        fwrite(rad.data(), sizeof(float), rad.size(), fp);
        fclose(fp);
    }
    printf("wrote %s\n", A.out.c_str());
    bhrt_scene_free(scene);
    return 0;
}
