#!/usr/bin/env python3
"""bench.py — Mrays/s + wall-clock per frame of the HIP render path on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c3room|c4|c5|c1] [--scaling weak|strong] [--no-configs]

A "step" = one full frame of the workload through the hot path (camera rays -> wavefront
trace/shade steps -> combine -> resolve [-> RCCL framebuffer gather when N > 1]).
Headline workload (N = 1 default) = BASELINE.json configs[2], the largest single-GPU 1080p configuration: tests/scenes/c3_mesh.xml
(100,352-triangle mesh over a textured plane), 1920x1080, 64 spp, GI depth 3, internal bounces 16, keyed RNG seed 0 — the BVH walk
north_star is about dominates it.  The same JSON line carries, under "configs", the other configurations timed the same way with a few
steps each: c2 (BASELINE configs[1]: glass spheres, 16 spp, no mesh), c3room (the closed Cornell room of proj13.xml around the same
mesh), c4 (3840x2160, 32 spp per GPU = 256 spp on 8) and c5 (1 M caustic photons + k-NN gather, 64 spp) — each with its own `roofline`.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); interleaved 32x32 tiles (tile t -> rank t mod N),
one all_gather of the packed tile buffers over xGMI inside the timed region.  --scaling weak (default): spp scaled with N so per-GPU
work is fixed; --scaling strong: BASELINE's own spp (64; 256 for c4) split over the ranks by tiles, total work fixed.
`python bench.py --gpus N` without a torchrun environment starts the N ranks itself.
The scene is resident in HBM before the timed region; outputs stay in HBM.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, spp per GPU, gi)
    "c1": ("tests/scenes/c1_sphere_plane.xml", 640, 480, 1, -1),
    "c2": ("tests/scenes/c2_glass.xml", 1920, 1080, 16, 3),
    "c3": ("tests/scenes/c3_mesh.xml", 1920, 1080, 64, 3),
    # SURVEY.md 8(d) C3, second half: the closed room of Resource/Data/proj13.xml with the same mesh in the teapot's place
    "c3room": ("tests/scenes/c3_room.xml", 1920, 1080, 64, 3),
    # BASELINE config 4: the mesh scene at 3840x2160, 256 spp on 8 GPUs = 32 spp per GPU (weak scaling like every workload here)
    "c4": ("tests/scenes/c4_mesh_4k.xml", 3840, 2160, 32, 3),  # --scaling strong: 256 spp split by tiles (STRONG_SPP)
    # BASELINE config 5: caustic photon map, 1 M photons, k = 1000, r = 0.5 (photon build timed separately, see "photon_build_s")
    "c5": ("tests/scenes/c5_caustics_hd.xml", 1920, 1080, 64, 3),
}
HEADLINE = "c3"
SIDE_CONFIGS = ["c2", "c3room", "c4", "c5"]  # reported under "configs" beside the headline
STRONG_SPP = {"c4": 256}  # --scaling strong: BASELINE's spp for the whole job (default: the workload's own per-GPU figure, e.g. 64 for c3 / c5)
CPU_SAMPLE_SPP = {"c3": 32, "c3room": 2, "c4": 2, "c5": 2}  # cpu_baseline: samples per pixel of the bounded CPU sample (whole frame, fewer samples)
BYTES_PER_CLOSEST_RAY = 56   # 32 B ray read + 24 B hit write
BYTES_PER_SHADOW_RAY = 36    # 32 B read + 4 B visibility write
BYTES_PER_SHADE_VERTEX = 92  # 24 hit + 32 ray read, 12 radiance, 24 next-ray writes
BYTES_PER_PHOTON_VISITED = 24  # 24 B x candidates scanned per query (+ 24 B out per query)
HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TILE = 32


def ensure_assets():
    mesh = os.path.join(ROOT, "tests/scenes/gen/mesh_224.obj")
    if not os.path.exists(mesh):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_mesh
        gen_mesh.generate(mesh, 224)


def cpu_baseline(scene_path, spp, gi, full_spp=None):
    """Reference CPU path on a bounded sample of the same workload — the whole frame at `spp` samples per pixel (of the workload's
    `full_spp`: every pixel, fewer samples; ~10-30 s of CPU work) — on this box's host cores.

    kind "reference": the reference itself (oracle/_ref/ref_harness, compiled from /root/reference in the dev
    container) run as one single-threaded process per row band (its RNG is a process-global); the harness reports its
    own scene-load and render times, `value` uses the slowest band's render time (= the parallel frame time), the
    wall clock incl. process start and scene load is reported beside it;
    kind "port": the oracle (sequential RNG + libm = the mode pinned bit-for-bit to the reference), OpenMP.
    Ray counts come from the oracle (identical control flow)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import bhraytracer_amd as B
    sc = B.Scene(scene_path)
    blob = sc.flat_bytes()
    W, H = sc.width, sc.height
    host_cores = os.cpu_count() or 1
    cores = max(1, min(host_cores, 16))  # the GPU box's CPU share for one GPU is 16
    region = (0, 0, W, H)
    t0 = time.time()
    ro = O.render(blob, W, H, spp, gi=gi, rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM, region=region, threads=cores,
                  want_samples=False)
    port_s = time.time() - t0
    rays = ro["stats"].closest_rays + ro["stats"].shadow_rays
    sample = (f"one whole step: {W}x{H} x {spp} spp ({W * H * spp} camera samples, {rays} rays)" if not full_spp or full_spp == spp else
              f"the whole {W}x{H} frame at {spp} of the workload's {full_spp} samples per pixel ({W * H * spp} camera samples, {rays} rays)")
    out = {"value": rays / port_s / 1e6, "unit": "Mrays/s", "cores": cores, "host_cores": host_cores, "kind": "port", "sample": sample,
           "seconds": port_s}
    harness = os.path.join(ROOT, "oracle/_ref/ref_harness")
    if os.path.exists(harness):
        try:
            tmp = tempfile.mkdtemp(prefix="bhrt_ref_")
            bounds = [H * k // cores for k in range(cores + 1)]
            t0 = time.time()
            procs = []
            for k in range(cores):
                cmd = [harness, os.path.abspath(scene_path), os.path.join(tmp, f"b{k}"), "--spp", str(spp), "--gi", str(gi),
                       "--region", "0", str(bounds[k]), str(W), str(bounds[k + 1]), "render"]
                procs.append(subprocess.Popen(cmd, cwd=os.path.dirname(os.path.abspath(scene_path)),
                                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
            errs = [p.communicate()[1] for p in procs]
            ok = all(p.returncode == 0 for p in procs)
            wall_s = time.time() - t0
            if ok:
                tm = [re.search(r"load ([0-9.]+) s, commands ([0-9.]+) s", e) for e in errs]
                render_s = max(float(m.group(2)) for m in tm) if all(tm) else wall_s
                load_s = max(float(m.group(1)) for m in tm) if all(tm) else 0.0
                out = {"value": rays / render_s / 1e6, "unit": "Mrays/s", "cores": cores, "host_cores": host_cores, "kind": "reference",
                       "sample": sample + f"; {cores} single-threaded reference processes, one row band each",
                       "seconds": render_s, "scene_load_seconds": load_s, "wall_seconds_incl_process_start": wall_s,
                       "value_incl_process_start": rays / wall_s / 1e6, "port_value": rays / port_s / 1e6}
            subprocess.run(["rm", "-rf", tmp])
        except Exception:
            pass
    return out


def pmc_profile(workload):
    """Per-launch HBM traffic (PMC, separate rocprofv3 --pmc passes) and VALU issue fraction of the workload's kernels, as
    committed under profiles/ (tools/pmc_summary.py writes it; measured once per round, not per bench run)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(workload, {})
    except Exception:
        return {}


def run_workload(name, steps, warmup, ctx, photons, samples_per_pass=0, timers=None, extras=False):
    """Times `steps` frames of workload `name` (barrier + synchronize on both sides, max over ranks)."""
    import torch
    import bhraytracer_amd as B
    import bhraytracer_amd.dist as BD
    rank, N, dev, dist, rehearse = ctx["rank"], ctx["N"], ctx["dev"], ctx["dist"], ctx["rehearse"]
    scene_rel, W, H, spp1, gi = WORKLOADS[name]
    scene_path = os.path.join(ROOT, scene_rel)
    sc = B.Scene(scene_path)
    assert (sc.width, sc.height) == (W, H), "scene size differs from the workload table"
    sc.upload(ctx["local_rank"])
    if ctx.get("scaling", "weak") == "strong":  # total work fixed: BASELINE's spp for the whole job, the frame split over the ranks by tiles
        spp = STRONG_SPP.get(name, spp1)
    else:
        spp = spp1 * N  # weak scaling: per-GPU sample count is fixed
    opts = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0, rank=rank, world_size=N, tile_size=TILE)
    opts.samples_per_pass = samples_per_pass
    opts.timers = timers if timers is not None else (0 if name == "c2" else 1)
    photon_build_s = None
    if name == "c5":
        t0 = time.perf_counter()
        if N > 1:  # emission sharded over the ranks by emission-index range, one all_gather per batch, the same map on every rank
            BD.photon_build_sharded(sc, opts, photons, rank, N, device=None if rehearse else dev)
        else:
            sc.photon_build(opts, photons)
        torch.cuda.synchronize()
        photon_build_s = time.perf_counter() - t0
        opts.photon_map = 1

    # Two frame buffers (+ exchange scratch) used alternately: the exchange of frame k (torch's stream) overlaps the
    # rendering of frame k+1 (the library's stream); a buffer is rendered into again only after its last exchange is done.
    n_buf = 2 if N > 1 else 1
    bufs = [(torch.zeros((H, W, 3), dtype=torch.uint8, device=dev), torch.zeros((H, W, 3), dtype=torch.float32, device=dev), {},
             torch.cuda.Event()) for _ in range(n_buf)]
    frame_no = [0]

    def step():
        rgb, rad, scratch, ev = bufs[frame_no[0] % n_buf]
        frame_no[0] += 1
        if N > 1:
            ev.synchronize()  # no-op until the event has been recorded once
        st = sc.render_dev(opts, rgb.data_ptr(), rad.data_ptr())
        if N > 1:
            # pack -> ONE all_gather of byte blocks (RCCL; through the host with gloo when rehearsing on one GPU) -> unpack
            BD.gather_frame_dev(rgb, rad, TILE, rank, N, scratch=scratch, via_host=rehearse)
            ev.record()
        return st

    def sync():
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()  # untimed: first use allocates the wavefront workspace (tens of GB of hipMalloc) — not part of any timed or warmup step
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    agg = None
    for _ in range(steps):
        d = step().as_dict()
        agg = d if agg is None else {k: agg[k] + d[k] for k in d}
    sync()
    elapsed = time.perf_counter() - t0
    rays_local = agg["closest_rays"] + agg["shadow_rays"]
    if N > 1:
        rdev = torch.device("cpu") if rehearse else dev
        tt = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays_local, agg["closest_rays"], agg["camera_samples"]], dtype=torch.float64, device=rdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total, closest_total, samples_total = (float(x) for x in rr.tolist())
    else:
        rays_total, closest_total, samples_total = float(rays_local), float(agg["closest_rays"]), float(agg["camera_samples"])
    # The any-hit kernels of a wave step run BESIDE the next step's closest-hit kernels (second stream, DESIGN.md 4 "Any-hit work beside the pass"):
    # inside the timed region the kernel groups share the GPU, so their HIP-event times overlap and sum to more than the frame.  Two more frames,
    # untimed, with that switched off give every group's time ALONE — what `roofline.frac_alone` is computed from.
    alone = None
    if sc.info.n_meshes > 0 and N == 1 and opts.timers >= 1 and not ctx.get("no_alone"):
        sc.knob("shadow_overlap", 0)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = step().as_dict()
        torch.cuda.synchronize()
        alone = dict(d, ms_per_step=(time.perf_counter() - t0) * 1e3)
        sc.knob("shadow_overlap", 1)
    found_per_query = None
    if name == "c5" and rank == 0 and N == 1:  # one more frame, untimed, with the statistics instantiation of the lane pass (7 % slower: not the timed kernel)
        sc.knob("gather_stats", 1)
        d = step().as_dict()
        torch.cuda.synchronize()
        sc.knob("gather_stats", 0)
        found_per_query = d.get("photon_found", 0) / max(1, d.get("photon_lane_queries", 0))
    extra = None
    if extras and rank == 0:
        # the two lines SURVEY.md 8(d) asks for beside the frame time: the device-to-host copy of the frame (RGB8 + float radiance, pinned
        # destination) and the PNG encode (RenderImage::SaveImage, scene.h:628-644 -> bhrt_save_png), each timed on its own after the run
        import numpy as np
        rgb, rad = bufs[0][0], bufs[0][1]
        h_rgb, h_rad = torch.empty(rgb.shape, dtype=rgb.dtype).pin_memory(), torch.empty(rad.shape, dtype=rad.dtype).pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h_rgb.copy_(rgb, non_blocking=True); h_rad.copy_(rad, non_blocking=True)
        torch.cuda.synchronize()
        d2h_ms = (time.perf_counter() - t0) * 1e3
        with tempfile.TemporaryDirectory() as td:
            t0 = time.perf_counter()
            B.save_png(os.path.join(td, "frame.png"), h_rgb.numpy())
            png_ms = (time.perf_counter() - t0) * 1e3
            png_bytes = os.path.getsize(os.path.join(td, "frame.png"))
        extra = {"d2h_ms": d2h_ms, "d2h_bytes": int(h_rgb.numel() + 4 * h_rad.numel()), "png_encode_ms": png_ms, "png_bytes": png_bytes,
                 "note": "not part of ms_per_step: the timed region leaves the frame in HBM"}
    sc_n_meshes = sc.info.n_meshes
    sc.close()  # frees this workload's HBM (scene + up to ~190 GB of wavefront workspace) before the next one
    del bufs
    torch.cuda.empty_cache()
    if rank != 0:
        return None

    # roofline of the dominant kernel group on this rank (HIP-event time measured inside the library on its stream).
    # Algorithmic bytes per unit are SURVEY.md 8(d)'s: closest-hit ray 56 B, any-hit ray 36 B, shade vertex 92 B, photon visited 24 B.
    gather_s = agg.get("seconds_photon_gather", 0.0)
    k_times = {"k_trace_closest": agg["seconds_trace_closest"], "k_trace_shadow": agg["seconds_trace_shadow"],
               "k_shade": agg["seconds_shade"], "k_photon_gather": gather_s, "other": agg["seconds_other"]}
    fused_camera = sc_n_meshes == 0 and os.environ.get("BHRT_FUSED_CAMERA", "1") != "0"  # the camera rays of a mesh-free scene are traced inside k_shade
    units = {"k_trace_closest": (agg["closest_rays"] - (agg["camera_samples"] if fused_camera else 0), BYTES_PER_CLOSEST_RAY, agg["launches_trace_closest"]),
             "k_trace_shadow": (agg["shadow_rays"], BYTES_PER_SHADOW_RAY, agg["launches_trace_shadow"]),
             "k_shade": (agg["closest_rays"], BYTES_PER_SHADE_VERTEX, agg["wave_iterations"]),  # one k_shade per wave step (the camera step of a mesh-free scene has no trace kernel of its own)
             "k_photon_gather": (agg.get("photon_nodes_visited", 0), BYTES_PER_PHOTON_VISITED, max(1, agg["passes"]))}
    alone_keys = {"k_trace_closest": "seconds_trace_closest", "k_trace_shadow": "seconds_trace_shadow", "k_shade": "seconds_shade", "k_photon_gather": "seconds_photon_gather"}
    # the dominant group: by the groups' times ALONE where they were measured (beside the pass the any-hit group's event time is mostly waiting for SIMDs)
    overlapped = sc_n_meshes > 0 and os.environ.get("BHRT_SHADOW_OVERLAP", "1") != "0"
    dom = max([k for k in units if not (overlapped and not alone and k == "k_trace_shadow")],
              key=(lambda k: alone.get(alone_keys[k], 0.0)) if alone else (lambda k: k_times[k]))
    n_units, bpu, launches = units[dom]
    launches = max(1, launches)
    avg_launch_s = k_times[dom] / launches
    units_per_launch = n_units / launches
    achieved = bpu * units_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    alone_key = alone_keys[dom]
    prof = pmc_profile(name)
    per_kernel = {k: {"seconds": k_times[k], "units": units[k][0], "bytes_per_unit": units[k][1], "launches": units[k][2],
                      "GBps": (units[k][0] * units[k][1] / k_times[k] / 1e9) if k_times[k] > 0 else 0.0} for k in units}
    group_of = {"k_trace_closest": "k_trace_closest + k_trace_mesh + park sort (closest-hit group of one wave step)",
                "k_trace_shadow": "k_trace_shadow(_park) + k_shadow_mesh (any-hit group of one wave step)", "k_shade": "k_shade",
                "k_photon_gather": "k_photon_gather_* + cell sort (caustic gather of one pass)"}
    res = {
        "workload": f"{os.path.basename(scene_rel)} {W}x{H}, {spp} spp ({'%g' % (spp / N)} per GPU, {ctx.get('scaling', 'weak')} scaling), GI depth {gi}, "
                    f"internal bounces 16, keyed RNG seed 0, {TILE}x{TILE} interleaved tiles over {N} GPU(s)",
        "steps": steps, "warmup": warmup,
        "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": elapsed / steps * 1e3,
        "rays_per_frame": rays_total / steps, "camera_samples_per_frame": samples_total / steps,
        "rays_per_camera_sample": rays_total / max(samples_total, 1.0),
        "wave_steps_per_frame": agg["wave_iterations"] / steps, "passes_per_frame": agg["passes"] / steps,
        "kernel_seconds": k_times, "timers": {0: "k_shade only", 1: "every kernel group", -1: "none"}.get(opts.timers),
        "kernels": per_kernel,
        "roofline": {"bound": "hbm", "kernel": dom, "kernels_timed": group_of[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": prof.get(dom + "_bytes_per_launch"),
                     "avg_launch_ms": avg_launch_s * 1e3, "units_per_launch": units_per_launch, "bytes_per_unit": bpu,
                     # what actually limits the kernel (PMC passes of this round, profiles/pmc_traffic.json): the share of the SIMDs' VALU issue
                     # slots it uses, the lanes enabled in an average VALU instruction, and their product = the share of the chip's lane-slots that do work
                     "limited_by": "valu issue at low lane use + dependent fetches (not HBM): see valu_issue_frac / valu_lanes_per_inst",
                     "valu_issue_frac": prof.get(dom + "_valu_issue_frac"),
                     "valu_lanes_per_inst": prof.get(dom + "_valu_lanes_per_inst"),
                     "valu_lane_frac": (prof.get(dom + "_valu_issue_frac") * prof.get(dom + "_valu_lanes_per_inst") / 64.0)
                     if prof.get(dom + "_valu_issue_frac") and prof.get(dom + "_valu_lanes_per_inst") else None,
                     "note": "`bound`/`frac` are the HBM roofline the contract asks for (algorithmic bytes of SURVEY.md 8(d) over the kernel group's "
                             "HIP-event time); the scene is cache-resident and these kernels are not HBM-bound (DESIGN.md 4)"},
    }
    if alone and alone.get(alone_key, 0) > 0:
        # the same group with the GPU to itself (one untimed frame with the any-hit work in front of the next step instead of beside it)
        units_alone = {"k_trace_closest": alone["closest_rays"], "k_trace_shadow": alone["shadow_rays"], "k_shade": alone["closest_rays"],
                       "k_photon_gather": alone.get("photon_nodes_visited", 0)}[dom]
        res["roofline"]["achieved_alone"] = bpu * units_alone / alone[alone_key] / 1e9
        res["roofline"]["frac_alone"] = res["roofline"]["achieved_alone"] / HBM_PEAK_GBPS
        res["roofline"]["note"] += ("; in the timed region the any-hit group of a wave step runs beside the next step's closest-hit group on a second stream, so "
                                    "`frac` is the group's rate while it SHARES the GPU and `kernel_seconds` sum to more than the frame; `frac_alone` is the "
                                    "same group in one untimed frame with that overlap switched off (`ms_per_step_without_overlap`)")
        res["ms_per_step_without_overlap"] = alone["ms_per_step"]
        res["kernel_seconds_without_overlap"] = {"k_trace_closest": alone["seconds_trace_closest"], "k_trace_shadow": alone["seconds_trace_shadow"],
                                                 "k_shade": alone["seconds_shade"], "k_photon_gather": alone.get("seconds_photon_gather", 0.0), "other": alone["seconds_other"]}
    if dom == "k_photon_gather" and res["roofline"]["traffic"] and avg_launch_s > 0:
        # the photon map is cache-resident: 24 B per examined node is an ALGORITHMIC rate served from L2 / Infinity Cache.  `frac` is the share of
        # the HBM peak that the MEASURED traffic (FETCH_SIZE + WRITE_SIZE passes) amounts to; the algorithmic figure keeps a key of its own.
        res["roofline"]["achieved_algorithmic"] = achieved
        res["roofline"]["frac_algorithmic"] = achieved / HBM_PEAK_GBPS
        res["roofline"]["achieved"] = res["roofline"]["traffic"] / avg_launch_s / 1e9
        res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBPS
    if dom == "k_shade" and fused_camera:
        # The camera step of a mesh-free scene is ONE kernel: k_shade traces its camera rays itself, so the launches timed here also do SURVEY's
        # closest-hit unit (56 B) for every camera sample.  `frac` above stays the conservative figure (shade vertices only); this one adds them.
        with_trace = (bpu * n_units + BYTES_PER_CLOSEST_RAY * agg["camera_samples"]) / launches / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        res["roofline"]["achieved_incl_camera_trace"] = with_trace
        res["roofline"]["frac_incl_camera_trace"] = with_trace / HBM_PEAK_GBPS
    if extra:
        res["wall_clock_extra"] = extra
    if photon_build_s is not None:
        res["photon"] = {"build_s": photon_build_s, "gather_s_per_frame": gather_s / steps,
                         "heavy_pass_s_per_frame": agg.get("seconds_photon_heavy", 0.0) / steps,
                         "queries_per_frame": agg.get("photon_queries", 0) / steps,
                         "heavy_queries_per_frame": agg.get("photon_heavy_queries", 0) / steps,
                         "wave_queries_per_frame": agg.get("photon_wave_queries", 0) / steps,
                         "exact_replay_queries_per_frame": agg.get("photon_exact_queries", 0) / steps,
                         "nodes_visited_per_frame": agg.get("photon_nodes_visited", 0) / steps,
                         # the lane pass alone: kd nodes examined per answered query against the photons the answer is made of (the floor of any walk)
                         "lane_pass_nodes_per_query": agg.get("photon_lane_nodes", 0) / max(1, agg.get("photon_lane_queries", 0)),
                         "lane_pass_found_per_query": found_per_query}
    return res


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks (one per GPU) as a child
    torch.distributed.run, before anything here touches the GPU, and leave with its exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = spp x N, per-GPU work fixed (default); strong = BASELINE's spp (64; 256 for c4) split over the ranks by tiles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone", action="store_true", help="skip the two untimed frames that time the kernel groups alone (any-hit work in front of the next step instead of beside it): profiling runs whose kernel statistics should hold frames of one kind only")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the 'configs' object (c3, c3room, c4, c5)")
    ap.add_argument("--configs", default=",".join(SIDE_CONFIGS), help="workloads reported under 'configs' (default run of the headline workload only)")
    ap.add_argument("--config-steps", type=int, default=3)
    ap.add_argument("--config-warmup", type=int, default=1)
    ap.add_argument("--samples-per-pass", type=int, default=0, help="camera samples in flight per wavefront pass (0 = library default)")
    ap.add_argument("--timers", type=int, default=None, help="bhrt_opts.timers: 0 = HIP events around k_shade only (default for c2, whose dominant kernel it is), 1 = around every kernel group (default otherwise; costs ~0.2 ms of event gaps per frame), -1 = none")
    ap.add_argument("--cpu-sample-spp", type=int, default=0, help="samples per pixel of the cpu_baseline leg (0 = the bounded default of the workload)")
    ap.add_argument("--photons", type=int, default=1000000, help="photon budget of workload c5 (MAX_CausticPhotonCount, Main.cpp:53)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: all ranks share cuda:0 and the gather goes through gloo on host copies, "
                         "to rehearse the N > 1 control flow on a one-GPU box (numbers are meaningless)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and (world_env is None or int(world_env) != args.gpus):
        if world_env is not None and int(world_env) != 1:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with --nproc-per-node {args.gpus}")
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = max(1, int(world_env or "1"))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    elif N > 1 and torch.cuda.device_count() < N:
        raise SystemExit(f"bench.py: {N} ranks but only {torch.cuda.device_count()} GPU(s) visible (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if N > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=5))
        else:  # a rank that dies must not leave the others waiting in a collective for RCCL's default half hour
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))
    if rank == 0:
        ensure_assets()
    if N > 1:
        dist.barrier()
    ctx = {"rank": rank, "local_rank": local_rank, "N": N, "dev": dev, "dist": dist, "rehearse": args.rehearse_on_one_gpu, "scaling": args.scaling, "no_alone": args.no_alone}

    head = run_workload(args.workload, args.steps, args.warmup, ctx, args.photons, args.samples_per_pass, args.timers, extras=True)
    configs = {}
    if args.workload == HEADLINE and not args.no_configs:
        for name in [c for c in args.configs.split(",") if c]:
            try:  # a side workload that fails is reported as such; the headline line is printed regardless
                configs[name] = run_workload(name, 20 if name == "c2" else args.config_steps, 3 if name == "c2" else args.config_warmup, ctx, args.photons)
            except Exception as e:  # noqa: BLE001
                configs[name] = {"error": f"{type(e).__name__}: {e}"[:400]}

    if rank == 0:
        scene_rel, W, H, spp1, gi = WORKLOADS[args.workload]
        out = {
            "metric": "Mrays/s (closest-hit + any-hit rays per second), with wall-clock per frame",
            "value": head["value"],
            "unit": "Mrays/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": head["workload"], "rays_per_frame": head["rays_per_frame"],
                       "camera_samples_per_frame": head["camera_samples_per_frame"],
                       "framebuffer_gather": "one rccl all_gather of packed tiles (float radiance + rgb8), native pack/unpack kernels" if N > 1 else "none"},
            "kernel_seconds": head["kernel_seconds"], "timers": head["timers"], "kernels": head["kernels"],
            "wave_steps_per_frame": head["wave_steps_per_frame"],
            "roofline": head["roofline"],
        }
        for k in ("ms_per_step_without_overlap", "kernel_seconds_without_overlap"):
            if k in head:
                out[k] = head[k]
        if "photon" in head:
            out["photon"] = head["photon"]
        if "wall_clock_extra" in head:
            out["wall_clock_extra"] = head["wall_clock_extra"]
        if configs:
            out["configs"] = configs
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(os.path.join(ROOT, scene_rel), args.cpu_sample_spp or CPU_SAMPLE_SPP.get(args.workload, spp1), gi, full_spp=spp1)
        print(json.dumps(out), flush=True)
    if N > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 — the line is out; a broken group must not hide the exit code below
            pass
    failed = [k for k, v in configs.items() if isinstance(v, dict) and "error" in v]
    if failed:  # the line is printed with the error in it, but a side workload that breaks fails the run
        print(f"bench.py: side workload(s) failed: {failed}", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
