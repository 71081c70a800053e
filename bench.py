#!/usr/bin/env python3
"""bench.py — Mrays/s + wall-clock per frame of the HIP render path on BASELINE.json's config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c1]

A "step" = one full frame of the workload through the hot path (camera rays -> wavefront
trace/shade steps -> combine -> resolve [-> RCCL framebuffer gather when N > 1]).
N = 1 default workload = BASELINE.json configs[1]: tests/scenes/c2_glass.xml, 1920x1080, 16 spp,
GI depth 3 (4 levels = "reflection/refraction depth 4"), internal bounces 16, keyed RNG seed 0.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); interleaved 32x32 tiles
(tile t -> rank t mod N), spp scaled to 16*N so per-GPU work is fixed (weak
scaling), then one all_gather of the packed tile buffers over xGMI inside the timed region.
The scene is resident in HBM before the timed region; outputs stay in HBM.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, spp, gi)
    "c1": ("tests/scenes/c1_sphere_plane.xml", 640, 480, 1, -1),
    "c2": ("tests/scenes/c2_glass.xml", 1920, 1080, 16, 3),
    "c3": ("tests/scenes/c3_mesh.xml", 1920, 1080, 64, 3),
    # BASELINE config 4: the mesh scene at 3840x2160, 256 spp on 8 GPUs = 32 spp per GPU (weak scaling like every workload here)
    "c4": ("tests/scenes/c4_mesh_4k.xml", 3840, 2160, 32, 3),
    # BASELINE config 5: caustic photon map, 1 M photons, k = 1000, r = 0.5 (photon build timed separately, see "photon_build_s")
    "c5": ("tests/scenes/c5_caustics_hd.xml", 1920, 1080, 64, 3),
}
BYTES_PER_CLOSEST_RAY = 56  # 32 B ray read + 24 B hit write
BYTES_PER_SHADOW_RAY = 36   # 32 B read + 4 B visibility write
BYTES_PER_SHADE_VERTEX = 92  # 24 hit + 32 ray read, 12 radiance, 24 next-ray writes
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def ensure_assets():
    mesh = os.path.join(ROOT, "tests/scenes/gen/mesh_224.obj")
    if not os.path.exists(mesh):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_mesh
        gen_mesh.generate(mesh, 224)


def cpu_baseline(scene_path, spp, gi):
    """Reference CPU path on a bounded sample of the same workload, on this box's host cores.

    kind "reference": the reference itself (oracle/_ref/ref_harness, compiled from /root/reference in the dev
    container) run as one single-threaded process per row band (its RNG is a process-global);
    kind "port": the oracle (sequential RNG + libm = the mode pinned bit-for-bit to the reference), OpenMP.
    Ray counts come from the oracle (identical control flow)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import bhraytracer_amd as B
    sc = B.Scene(scene_path)
    blob = sc.flat_bytes()
    W, H = sc.width, sc.height
    cores = max(1, min(os.cpu_count() or 1, 16))  # the GPU box's CPU share for one GPU
    rows = max(cores, (H // 8) // cores * cores)  # about 1/8 of the frame, a multiple of the core count
    y0 = (H - rows) // 2
    region = (0, y0, W, y0 + rows)
    t0 = time.time()
    ro = O.render(blob, W, H, spp, gi=gi, rng=O.RNG_SEQUENTIAL, math=O.MATH_LIBM, region=region, threads=cores,
                  want_samples=False)
    port_s = time.time() - t0
    rays = ro["stats"].closest_rays + ro["stats"].shadow_rays
    sample = f"rows {y0}..{y0 + rows} of {W}x{H} ({rows * W} pixels x {spp} spp, {rays} rays)"
    out = {"value": rays / port_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": sample}
    harness = os.path.join(ROOT, "oracle/_ref/ref_harness")
    if os.path.exists(harness):
        try:
            band = rows // cores
            tmp = tempfile.mkdtemp(prefix="bhrt_ref_")
            t0 = time.time()
            procs = []
            for k in range(cores):
                a, b = y0 + k * band, y0 + (k + 1) * band
                cmd = [harness, os.path.abspath(scene_path), os.path.join(tmp, f"b{k}"), "--spp", str(spp), "--gi", str(gi),
                       "--region", "0", str(a), str(W), str(b), "render"]
                procs.append(subprocess.Popen(cmd, cwd=os.path.dirname(os.path.abspath(scene_path)),
                                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
            ok = all(p.wait() == 0 for p in procs)
            ref_s = time.time() - t0
            if ok:
                out = {"value": rays / ref_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
                       "sample": sample + f"; {cores} single-threaded reference processes (incl. scene load)",
                       "port_value": rays / port_s / 1e6}
            subprocess.run(["rm", "-rf", tmp])
        except Exception:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--samples-per-pass", type=int, default=0, help="camera samples in flight per wavefront pass (0 = library default)")
    ap.add_argument("--timers", type=int, default=None, help="bhrt_opts.timers: 0 = HIP events around k_shade only (default for c2, whose dominant kernel it is), 1 = around every kernel group (default otherwise; costs ~0.2 ms of event gaps per frame), -1 = none")
    ap.add_argument("--photons", type=int, default=1000000, help="photon budget of workload c5 (MAX_CausticPhotonCount, Main.cpp:53)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: all ranks share cuda:0 and the gather goes through gloo on host copies, "
                         "to rehearse the N > 1 control flow on a one-GPU box (numbers are meaningless)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    N = max(1, world)

    import numpy as np
    import torch
    import bhraytracer_amd as B
    import bhraytracer_amd.dist as BD

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if N > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    if rank == 0:
        ensure_assets()
    if N > 1:
        dist.barrier()
    scene_rel, W, H, spp1, gi = WORKLOADS[args.workload]
    scene_path = os.path.join(ROOT, scene_rel)
    sc = B.Scene(scene_path)
    assert (sc.width, sc.height) == (W, H), "scene size differs from the workload table"
    sc.upload(local_rank)
    spp = spp1 * N  # weak scaling: per-GPU sample count is fixed
    tile = 32
    opts = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0, rank=rank, world_size=N, tile_size=tile)
    opts.samples_per_pass = args.samples_per_pass
    opts.timers = args.timers if args.timers is not None else (0 if args.workload == "c2" else 1)
    photon_build_s = None
    if args.workload == "c5":
        t0 = time.perf_counter()
        if N > 1:  # emission sharded over the ranks by emission-index range, one all_gather per batch, the same map on every rank
            n_ph = BD.photon_build_sharded(sc, opts, args.photons, rank, N, device=None if args.rehearse_on_one_gpu else dev)
        else:
            n_ph = sc.photon_build(opts, args.photons)
        torch.cuda.synchronize()
        photon_build_s = time.perf_counter() - t0
        opts.photon_map = 1

    # Two frame buffers (+ exchange scratch) used alternately: the exchange of frame k (torch's stream) overlaps the
    # rendering of frame k+1 (the library's stream); a buffer is rendered into again only after its last exchange is done.
    n_buf = 2 if N > 1 else 1
    bufs = [(torch.zeros((H, W, 3), dtype=torch.uint8, device=dev), torch.zeros((H, W, 3), dtype=torch.float32, device=dev), {},
             torch.cuda.Event()) for _ in range(n_buf)]
    d_rgb, d_rad = bufs[0][0], bufs[0][1]
    frame_no = [0]

    def step():
        rgb, rad, scratch, ev = bufs[frame_no[0] % n_buf]
        frame_no[0] += 1
        if N > 1:
            ev.synchronize()  # no-op until the event has been recorded once
        st = sc.render_dev(opts, rgb.data_ptr(), rad.data_ptr())
        if N > 1:
            # pack -> ONE all_gather of byte blocks (RCCL; through the host with gloo when rehearsing on one GPU) -> unpack
            BD.gather_frame_dev(rgb, rad, tile, rank, N, scratch=scratch, via_host=args.rehearse_on_one_gpu)
            ev.record()
        return st

    def sync():
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()  # untimed: first use allocates the wavefront workspace (tens of GB of hipMalloc) — not part of any timed or warmup step
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    agg = None
    for _ in range(args.steps):
        st = step()
        d = st.as_dict()
        agg = d if agg is None else {k: agg[k] + d[k] for k in d}
    sync()
    elapsed = time.perf_counter() - t0
    rays_local = agg["closest_rays"] + agg["shadow_rays"]
    if N > 1:
        rdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        tt = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays_local, agg["closest_rays"], agg["camera_samples"]], dtype=torch.float64, device=rdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total, closest_total, samples_total = (float(x) for x in rr.tolist())
    else:
        rays_total, closest_total, samples_total = float(rays_local), float(agg["closest_rays"]), float(agg["camera_samples"])

    if rank == 0:
        # roofline of the dominant kernel on this rank (HIP-event time measured inside the library on its stream).
        # Algorithmic bytes per unit are SURVEY.md 8(d)'s: closest-hit ray 56 B, any-hit ray 36 B, shade vertex 92 B.
        k_times = {"k_trace_closest": agg["seconds_trace_closest"], "k_trace_shadow": agg["seconds_trace_shadow"],
                   "k_shade": agg["seconds_shade"], "other": agg["seconds_other"]}
        units = {"k_trace_closest": (agg["closest_rays"], BYTES_PER_CLOSEST_RAY, agg["launches_trace_closest"]),
                 "k_trace_shadow": (agg["shadow_rays"], BYTES_PER_SHADOW_RAY, agg["launches_trace_shadow"]),
                 "k_shade": (agg["closest_rays"], BYTES_PER_SHADE_VERTEX, agg["launches_trace_closest"])}
        dom = max(units, key=lambda k: k_times[k])
        n_units, bpu, launches = units[dom]
        launches = max(1, launches)
        avg_launch_s = k_times[dom] / launches
        units_per_launch = n_units / launches
        achieved = bpu * units_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get(args.workload, {}).get(dom + "_bytes_per_launch")
            except Exception:
                traffic = None
        per_kernel = {k: {"seconds": k_times[k], "units": units[k][0], "bytes_per_unit": units[k][1], "launches": units[k][2],
                          "GBps": (units[k][0] * units[k][1] / k_times[k] / 1e9) if k_times[k] > 0 else 0.0} for k in units}
        out = {
            "metric": "Mrays/s (closest-hit + any-hit rays per second), with wall-clock per frame",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{os.path.basename(scene_rel)} {W}x{H}, {spp} spp ({spp1} per GPU), GI depth {gi}, "
                                   f"internal bounces 16, keyed RNG seed 0, {tile}x{tile} interleaved tiles over {N} GPU(s)",
                       "rays_per_frame": rays_total / args.steps, "camera_samples_per_frame": samples_total / args.steps,
                       "framebuffer_gather": "one rccl all_gather of packed tiles (float radiance + rgb8), native pack/unpack kernels" if N > 1 else "none"},
            "photon_build_s": photon_build_s, "photon_gather_s_per_frame": (agg.get("reserved0", 0.0) / args.steps) if photon_build_s else None,
            "photon_heap_pass_s_per_frame": (agg.get("reserved1", 0.0) / args.steps) if photon_build_s else None,
            "photon_heap_queries_per_frame": (agg.get("reserved2", 0.0) / args.steps) if photon_build_s else None,
            "photon_wave_queries_per_frame": (agg.get("reserved3", 0.0) / args.steps) if photon_build_s else None,
            "kernel_seconds": k_times, "timers": {0: "k_shade only", 1: "every kernel group", -1: "none"}.get(opts.timers),
            "kernels": per_kernel,
            "wave_steps_per_frame": agg["wave_iterations"] / args.steps,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "avg_launch_ms": avg_launch_s * 1e3, "units_per_launch": units_per_launch,
                         "bytes_per_unit": bpu,
                         "note": "scene is cache-resident; traversal/shading are latency- and ALU-bound, not HBM-bound (DESIGN.md)"},
        }
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_path, spp1, gi)
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
