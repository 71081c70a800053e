#!/usr/bin/env python3
"""Load balance of the interleaved tile partition (tile t -> rank t mod N), measured on ONE GPU with logical ranks:
every rank of an N-rank job renders its tiles of one frame in turn and the library's wall clock and ray counts are recorded per rank.
On N GPUs the frame time is the slowest rank's, so max / mean over the ranks is what the static interleave costs.
    python tools/rank_balance.py [N] [workload ...]   ->  JSON on stdout (committed as profiles/<round>/rank_balance.json)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bhraytracer_amd as B
import bench


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    wls = sys.argv[2:] or ["c2", "c3", "c3room", "c4", "c5"]
    bench.ensure_assets()
    out = {"world": N, "tile": bench.TILE, "note": "logical ranks on one MI355X, one frame per workload at BASELINE's total sample count (spp of the "
                                                     "bench workload x 1 GPU), split over N ranks; seconds = bhrt_stats.seconds_total of each rank's render"}
    for wl in wls:
        scene_rel, W, H, spp, gi = bench.WORKLOADS[wl]
        sc = B.Scene(os.path.join(ROOT, scene_rel)); sc.upload(0)
        base = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0, tile_size=bench.TILE)
        if wl == "c5":
            sc.photon_build(base, 1000000)
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        secs, rays = [], []
        for rep in range(2):  # the first round allocates the workspace
            secs, rays = [], []
            for r in range(N):
                o = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0, tile_size=bench.TILE, rank=r, world_size=N, photon_map=1 if wl == "c5" else 0)
                st = sc.render_dev(o, rgb.data_ptr(), rad.data_ptr())
                secs.append(st.seconds_total); rays.append(st.closest_rays + st.shadow_rays)
        mean = sum(secs) / N
        out[wl] = {"seconds_per_rank": secs, "rays_per_rank": rays, "max_over_mean_seconds": max(secs) / mean, "min_over_mean_seconds": min(secs) / mean,
                   "max_over_mean_rays": max(rays) / (sum(rays) / N), "frame_seconds_if_parallel": max(secs), "sum_seconds": sum(secs)}
        sc.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
