#!/usr/bin/env python3
"""N logical ranks on ONE GPU (gloo, blocks moved through the host): every rank renders its tiles, the frame is
exchanged with the native pack / unpack kernels, and rank 0 checks the result byte for byte against its own
single-rank render.  Run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/verify_multi_rank.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bhraytracer_amd as B
from bhraytracer_amd import dist as BD

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
ok = True
for scene, spp in (("c2_glass_small.xml", 4), ("c3_mesh_small.xml", 2)):
    sc = B.Scene(os.path.join(ROOT, "tests", "scenes", scene)); sc.upload(0)
    W, H = sc.width, sc.height
    for tile in (32, 16):
        opts = B.default_opts(spp=spp, gi_bounces=2, seed=5, rank=rank, world_size=world, tile_size=tile)
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        sc.render_dev(opts, rgb.data_ptr(), rad.data_ptr())
        BD.gather_frame_dev(rgb, rad, tile, rank, world, via_host=True)
        torch.cuda.synchronize()
        if rank == 0:
            o1 = B.default_opts(spp=spp, gi_bounces=2, seed=5)
            rgb1, rad1, _ = sc.render(o1)
            same = bool((rgb.cpu().numpy() == rgb1).all()) and rad.cpu().numpy().tobytes() == rad1.tobytes()
            print(f"{scene} {W}x{H} tile {tile} world {world}: {'identical' if same else 'MISMATCH'}", flush=True)
            ok = ok and same
# the caustic photon map built by all ranks together (emission sharded by index range, one all_gather per batch) == built alone
sc = B.Scene(os.path.join(ROOT, "tests", "scenes", "c5_caustics.xml")); sc.upload(0)
o = B.default_opts(seed=3)
n = BD.photon_build_sharded(sc, o, 20000, rank, world, batch=1 << 18)
shared = sc.photon_get()
sc.photon_build(o, 20000)
same = n == 20000 and bool((shared == sc.photon_get()).all())
ok = ok and same
print(f"rank {rank}: photon map built over {world} ranks: {'identical' if same else 'MISMATCH'}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
