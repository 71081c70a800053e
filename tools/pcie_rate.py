#!/usr/bin/env python3
"""C2 frame through bhrt_render (host buffers: RGB8 + float radiance cross PCIe) next to bhrt_render_dev."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bhraytracer_amd as B
import bench
scene_rel, W, H, spp, gi = bench.WORKLOADS["c2"]
sc = B.Scene(os.path.join(ROOT, scene_rel)); sc.upload(0)
opts = B.default_opts(spp=spp, gi_bounces=gi, internal_bounces=16, seed=0)
rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for name, fn in (("bhrt_render_dev", lambda: sc.render_dev(opts, rgb.data_ptr(), rad.data_ptr())), ("bhrt_render (host buffers)", lambda: sc.render(opts)[2])):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); rays = 0
    for _ in range(5):
        st = fn(); rays += st.closest_rays + st.shadow_rays
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / 5 * 1e3:.2f} ms/frame, {rays / dt / 1e6:.0f} Mrays/s")
