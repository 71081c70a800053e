#!/usr/bin/env python3
"""Whole-frame parity soak (needs a GPU; the oracle is the checker): every sample of full-size frames of the bench scenes, several seeds,
HIP path against the oracle's keyed-RNG run, bit for bit.  ~100 s on a GPU box for 3.3e8 samples / 1.9e9 rays.
Usage: python tools/soak_parity.py [caustics] [leaf_skip]     (caustics: BASELINE config 5 instead — 1 M photons, whole frame, ~1 min;
leaf_skip: the mesh scenes with bhrt_opts.leaf_skip = 1)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bhraytracer_amd as B, oracle_lib as O

if "caustics" in sys.argv[1:]:
    sc = B.Scene(os.path.join("tests/scenes", "c5_caustics_hd.xml")); blob = sc.flat_bytes()
    N = 1000000
    sc.photon_build(B.default_opts(seed=0), N)
    bal, _, n_emit = O.photon_build(blob, N, seed=0)
    print("caustic map, %d photons from %d emissions: identical on both sides: %s" % (N, n_emit, np.array_equal(sc.photon_get(), bal)), flush=True)
    region = (0, 0, sc.width, sc.height)
    opts = B.default_opts(spp=2, gi_bounces=3, seed=1, photon_map=1); opts.photon_exact = 1
    gx, stx = sc.render_samples(opts, *region)
    ro = O.render(blob, sc.width, sc.height, 2, gi=3, seed=1, region=region, photon=1, threads=16)["samples"]
    same = (gx.view(np.uint32) == ro.view(np.uint32)) | (np.isnan(gx) & np.isnan(ro))
    print("exact replay: mismatching values %d of %d, heavy queries %d" % (int((~same).sum()), same.size, stx.photon_heavy_queries), flush=True)
    opts.photon_exact = 0
    gs, st = sc.render_samples(opts, *region)
    print("selection pass: max |diff| %.3g (bar 1e-4), differing values %d" % (float(np.nanmax(np.abs(gs - ro))), int((gs.view(np.uint32) != ro.view(np.uint32)).sum())), flush=True)
    sys.exit(0)
total = 0; bad = 0; rays = 0
t00 = time.time()
for name, spp, seeds in (("c3_room", 16, (1, 2, 3, 4)), ("c3_mesh", 16, (1, 2)), ("c4_mesh_4k", 4, (1, 2)), ("c2_glass", 16, (1, 2))):
    sc = B.Scene(os.path.join("tests/scenes", name + ".xml")); blob = sc.flat_bytes()
    for seed in seeds:
        t0 = time.time()
        o = B.default_opts(spp=spp, gi_bounces=3, seed=seed)
        o.leaf_skip = 1 if "leaf_skip" in sys.argv[1:] else 0
        gs, st = sc.render_samples(o, 0, 0, sc.width, sc.height)
        ro = O.render(blob, sc.width, sc.height, spp, gi=3, seed=seed, region=(0, 0, sc.width, sc.height), threads=16)["samples"]
        same = (gs.view(np.uint32) == ro.view(np.uint32)) | (np.isnan(gs) & np.isnan(ro))
        nb = int((~same).sum())
        total += gs.shape[0] * spp; bad += nb; rays += st.closest_rays + st.shadow_rays
        print(name, "spp", spp, "seed", seed, "%.0fs" % (time.time() - t0), "mismatching values:", nb, "deferred (axis-parallel) rays", st.deferred_rays, flush=True)
        del gs, ro, same
print("samples compared: %d, rays behind them: %d, mismatching values: %d, %.0f s" % (total, rays, bad, time.time() - t00))
