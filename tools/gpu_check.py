#!/usr/bin/env python3
"""Quick GPU-vs-oracle parity probe (development aid; the real checks are tests/ -m gpu)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bhraytracer_amd as B
import oracle_lib as O

def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)

def check_scene(path, spp=4, gi=3, region=None, render=True):
    print("==", path, flush=True)
    sc = B.Scene(path)
    fv = sc.flat_view(); blob = sc.flat_bytes()
    o, d = O.primary_rays(fv)
    t0 = time.time(); g = sc.trace_closest(o, d, 1); t1 = time.time()
    r = O.trace_closest(blob, o, d, 1)
    print(f" primary: n={len(o)} hits={int((r['node']>=0).sum())} node_eq={np.array_equal(g['node'], r['node'])} "
          f"prim_eq={np.array_equal(g['prim'], r['prim'])} t_biteq={np.array_equal(bits(g['t']), bits(r['t']))} "
          f"front_eq={np.array_equal(g['front'][r['node']>=0], r['front'][r['node']>=0])} gpu_s={t1-t0:.3f}", flush=True)
    for side in (2, 3):
        g2 = sc.trace_closest(o, d, side); r2 = O.trace_closest(blob, o, d, side)
        print(f" side={side}: node_eq={np.array_equal(g2['node'], r2['node'])} t_biteq={np.array_equal(bits(g2['t']), bits(r2['t']))} prim_eq={np.array_equal(g2['prim'], r2['prim'])}")
    # shadow rays from primary hit points to the first point light
    hit = r['node'] >= 0
    P = r['attrs'][hit][:, 1:4]
    lights = [l for l in fv.lights if l.type == 2]
    if lights and hit.any():
        L = np.array(list(lights[-1].vec), np.float32)
        sd = (L[None, :] - P).astype(np.float32)
        gv = sc.trace_shadow(P, sd, 1.0); rv = O.trace_shadow(blob, P, sd, 1.0)
        print(f" shadow: n={len(P)} occluded={int((rv==0).sum())} equal={np.array_equal(gv, rv)}")
    if render:
        W, H = sc.width, sc.height
        reg = region or (0, 0, W, H)
        opts = B.default_opts(spp=spp, gi_bounces=gi)
        t0 = time.time(); gs, st = sc.render_samples(opts, *reg); t1 = time.time()
        ro = O.render(blob, W, H, spp, gi=gi, region=reg)
        rs = ro['samples']
        eq = (bits(gs) == bits(rs)) | (np.isnan(gs) & np.isnan(rs))
        bad = ~eq.all(axis=2)
        print(f" render samples: {bad.size} samples, mismatching={int(bad.sum())} maxabs={np.nanmax(np.abs(gs-rs)):.3g} gpu_s={t1-t0:.3f}")
        print(f"   gpu stats: closest={st.closest_rays} shadow={st.shadow_rays} shade={st.shade_calls} iters={st.wave_iterations} | oracle closest={ro['stats'].closest_rays} shadow={ro['stats'].shadow_rays} shade={ro['stats'].shade_calls}")
        if bad.any():
            for p, s in np.argwhere(bad)[:5]: print("   pix", p, "s", s, "gpu", gs[p, s], "oracle", rs[p, s])
        rgb, rad, st2 = sc.render(opts)
        ro2 = O.render(blob, W, H, spp, gi=gi, want_samples=False)
        print(f" full frame: rgb8_equal={np.array_equal(rgb, ro2['rgb8'])} radiance_biteq={np.array_equal(bits(rad), bits(ro2['radiance']))} maxabs={np.abs(rad-ro2['radiance']).max():.3g} secs={st2.seconds_total:.3f}")

if __name__ == "__main__":
    print("devices:", B.device_count())
    S = os.path.join(ROOT, "tests", "scenes")
    check_scene(os.path.join(S, "c1_sphere_plane.xml"), spp=2)
    check_scene(os.path.join(S, "c3_mesh_small.xml"), spp=2)
    import tempfile
    txt = open(os.path.join(S, "c2_glass.xml")).read().replace('<width value="1920"/>', '<width value="480"/>').replace('<height value="1080"/>', '<height value="270"/>')
    p = os.path.join(tempfile.gettempdir(), "c2_small.xml"); open(p, "w").write(txt)
    check_scene(p, spp=4)
